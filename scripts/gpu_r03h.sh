#!/bin/bash
# round 3: PMC traffic of the headline MLP kernel and of the training kernels (separate --pmc passes), train kernel stats, MALL probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_fwd_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_fwd_$c -o pmc -- python $GRAFT_REPO_ROOT/scripts/prof_fwd.py --iters 6 > /tmp/pmc_fwd_$c.log 2>&1
  python - /tmp/pmc_fwd_$c $c <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r03h_pmc_fwd.txt
import csv, sys, glob, collections
d, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    n = r["Kernel_Name"]
    k = "k_mlp_bf16" if "k_mlp_bf16" in n else "k_composite_resample" if "composite_resample" in n else "k_volumetric_rendering" if "volumetric" in n else "k_ray_prologue" if "prologue" in n else None
    if k: acc[k].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{c} {k}: mean per dispatch {sum(v)/len(v):.6g} KB (n={len(v)})")
PY
done
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h_pmc_train bash $GRAFT_REPO_ROOT/scripts/pmc_train.sh > $GRAFT_REPO_ROOT/gpurun_out/r03h_pmc_train.log 2>&1
tail -8 $GRAFT_REPO_ROOT/gpurun_out/r03h_pmc_train.log
rm -rf /tmp/prof_train
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 20 --warmup 3 --no-graph --no-cpu-baseline --preheat-seconds 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r03h_train_prof.json 2>/dev/null
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03h_train_kernel_stats.csv && head -8 "$f" | cut -c1-150
cd $GRAFT_REPO_ROOT && timeout 200 python scripts/micro/mall_probe.py > gpurun_out/r03h_mall_probe.log 2>&1; cat gpurun_out/r03h_mall_probe.log | tail -6
