#!/bin/bash
# Round-end evidence: GPU test suite, inference bench + rocprof kernel stats, train bench + stats, PMC traffic passes
# (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes) for the MLP kernels.  usage: gpu_final.sh TAG
set -u
TAG=${1:-r01i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu_$TAG.log
timeout 600 python bench.py --steps 50 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench_$TAG.json
timeout 600 python bench.py --mode train --steps 20 --warmup 3 > $OUT/bench_train_$TAG.json 2> /dev/null; cut -c1-200 $OUT/bench_train_$TAG.json
timeout 600 python bench.py --mode render --steps 5 --warmup 2 --no-graph > $OUT/bench_render_$TAG.json 2> /dev/null; cut -c1-200 $OUT/bench_render_$TAG.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof inference rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -o bench -- python $ROOT/bench.py --mode train --steps 7 --warmup 2 > $OUT/rocprof_train_$TAG.log 2>&1; echo "rocprof train rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  # the headline kernel (fused-IPE k_mlp_bf16 launched by mipnerf_forward) as bench.py runs it
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcb_${TAG}_$c -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmcb_${TAG}_$c.log 2>&1; echo "pmc bench $c rc=$?"
  f=$(find $OUT/pmcb_${TAG}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_mlp_bf16" in r["Kernel_Name"]]
print(f"  bench k_mlp_bf16 {sys.argv[1].split('_')[-3] if False else ''} mean per dispatch {sum(v)/max(1,len(v)):.6g} KB (n={len(v)})")
PY
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}_$c -o pmc -- python $ROOT/scripts/prof_train.py --iters 3 > $OUT/pmc_${TAG}_$c.log 2>&1; echo "pmc $c rc=$?"
  f=$(find $OUT/pmc_${TAG}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "wgrad" if "k_mlp_wgrad" in n else "trainfwd" if "trainfwd" in n else "dgrad" if "dgrad" in n else "infer" if "k_mlp_bf16" in n else None
    if k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"  {k} {c}: mean per dispatch {sum(v)/len(v):.6g} KB (n={len(v)})")
PY
done
