#!/bin/bash
# round 5, third GPU call: parity suite on the bf16 kernel of the two-view-layer variant and the fused small launches at every K bucket;
# the second schedule of the recompute-in-wgrad probe (two accumulator chains, operands four k-steps ahead) against the first and the product
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r05c}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -q -m gpu -k "not trained_unbounded_field" 2>&1 | tail -15 > gpurun_out/${T}_pytest_gpu_tail.txt
tail -8 gpurun_out/${T}_pytest_gpu_tail.txt
cp gpurun_out/parity.jsonl gpurun_out/${T}_parity.jsonl
C=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc
export MIPNERF_ALLOW_EXPERIMENT_LIB=1 MIPNERF_ZERO_SCRATCH=1
LIBS="libmipnerf_hip.so libmipnerf_hip_skrc.so libmipnerf_hip_skrc2.so"
for i in 1 2 3; do
  for lib in $LIBS; do
    MIPNERF_LIB=$C/$lib timeout 200 python bench.py --mode train --steps 50 --warmup 5 --no-cpu-baseline --preheat-seconds 2 2>gpurun_out/ab_train.err | python -c "
import sys, json
ls = [x for x in sys.stdin if x.startswith('{')]
if not ls: print('$lib', 'no line (non-finite loss?)')
else:
    l = json.loads(ls[-1]); print('$lib', l['ms_per_step'])"
  done
done | tee gpurun_out/${T}_recompute_probe_sched_ab.txt
cd /tmp
for lib in libmipnerf_hip_skrc2.so; do
  MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t -o train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 30 --warmup 5 --no-graph --no-cpu-baseline --preheat-seconds 2 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_train_${lib%.so}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t
  echo "== $lib"; head -4 $GRAFT_REPO_ROOT/gpurun_out/${T}_train_${lib%.so}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-300
done
# cycle counters of the weight-gradient kernel, product vs second probe schedule (separate --pmc passes; busy cycles and MFMA-busy cycles)
for lib in libmipnerf_hip.so libmipnerf_hip_skrc2.so; do
  for c in "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=$(echo $c | tr ' ' '_')
    MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_c/${lib%.so}_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-graph --no-cpu-baseline --preheat-seconds 0 > /dev/null 2>&1; echo "pmc $lib $tag rc=$?"
  done
done
python - $GRAFT_REPO_ROOT/gpurun_out/pmc_c <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/${T}_recompute_probe_cycles.txt
import csv, glob, sys, collections
out = sys.argv[1]
for lib in ("libmipnerf_hip", "libmipnerf_hip_skrc2"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{lib}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp_wgrad" in r["Kernel_Name"]:
                vals["wgrad"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in vals.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        print(lib, k, {c: f"{x:.4g}" for c, x in m.items()},
              "MFMA-busy / SQ-busy = %.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1)))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_c
