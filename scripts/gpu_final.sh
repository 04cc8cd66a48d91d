#!/bin/bash
# Round-end evidence (round 2): GPU test suite, the default bench line (headline + train + render sub-records), rocprofv3
# kernel stats of the same command and of the fp32 / train-only runs, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in
# separate passes, as the guide prescribes).  usage: gpu_final.sh TAG     (everything lands in gpurun_out/TAG/)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $ROOT/gpurun_out/parity.jsonl
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
cp $ROOT/gpurun_out/parity.jsonl $OUT/parity.jsonl 2>/dev/null
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
timeout 300 python bench.py --precision fp32 --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 10 > $OUT/bench_fp32.json 2>/dev/null; cut -c1-200 $OUT/bench_fp32.json
timeout 300 python bench.py --precision fp32 --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 5 --rays 8192 --samples 256 > $OUT/bench_fp32_c4.json 2>/dev/null
timeout 300 python bench.py --precision fp32 --mode train --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_fp32_train.json 2>/dev/null; cut -c1-200 $OUT/bench_fp32_train.json
timeout 300 python bench.py --mode train --no-graph --steps 30 > $OUT/bench_train_eager.json 2>/dev/null
cd /tmp
stats() {  # name, args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o bench -- python $ROOT/bench.py "$@" > $OUT/rocprof_$name.log 2>&1; echo "rocprof $name rc=$?"
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
}
stats bench_inference --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 20      # the headline kernel alone: its average = roofline.launch_ms
stats bench --no-cpu-baseline --sustain-seconds 0 --steps 20                                  # the default line (inference + train + render sub-records)
stats train --mode train --steps 10 --warmup 2 --no-graph
stats fp32_inference --precision fp32 --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 5
stats fp32_train --precision fp32 --mode train --steps 3 --warmup 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --mode inference --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 > $OUT/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_mlp_bf16" in r["Kernel_Name"]]
print(f"{sys.argv[2]} k_mlp_bf16 mean per dispatch {sum(v)/max(1,len(v)):.6g} KB (n={len(v)})")
PY
  rm -rf $OUT/pmc_$c
done
OUT=$OUT bash $ROOT/scripts/pmc_train.sh > $OUT/pmc_train.log 2>&1; tail -6 $OUT/pmc_train.log
date
