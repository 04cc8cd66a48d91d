#!/bin/bash
# One GPU session: self-test, parity tests, bench, rocprof.  Everything is logged under gpurun_out/.
# usage: scripts/gpu_round.sh [tag]
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== rocminfo ==" > $OUT/env_$TAG.log
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name") >> $OUT/env_$TAG.log 2>&1
echo "== smoke ==" | tee $OUT/smoke_$TAG.log
timeout 600 python __graft_entry__.py smoke >> $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke_$TAG.log
tail -5 $OUT/smoke_$TAG.log
echo "== pytest -m gpu =="
rm -f $OUT/parity.jsonl
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -40 $OUT/pytest_gpu_$TAG.log
echo "== bench =="
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
cat $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
echo "== rocprof =="
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $ROOT
find $OUT/prof_$TAG -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
