#!/bin/bash
# GPU session for the native training kernels: their parity tests first, then the whole GPU suite, then the benches.
set -u
TAG=${1:-trn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -k native > $OUT/pytest_native_$TAG.log 2>&1; echo "pytest native rc=$?" | tee -a $OUT/pytest_native_$TAG.log
tail -40 $OUT/pytest_native_$TAG.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "not native" > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -15 $OUT/pytest_gpu_$TAG.log
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_train_$TAG.json 2> $OUT/bench_train_$TAG.err; echo "bench train rc=$?"
cat $OUT/bench_train_$TAG.json; tail -3 $OUT/bench_train_$TAG.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -o bench -- python $ROOT/bench.py --mode train --steps 5 --warmup 2 > $OUT/rocprof_train_$TAG.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof_train_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -16
