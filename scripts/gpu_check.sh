#!/bin/bash
# Quick GPU session: parity tests + inference bench + train bench.  usage: scripts/gpu_check.sh [tag]
set -u
TAG=${1:-chk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|Compute Unit" | head -4; nproc; lscpu | grep "Model name") > $OUT/env_$TAG.log 2>&1
rm -f $OUT/parity.jsonl
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -15 $OUT/pytest_gpu_$TAG.log
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > $OUT/bench_train_$TAG.json 2> $OUT/bench_train_$TAG.err; echo "bench train rc=$?"
cat $OUT/bench_train_$TAG.json; tail -3 $OUT/bench_train_$TAG.err
