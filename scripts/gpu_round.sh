#!/bin/bash
# One gpurun call = tests + bench + micro-benchmarks, everything written under gpurun_out/$TAG/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r02a [tests|notests] [extra commands file]'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  echo "== pytest -m gpu" ; date
  timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -25 $OUT/pytest.log | cut -c1-300
  [ -f gpurun_out/parity.jsonl ] && cp gpurun_out/parity.jsonl $OUT/parity.jsonl
fi
echo "== bench" ; date
timeout 600 python bench.py --no-cpu-baseline --sustain-seconds 0.5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 4000 $OUT/bench.json; tail -5 $OUT/bench.err
if [ -n "$3" ] && [ -f "$3" ]; then
  echo "== extra: $3"; date
  OUT=$OUT bash "$3" 2>&1 | tee $OUT/extra.log | tail -60
fi
date
