#!/bin/bash
# One gpurun call = tests + bench + micro-benchmarks, everything written under gpurun_out/$TAG/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r02a'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" ; date
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
[ -f gpurun_out/parity.jsonl ] && cp gpurun_out/parity.jsonl $OUT/parity.jsonl
echo "== bench" ; date
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
if [ -x scripts/micro/mfma_peak.out ]; then
  echo "== mfma_peak" ; date
  timeout 200 scripts/micro/mfma_peak.out 2.0 > $OUT/mfma_peak.json 2>&1; cat $OUT/mfma_peak.json
fi
date
