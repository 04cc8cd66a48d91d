#!/bin/bash
# bf16 training-step check: train tests, graphed bench, eager rocprofv3 kernel stats.   usage: gpu_bf16_train.sh TAG
set -u
TAG=${1:-trn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python bench.py --mode train --no-cpu-baseline --steps 50 --warmup 10 > $OUT/bench_train.json 2>$OUT/bench.err; cut -c1-200 $OUT/bench_train.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --mode train --steps 10 --warmup 2 --no-graph --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/train_kernel_stats.csv
rm -rf $OUT/prof
python - $OUT/train_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
