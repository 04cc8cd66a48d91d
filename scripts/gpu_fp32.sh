#!/bin/bash
# fp32 parity-mode check: the GPU tests that touch the fp32 training path, the fp32 training bench and its rocprofv3 kernel stats.
# usage: gpu_fp32.sh TAG   (outputs in gpurun_out/TAG/)
set -u
TAG=${1:-fp32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "fp32 or f32 or variant or trajectory or grad" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python bench.py --precision fp32 --mode train --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_fp32_train.json 2>$OUT/bench.err; cut -c1-200 $OUT/bench_fp32_train.json
timeout 300 python bench.py --precision fp32 --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 10 > $OUT/bench_fp32.json 2>/dev/null; cut -c1-120 $OUT/bench_fp32.json; python - $OUT/bench_fp32.json <<'PY'
import json,sys; r=json.load(open(sys.argv[1])); print('fp32 inference roofline', r['roofline']['frac'], r['roofline']['launch_ms'])
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --precision fp32 --mode train --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/fp32_train_kernel_stats.csv && head -8 $OUT/fp32_train_kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof
