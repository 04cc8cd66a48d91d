#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_stages.py tests/test_gpu_forward.py -m gpu -q 2>&1 | tail -8
for i in 1 2 3; do timeout 200 python bench.py --mode train --steps 60 --warmup 5 --no-cpu-baseline --preheat-seconds 1.5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', l['ms_per_step'])"; done
