#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_train.py tests/test_gpu_reference_parity.py tests/test_gpu_unbounded.py -m gpu -q -k "fp32 or variant or unbounded or wide or two_view" 2>&1 | tail -8
timeout 200 python bench.py --mode all --no-cpu-baseline --steps 10 --preheat-seconds 0.5 --ceiling-seconds 0 --sustain-seconds 0 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', l['fp32']['ms_per_step'], l['fp32']['roofline']['frac'])"
