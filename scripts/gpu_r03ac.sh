#!/bin/bash
# round 3: k_mlp_f32 with the wide (672) encoding streamed from global memory, 64-sample tiles: parity tests, then the fp32 record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unbounded.py tests/test_gpu_forward.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --mode all --steps 30 --warmup 3 --no-cpu-baseline --ceiling-seconds 0 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readline()); f = l['fp32']; print('fp32 bounded', f['ms_per_step'], f['roofline']['launch_ms'], f['roofline']['frac'], 'unbounded', f['unbounded']['ms_per_step'], f['unbounded']['launch_ms'], f['unbounded']['frac'])"
done | tee gpurun_out/r03ac_fp32_stream_enc.txt
