#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_forward.py tests/test_gpu_unbounded.py -m gpu -q -k "hip_graph or render_image or graphed_frame or unbounded_model_trains or full_frame" 2>&1 | tail -6
for round in 1 2; do for lanes in 1 2 3; do
  echo -n "lanes $lanes: "; MIPNERF_FRAME_LANES=$lanes timeout 300 python bench.py --mode render --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['roofline']['frac'])"
done; done 2>&1 | tee gpurun_out/r03l_render_lanes.log
