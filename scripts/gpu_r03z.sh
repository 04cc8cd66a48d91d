#!/bin/bash
# round 3: frame time vs chunk size (configs[4])
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 8192 32768 131072 640000 8192; do
  timeout 300 python bench.py --mode render --steps 8 --warmup 2 --render-chunk $c --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readline()); print('chunk', $c, 'ms/frame', l['ms_per_step'], 'frac', l['roofline']['frac'])"
done | tee gpurun_out/r03z_render_chunks.txt
