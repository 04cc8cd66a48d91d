#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for round in 1 2; do for g in 256 248 224 192 128; do timeout 100 python scripts/prof_fwd.py --iters 200 --heat 1.0 --grid $g 2>/dev/null | grep forward; done; done | tee gpurun_out/r03r_grid.log
