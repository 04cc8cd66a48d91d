#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python scripts/quality_native.py --seeds 4 > gpurun_out/r03i_quality_native.log 2>&1
timeout 900 python scripts/quality_native.py --seeds 2 --precision fp32 >> gpurun_out/r03i_quality_native.log 2>&1
grep "^{" gpurun_out/r03i_quality_native.log
