#!/bin/bash
# round 3: edge sizes (zero rays, > 4 GiB buffers) and widths between the generated shapes (zero-padded on the containing one)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r03v_edges.txt
timeout 900 python -m pytest tests/test_gpu_forward.py -q -m gpu -k "constructor_variants" 2>&1 | tail -25 | tee gpurun_out/r03v_variants.txt
cp gpurun_out/parity.jsonl gpurun_out/r03v_parity.jsonl
