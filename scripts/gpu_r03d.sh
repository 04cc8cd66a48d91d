#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unbounded.py tests/test_gpu_reference_parity.py -m gpu -q -k "unbounded or full_size_training" 2>&1 | tail -150 > gpurun_out/r03d_pytest.txt
tail -8 gpurun_out/r03d_pytest.txt
