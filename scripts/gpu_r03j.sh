#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_quality.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r03j_pytest.txt
tail -5 gpurun_out/r03j_pytest.txt; grep quality gpurun_out/parity.jsonl | tail -1
