#!/bin/bash
# round 3, second GPU call: the whole parity suite without -x (every failure at once)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/r03b_pytest.txt
tail -8 gpurun_out/r03b_pytest.txt
