#!/bin/bash
# round 3, third GPU call: whole parity suite (new: unbounded model, quality smoke, fixed tolerances)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -120 > gpurun_out/r03c_pytest.txt
tail -8 gpurun_out/r03c_pytest.txt
