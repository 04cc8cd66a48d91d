#!/bin/bash
# round 4, first GPU call: the go / no-go probe of the register-resident fp32 kernel + the parity suite on the round's first changes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 scripts/micro/f32r_probe.out 2.0 2>&1 | tee gpurun_out/r04a_f32r_probe.txt
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r04a_pytest.txt
tail -4 gpurun_out/r04a_pytest.txt
