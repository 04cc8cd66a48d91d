#!/bin/bash
# round 3, first GPU call: the whole parity suite (new: full-size training goldens, non-contiguous rays, 1-rank RCCL collective
# path), the default bench line (new: ceiling, fp32, truthful hip_graph), and the CU->CU hand-off probe (milestone 1).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r03a_pytest.txt
tail -5 gpurun_out/r03a_pytest.txt
timeout 120 python scripts/handoff_probe.py --out gpurun_out/r03a_handoff_probe.jsonl > gpurun_out/r03a_handoff_probe.log 2>&1
tail -3 gpurun_out/r03a_handoff_probe.log
timeout 600 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -c 1500 gpurun_out/r03a_bench.json
