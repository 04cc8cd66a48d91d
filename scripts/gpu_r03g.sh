#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 200 python scripts/handoff_probe.py --quick --out gpurun_out/r03g_handoff_probe.jsonl > gpurun_out/r03g_handoff_probe.log 2>&1
tail -2 gpurun_out/r03g_handoff_probe.log | cut -c1-400
