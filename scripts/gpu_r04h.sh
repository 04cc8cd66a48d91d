#!/bin/bash
# round 4: PMC traffic passes (-> profiles/mlp_pmc.json), the new 1-rank RCCL tests, the default bench line with scale_model
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
ROUND=4 bash scripts/pmc_traffic.sh 2>&1 | tail -12
cp gpurun_out/pmc_traffic.json profiles/mlp_pmc.json
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r04h_bench.json 2> gpurun_out/r04h_bench.err
tail -2 gpurun_out/r04h_bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r04h_bench.json"))
print("headline", l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["traffic"], l["roofline"]["traffic_source"])
print("train", l["train"]["ms_per_step"], l["train"]["roofline"]["traffic"], l["train"].get("cpu_baseline"))
print("scale_model", json.dumps(l.get("scale_model"))[:1500])
PY
