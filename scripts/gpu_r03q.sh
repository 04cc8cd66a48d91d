#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r03q_bench_$1.json 2>/dev/null
python - $1 <<'PY'
import json, sys
l = json.load(open(f"gpurun_out/r03q_bench_{sys.argv[1]}.json"))
print("headline", l["ms_per_step"], l["roofline"]["frac"], "sustained", l["sustained"]["ms_per_step"], "train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"], "fp32", l["fp32"]["roofline"]["frac"], "ceiling", l["ceiling"]["register_fed"], l["ceiling"]["lds_fed"], "of_ceiling", l["roofline"]["frac_of_measured_lds_fed_ceiling"])
PY
