#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_datasets.py -m gpu -q 2>&1 | tail -5
timeout 600 python bench.py > gpurun_out/r03n_bench.json 2> gpurun_out/r03n_bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r03n_bench.json"))
print("headline", l["value"], l["ms_per_step"], l["roofline"]["frac"], "sustained", l["sustained"]["ms_per_step"], "train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"], "fp32", l["fp32"]["roofline"]["frac"], "ceiling", l["ceiling"]["register_fed"], l["ceiling"]["lds_fed"], "cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["cores"])
PY
