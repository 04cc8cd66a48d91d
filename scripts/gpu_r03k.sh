#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_unbounded.py tests/test_gpu_forward.py tests/test_gpu_reference_parity.py -m gpu -q -k "unbounded or fused_small or full_size_every_ray or two_view" 2>&1 | tail -15 > gpurun_out/r03k_pytest.txt
tail -5 gpurun_out/r03k_pytest.txt
timeout 300 python bench.py --mode all --no-cpu-baseline --steps 20 > gpurun_out/r03k_bench.json 2>gpurun_out/r03k_bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r03k_bench.json"))
print("headline", l["ms_per_step"], l["roofline"]["frac"], "fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"], "train", l["train"]["ms_per_step"], l["train"]["roofline"].get("hbm_view"))
PY
