#!/bin/bash
# round 4: k_mlp_f32r A/B iteration: fp32 parity tests + configs[3] bench + cycle counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_unbounded.py tests/test_gpu_stages.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_reference_parity.py -x -q -m gpu -k "every_ray" 2>&1 | tail -2
timeout 300 python bench.py --mode all --no-cpu-baseline --ceiling-seconds 0 --steps 20 > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r04i_bench.json").readline())
print("fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"], l["fp32"]["roofline"]["launch_ms"], "unbounded", l["fp32"]["unbounded"]["ms_per_step"], l["fp32"]["unbounded"]["frac"])
PY
F32_CYCLES_OUT=r04i_f32r_cycles.txt bash scripts/pmc_f32_cycles.sh 2>&1 | tail -2
