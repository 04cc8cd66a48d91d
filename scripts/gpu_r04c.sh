#!/bin/bash
# round 4: the register-resident fp32 kernel (k_mlp_f32r): parity suite + configs[3]-shaped fp32 bench, new vs old kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04c_pytest.txt
tail -25 gpurun_out/r04c_pytest.txt
timeout 300 python bench.py --mode inference --precision fp32 --rays 8192 --samples 256 --steps 10 --warmup 3 --no-cpu-baseline --ceiling-seconds 0.3 > gpurun_out/r04c_bench_fp32_c4.json 2> gpurun_out/r04c_bench_fp32_c4.err
tail -3 gpurun_out/r04c_bench_fp32_c4.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r04c_bench_fp32_c4.json"))
print("fp32 8192x256:", l["ms_per_step"], l["roofline"])
PY
