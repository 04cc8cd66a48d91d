#!/bin/bash
# round 4: default bench line with the register-resident fp32 kernel (configs[3] + unbounded model), old kernel for comparison
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err
tail -2 gpurun_out/r04f_bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r04f_bench.json"))
print("headline", l["ms_per_step"], l["roofline"]["frac"], "train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"])
print("fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"], l["fp32"]["roofline"]["launch_ms"], "unbounded", l["fp32"]["unbounded"])
print("ceiling", {k: v for k, v in l["ceiling"].items() if k not in ("variants",)})
PY
