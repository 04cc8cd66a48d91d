#!/bin/bash
# round 3, final state: smoke, the whole parity suite (as the driver runs it: -x), the default bench line, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/r03y_smoke.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r03y_pytest.txt
tail -4 gpurun_out/r03y_pytest.txt
timeout 600 python bench.py > gpurun_out/r03y_bench.json 2> gpurun_out/r03y_bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r03y_bench.json"))
print("headline", l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["launch_ms"], "sustained", l["sustained"]["ms_per_step"], l["sustained"].get("frac"))
print("train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"], "fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"], "ceiling", l["ceiling"]["register_fed"], l["ceiling"]["lds_fed"], "cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["cores"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03y_bench_steps20.json 2>/dev/null
python -c "
import json; l=json.load(open('gpurun_out/r03y_bench_steps20.json')); print('steps20', l['ms_per_step'], l['roofline']['frac'], l['train']['ms_per_step'])"
