#!/bin/bash
# round 3: full parity suite + default bench line + rocprofv3 kernel stats of the headline forward (fused small kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -100 > gpurun_out/r03e_pytest.txt
tail -6 gpurun_out/r03e_pytest.txt
timeout 600 python bench.py > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r03e_bench.json"))
print("headline", l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["launch_ms"], "sustained", l["sustained"]["ms_per_step"], l["sustained"].get("frac"))
print("train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"], "fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"], "ceiling", l["ceiling"]["register_fed"], l["ceiling"]["lds_fed"])
PY
cd /tmp
rm -rf /tmp/prof_r03e
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03e -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode inference --steps 50 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --ceiling-seconds 0 --preheat-seconds 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r03e_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03e_prof.err
f=$(find /tmp/prof_r03e -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03e_bench_kernel_stats.csv && head -8 "$f" | cut -c1-160
