#!/bin/bash
# round 4, end-of-round verification as the driver runs it: smoke, the whole parity suite (-x), the default bench line, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r04q}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/${T}_smoke.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/${T}_pytest_gpu_tail.txt
tail -3 gpurun_out/${T}_pytest_gpu_tail.txt
cp gpurun_out/parity.jsonl gpurun_out/${T}_parity.jsonl
timeout 900 python bench.py > gpurun_out/${T}_bench.out 2> gpurun_out/${T}_bench.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
lines = open(f"gpurun_out/{T}_bench.out").read().splitlines()
print("stdout lines:", len(lines), "| last line is the JSON:", lines[-1].startswith("{"))
l = json.loads(lines[-1])
open(f"gpurun_out/{T}_bench.json", "w").write(lines[-1] + "\n")
print("headline", l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["launch_ms"], "sustained", l["sustained"]["ms_per_step"])
print("train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"], "fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"],
      "unbounded", l["fp32"]["unbounded"]["ms_per_step"], l["fp32"]["unbounded"]["frac"], "unbounded bf16", l["fp32"]["unbounded"].get("bf16"))
print("ceiling", l["ceiling"]["register_fed"], l["ceiling"]["lds_fed"], l["ceiling"]["lds_and_dma_fed"], l["ceiling"]["fp32_register_fed"]["frac_of_peak"],
      "cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["cores"], "train cpu", l["train"]["cpu_baseline"]["value"])
print("scale_model", l["scale_model"]["measured_inputs"], l["scale_model"]["predicted"]["8"])
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 20 --ceiling-seconds 0 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b
head -3 $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-150
