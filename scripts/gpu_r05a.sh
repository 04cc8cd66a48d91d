#!/bin/bash
# round 5, first GPU call: the whole parity suite on the round's changes (per-case bf16 bounds, GraphedForward, DPP wave scans, interleaved
# CDF searches, 8-rank shared-GPU bench), smoke, the default bench line, kernel stats, and two A/Bs (DPP vs shuffle scans; graph vs eager)
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-r05a}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee gpurun_out/${T}_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -k "not trained_unbounded_field" 2>&1 | tail -40 > gpurun_out/${T}_pytest_gpu_tail.txt
tail -15 gpurun_out/${T}_pytest_gpu_tail.txt
cp gpurun_out/parity.jsonl gpurun_out/${T}_parity.jsonl
for i in 1 2 3; do
  for lib in libmipnerf_hip.so libmipnerf_hip_shfl.so; do
    echo -n "$lib: "; MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 120 python scripts/prof_fwd.py --iters 400 --heat 2 --graph 2>/dev/null | tail -2 | tr '\n' ' '; echo
  done
done | tee gpurun_out/${T}_dpp_graph_ab.txt
timeout 900 python bench.py > gpurun_out/${T}_bench.out 2> gpurun_out/${T}_bench.err
tail -c 600 gpurun_out/${T}_bench.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
lines = open(f"gpurun_out/{T}_bench.out").read().splitlines()
l = json.loads(lines[-1])
open(f"gpurun_out/{T}_bench.json", "w").write(lines[-1] + "\n")
print("headline", l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["launch_ms"], "eager", l["config"].get("eager_ms_per_step"), "graph", l["config"].get("hip_graph"),
      "sustained", l["sustained"]["ms_per_step"])
print("train", l["train"]["ms_per_step"], "render", l["render"]["ms_per_step"], "fp32", l["fp32"]["ms_per_step"], l["fp32"]["roofline"]["frac"])
print("trained_field", json.dumps(l.get("trained_field"))[:900])
print("unbounded bf16", l["fp32"]["unbounded"].get("bf16"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 20 --ceiling-seconds 0 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b
head -8 $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-150
