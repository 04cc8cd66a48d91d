#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_train2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train2 -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 40 --warmup 3 --no-graph --no-cpu-baseline --preheat-seconds 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r03u_train_prof.json 2>/dev/null
f=$(find /tmp/prof_train2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03u_train_kernel_stats.csv && head -14 "$f" | cut -c1-110
rm -rf /tmp/prof_fwd2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd2 -o t -- python $GRAFT_REPO_ROOT/bench.py --mode inference --steps 50 --warmup 5 --no-cpu-baseline --sustain-seconds 0 --ceiling-seconds 0 --preheat-seconds 1.0 > $GRAFT_REPO_ROOT/gpurun_out/r03u_fwd_prof.json 2>/dev/null
f=$(find /tmp/prof_fwd2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03u_bench_kernel_stats.csv && head -6 "$f" | cut -c1-110
