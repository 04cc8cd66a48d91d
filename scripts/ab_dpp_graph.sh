#!/bin/bash
# round 5: headline forward, DPP wave scans (product) vs the shuffle build (MLP_WAVE_DPP=0 MIPNERF_LIB_NAME=libmipnerf_hip_shfl.so python -m
# mipnerf_pl_amd.build), each as eager launches and as one hipGraph replay (model.GraphedForward), alternating; then rocprofv3 kernel stats of
# the bench's inference mode for both -> profiles/r05a_dpp_graph_ab.txt, r05b_inference_*_kernel_stats.csv
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r05}
C=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc
for i in 1 2 3; do
  for lib in libmipnerf_hip.so libmipnerf_hip_shfl.so; do
    echo -n "$lib: "; MIPNERF_LIB=$C/$lib timeout 120 python scripts/prof_fwd.py --iters 400 --heat 2 --graph 2>/dev/null | tail -2 | tr '\n' ' '; echo
  done
done | tee gpurun_out/${T}_dpp_graph_ab.txt
cd /tmp
for lib in libmipnerf_hip.so libmipnerf_hip_shfl.so; do
  MIPNERF_LIB=$C/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -o bench -- python $GRAFT_REPO_ROOT/bench.py --mode inference --no-cpu-baseline --sustain-seconds 0 --steps 20 --ceiling-seconds 0 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_inference_${lib%.so}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b
done
