#!/bin/bash
# round 6: upper bound of what a ONE-kernel bf16 inference of MipNerf(unbounded=True) could save.  Timing builds (WRONG results):
#   libmipnerf_hip_xpre.so    k_pre_gemm computes its two outputs and does not store them           (MLP_PRE_ABLATE_STORES=1)
#   libmipnerf_hip_xtrunk.so  the trunk takes pre_x / pre_acc from LDS instead of loading them       (MLP_TRUNK_ABLATE_PRELOADS=1)
#   libmipnerf_hip_xboth.so   both: the 6.4 GB per level of hand-off never touch HBM, every MFMA, LDS read and weight DMA of both kernels still runs
# A fused kernel does all the work of xboth (and reads the encoding twice, as k_pre_gemm does) -- it cannot be faster than xboth.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp MIPNERF_ALLOW_EXPERIMENT_LIB=1
OUT=gpurun_out/${TAG:-r06}_fused360_probe.txt
: > $OUT
for i in 1 2 3; do
  for lib in libmipnerf_hip.so libmipnerf_hip_xpre.so libmipnerf_hip_xtrunk.so libmipnerf_hip_xboth.so; do
    echo -n "$lib: " >> $OUT; MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 120 python scripts/micro/prof_unbounded.py bf16 30 2>/dev/null | tail -1 >> $OUT
  done
done
cd /tmp
for lib in libmipnerf_hip.so libmipnerf_hip_xboth.so; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_u
  MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_u -o u -- python $GRAFT_REPO_ROOT/scripts/micro/prof_unbounded.py bf16 20 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_u -name "*kernel_stats.csv" | head -1)
  echo "== $lib per-kernel (rocprofv3 --kernel-trace --stats)" >> $GRAFT_REPO_ROOT/$OUT
  [ -n "$f" ] && head -6 $f | python -c "
import csv, sys
for r in csv.DictReader(sys.stdin):
    print('  %-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))" >> $GRAFT_REPO_ROOT/$OUT
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_u
cat $GRAFT_REPO_ROOT/$OUT
