#!/bin/bash
# round 6: alternating A/B of builds of the one-kernel unbounded bf16 forward (libmipnerf_hip_<tag>.so built with MLP_FUSED_* knobs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r06}_fused360_ab.txt
: > $OUT
for i in 1 2 3; do
  for so in mipnerf_pl_amd/csrc/libmipnerf_hip*.so; do
    echo -n "$(basename $so): " >> $OUT; MIPNERF_LIB=$GRAFT_REPO_ROOT/$so timeout 120 python scripts/micro/prof_unbounded.py bf16 30 2>/dev/null | tail -1 >> $OUT
  done
done
cat $OUT
