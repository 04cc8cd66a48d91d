#!/bin/bash
# Interleaved A/B timing of MLP-kernel variants (libmipnerf_hip_<tag>.so built with MLP_* env knobs).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
LOG=$OUT/ab_mlp.log
: > $LOG
for round in 1 2 3; do
  for so in mipnerf_pl_amd/csrc/libmipnerf_hip*.so; do
    tag=$(basename $so .so)
    echo -n "round $round $tag: " >> $LOG
    MIPNERF_LIB=$ROOT/$so python scripts/prof_mlp.py --iters 30 >> $LOG 2>&1
  done
done
cat $LOG
