#!/bin/bash
# round 6: alternating A/B of builds of the 512-wide trunk's bf16 kernel (libmipnerf_hip_<tag>.so built with MLP_WIDE_* knobs, see
# gen_mlp_bf16.gen_kernel), three rounds; then the cycle counters of each build (own --pmc pass)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
LOG=$OUT/${TAG:-r06}_w512_ab.txt
: > $LOG
for round in 1 2 3; do
  for so in mipnerf_pl_amd/csrc/libmipnerf_hip*.so; do
    echo -n "$(basename $so): " >> $LOG
    MIPNERF_LIB=$ROOT/$so python scripts/micro/time_variant_mlp.py w512 2>&1 | tr '\n' ' ' | cut -c1-260 >> $LOG
    echo >> $LOG
  done
done
cat $LOG
