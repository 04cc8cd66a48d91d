#!/bin/bash
# alternating A/B of library builds on the headline forward (4096 x (128 + 128), bf16, fused IPE): ab_headline.sh "<lib> <lib> ..." [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS=$1; N=${2:-3}
for i in $(seq $N); do
  for lib in $LIBS; do
    echo -n "$lib: "; MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 120 python scripts/prof_fwd.py --iters 400 --heat 2 2>/dev/null | tail -1
  done
done
