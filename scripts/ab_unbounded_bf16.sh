#!/bin/bash
# A/B of library builds on the bf16 forward of the unbounded-scene model (alternating): ab_unbounded_bf16.sh "<lib> <lib> ..." [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS=$1; N=${2:-2}
for i in $(seq $N); do
  for lib in $LIBS; do
    echo -n "$lib: "; MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 120 python scripts/micro/prof_unbounded.py bf16 30 2>/dev/null | tail -1
  done
done
