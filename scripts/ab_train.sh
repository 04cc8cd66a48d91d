#!/bin/bash
# alternating A/B of library builds on the bf16 training step: ab_train.sh "<lib> <lib> ..." [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS=$1; N=${2:-3}
for i in $(seq $N); do
  for lib in $LIBS; do
    MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 200 python bench.py --mode train --steps 50 --warmup 5 --no-cpu-baseline --preheat-seconds 2 2>gpurun_out/ab_train.err | python -c "
import sys, json
ls = [x for x in sys.stdin if x.startswith('{')]
if not ls: print('$lib', 'no line (non-finite loss?)')
else:
    l = json.loads(ls[-1]); print('$lib', l['ms_per_step'])"
  done
done
