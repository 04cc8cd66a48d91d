#!/bin/bash
# round 3: LDS ring depth of the weight-gradient kernel (3 / 4 / 5 stages), alternating
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do for lib in hip st3 st5; do
  echo -n "$lib "; MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/libmipnerf_$lib.so timeout 200 python scripts/prof_train.py --iters 20 2>&1 | grep wgrad_ms | python -c "
import sys, json
l = json.loads(sys.stdin.readline()); print(round(l['wgrad_ms'], 4), round(l['wgrad_noreduce_ms'], 4))"
done; done | tee gpurun_out/r03ab_wgrad_stages.txt
