#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
A=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/libmipnerf_hip.so
B=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/libmipnerf_hip_prev.so
for round in 1 2 3; do for so in $A $B; do
  echo -n "$(basename $so): "; MIPNERF_LIB=$so timeout 200 python bench.py --mode train --steps 80 --warmup 5 --no-cpu-baseline --preheat-seconds 1.5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r03t_train_small_kernels_ab.log
