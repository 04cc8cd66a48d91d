#!/bin/bash
# round 3: weight-gradient workgroups per job ~ blocks + 4 (instead of equal): training tests, then three alternating A/B pairs
# of the training step against the previous library (mipnerf_pl_amd/csrc/libmipnerf_prev.so = the commit before, equal split)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_parity.py tests/test_gpu_forward.py -x -q -m gpu -k "train or grad or variants or native or adam" 2>&1 | tail -5
for i in 1 2 3; do
  for lib in prev hip; do
    MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/libmipnerf_$lib.so timeout 300 python bench.py --mode train --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readline()); print('$lib', l['ms_per_step'], l['roofline']['frac'])"
  done
done | tee gpurun_out/r03aa_wgrad_merge_ab.log
