#!/bin/bash
# A/B of builds of the library on the fp32 configs[3] forward (alternating): ab_f32r.sh "<lib> <lib> ..." [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS=$1; N=${2:-2}
for i in $(seq $N); do
  for lib in $LIBS; do
    MIPNERF_LIB=$GRAFT_REPO_ROOT/mipnerf_pl_amd/csrc/$lib timeout 200 python bench.py --mode inference --precision fp32 --rays 8192 --samples 256 --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')][-1]; l = json.loads(l)
print('$lib', l['ms_per_step'], l['roofline']['launch_ms'], l['roofline']['frac'])"
  done
done
