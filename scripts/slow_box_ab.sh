#!/bin/bash
# on a slow box (k_mlp_f32r launch > 18.5 ms at configs[3]) run the A/B given in $1; otherwise just report the launch time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L=$(timeout 200 python bench.py --mode inference --precision fp32 --rays 8192 --samples 256 --steps 6 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print(l['roofline']['launch_ms'])")
echo "launch_ms $L"
if python -c "import sys; sys.exit(0 if float('$L') > 18.5 else 1)"; then
  echo "SLOW BOX"
  bash scripts/ab_f32r.sh "$1" 2
fi
