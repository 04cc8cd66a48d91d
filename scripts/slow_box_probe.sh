#!/bin/bash
# round 4: when the box runs k_mlp_f32r slowly (launch > 18.5 ms at configs[3]), say why: clock / MFMA-busy / waits from a PMC pass, and the
# LDS-resident k_mlp_f32 and the in-process fp32 ceiling on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
L=$(timeout 200 python bench.py --mode inference --precision fp32 --rays 8192 --samples 256 --steps 6 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --preheat-seconds 1 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print(l['roofline']['launch_ms'])")
echo "launch_ms $L"
if python -c "import sys; sys.exit(0 if float('$L') > 18.5 else 1)"; then
  echo "SLOW BOX"
  F32_CYCLES_OUT=r04s_slow_box_cycles.txt bash scripts/pmc_f32_cycles.sh 2>&1 | tail -1
  python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from mipnerf_pl_amd import _lib as L
D = L.diag_lib()
r = (C.c_double * 3)()
L.diag_check(D.mipnerf_mfma_ceiling(10, 2, 1, 1.5, r, torch.cuda.current_stream().cuda_stream), "c")
print("fp32 register-fed ceiling", r[0], "TF", r[2], "GHz")
r = (C.c_double * 3)()
L.diag_check(D.mipnerf_mfma_ceiling(0, 2, 1, 1.5, r, torch.cuda.current_stream().cuda_stream), "c")
print("bf16 register-fed ceiling", r[0], "TF", r[2], "GHz")
PY
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | head -30
fi
