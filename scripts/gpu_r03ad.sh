#!/bin/bash
# round 3: does the ADDRESS PATTERN of the T-block stores matter beside a busy matrix pipe?  feeding mode 3 of the in-process ceiling
# with every wave filling its own region (today's tile-major layout) vs all waves writing adjacent chunks at every step (block-major)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/c3.py <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, '.')
from mipnerf_pl_amd import _lib as L
torch.cuda.init()
st = torch.cuda.current_stream().cuda_stream
r = (C.c_double * 3)()
L.check(L.lib().mipnerf_mfma_ceiling(3, 2, 1, 2.0, r, st), "c")
print("pattern", os.environ.get("MIPNERF_CEILING_STORE_PATTERN", "0"), "frac", round(r[0] / 2500, 4), "ms", round(r[1], 4), "store TB/s", round(256 * 8 * 256 * 7 * 1024 / (r[1] * 1e-3) / 1e12, 3), flush=True)
PY
for i in 1 2; do for p in 0 1; do MIPNERF_CEILING_STORE_PATTERN=$p python /tmp/c3.py 2>&1 | grep pattern; done; done | tee gpurun_out/r03ad_store_pattern.txt
