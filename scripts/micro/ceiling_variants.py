import ctypes as C, torch, sys
sys.path.insert(0, '.')
from mipnerf_pl_amd import _lib as L
torch.cuda.init()
st = torch.cuda.current_stream().cuda_stream
for rnd in (1,):
  for lds in (0, 1, 2, 3, 10, 2, 3, 10, 1, 0):
    r = (C.c_double * 3)()
    L.diag_check(L.diag_lib().mipnerf_mfma_ceiling(lds, 2, rnd, 2.0, r, st), "c")
    print("lds", lds, "random", rnd, "TF/s", round(r[0], 1), "frac", round(r[0] / (157.3 if lds == 10 else 2500), 4), "ms", round(r[1], 4), "GHz", round(r[2], 3), flush=True)
