"""Does the 256 MB Infinity Cache keep the dgrad output for a wgrad launched right after it?  Times k_mlp_wgrad on a slab
of M samples (a) straight after trainfwd + dgrad of the same slab, (b) after 3 GB of unrelated writes."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from mipnerf_pl_amd import MipNerf, _lib as L
dev = torch.device("cuda", 0)
lib = L.lib()
for M in (8192, 16384, 32768, 65536, 131072):
    N = 128; B = M // N
    model = MipNerf(num_samples=N, precision="bf16").to(dev)
    nctx = model.mlp.native(dev); h = nctx.handle
    st = torch.cuda.current_stream().cuda_stream
    enc = (torch.rand(B, N, 96, device=dev) * 2 - 1).to(torch.bfloat16)
    venc = torch.zeros(B, 32, device=dev, dtype=torch.bfloat16)
    d_raw = torch.randn(B, N, 4, device=dev) * 1e-3
    sz = nctx.train_sizes(M)
    act = torch.empty(sz[0], dtype=torch.uint8, device=dev); masks = torch.empty(sz[1], dtype=torch.uint8, device=dev)
    delta = torch.empty(sz[2], dtype=torch.uint8, device=dev); part = torch.empty(sz[3], dtype=torch.uint8, device=dev)
    raw = torch.empty(B, N, 4, device=dev); rgbs = torch.empty_like(raw)
    junk = torch.empty(3 * 1024 ** 3 // 4, device=dev)
    def fwd(): L.check(lib.mipnerf_mlp_forward_train(h, M, N, enc.data_ptr(), venc.data_ptr(), rgbs.data_ptr(), raw.data_ptr(), act.data_ptr(), masks.data_ptr(), st))
    def dg(): L.check(lib.mipnerf_mlp_dgrad(h, M, d_raw.data_ptr(), masks.data_ptr(), delta.data_ptr(), st))
    def wg(): L.check(lib.mipnerf_mlp_wgrad(h, M, act.data_ptr(), delta.data_ptr(), part.data_ptr(), None, 0, st))
    res = {}
    for mode in ("hot", "cold"):
        ts = []
        for _ in range(6):
            fwd(); dg()
            if mode == "cold": junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); wg(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = sorted(ts)[len(ts) // 2]
    gb = (M / 32) * 157 * 2048 / 1e9
    print(json.dumps(dict(M=M, operand_GB=round(gb, 3), wgrad_hot_ms=round(res["hot"], 4), wgrad_cold_ms=round(res["cold"], 4),
                          hot_TBps=round(gb / res["hot"], 2), cold_TBps=round(gb / res["cold"], 2))), flush=True)
