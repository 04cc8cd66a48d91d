"""debug: saved activations / masks / deltas of the unbounded model's bf16 training kernels against the numpy emulation, tile by tile"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import synthetic_inputs as syn  # noqa: E402
from mipnerf_pl_amd import MipNerf, _lib as L, ops  # noqa: E402
from mipnerf_pl_amd.mlp_plan import Arch  # noqa: E402
from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train_tile  # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
params = syn.make_params(seed=41, density_gain=6.0, xyz_dim=672)
m = MipNerf(num_samples=N, unbounded=True, precision="bf16")
m.load_state_dict({"mlp." + k: torch.from_numpy(v.copy()) for k, v in params.items()})
m = m.to(dev)
rng = np.random.default_rng(B * 100 + N)
enc = rng.uniform(-1, 1, (B, N, 672)).astype(np.float32)
v27 = rng.uniform(-1, 1, (B, 27)).astype(np.float32)
v32 = np.zeros((B, 32), np.float32)
v32[:, :27] = v27
d_raw = np.concatenate([rng.normal(0, 1e-2, (B, N, 3)), rng.normal(0, 1e-3, (B, N, 1))], -1).astype(np.float32)
e16 = torch.from_numpy(enc).to(dev).to(torch.bfloat16).contiguous()
v16 = torch.from_numpy(v32).to(dev).to(torch.bfloat16).contiguous()
nctx = m.mlp.native(torch.device(dev))
M = B * N
sz = nctx.train_sizes(M)
act = torch.zeros(sz[0], dtype=torch.uint8, device=dev)
masks = torch.zeros(sz[1], dtype=torch.uint8, device=dev)
delta = torch.zeros(sz[2], dtype=torch.uint8, device=dev)
raw = torch.empty(B, N, 4, device=dev)
rs = torch.empty_like(raw)
L.check(L.lib().mipnerf_mlp_forward_train(nctx.handle, M, N, e16.data_ptr(), v16.data_ptr(), rs.data_ptr(), raw.data_ptr(), act.data_ptr(),
                                          masks.data_ptr(), ops._stream()), "fwd")
dr = torch.from_numpy(d_raw).to(dev).contiguous()
L.check(L.lib().mipnerf_mlp_dgrad(nctx.handle, M, dr.data_ptr(), masks.data_ptr(), delta.data_ptr(), ops._stream()), "dgrad")
torch.cuda.synchronize()
arch = Arch(xyz_dim=672, feat_per_deg=42, bf16_kernels=False)
tp = TrainPlan.build(arch, pre_gemm=True)
names = [n for n, _ in arch.param_shapes()]
flat = np.concatenate([params[n].ravel() for n in names])
n_wt = (M + 31) // 32
HT_g = act[:((M + 255) // 256) * 8 * tp.NH * 2048].view(torch.bfloat16).float().cpu().numpy().reshape(-1, tp.NH, 2, 64, 8)
GT_g = delta.view(torch.bfloat16).float().cpu().numpy().reshape(-1, tp.NG, 2, 64, 8)
MK_g = masks.cpu().numpy().view(np.uint32).reshape(-1, tp.NMASK, 64, 4)
encf = e16.float().cpu().numpy().reshape(-1, 672)
viewf = np.repeat(v16.float().cpu().numpy(), N, axis=0)
inv_h = {v[0] + i: f"{k}[{i}]" for k, v in tp.h_blocks.items() for i in range(v[1])}
inv_g = {v[0] + i: f"{k}[{i}]" for k, v in tp.g_blocks.items() for i in range(v[1])}
for t in range(n_wt):
    idx = np.minimum(np.arange(t * 32, t * 32 + 32), M - 1)
    valid = np.arange(t * 32, t * 32 + 32) < M
    HT, GT, raw_e, ET, MK = emulate_train_tile(tp, flat, encf[idx], viewf[idx], d_raw.reshape(-1, 4)[idx], valid, True, return_masks=True)
    mk_diff = [int(np.unpackbits((MK_g[t, l] ^ MK[l]).view(np.uint8)).sum()) for l in range(tp.NMASK)]
    h_bad = [(inv_h[b], float(np.abs(HT_g[t, b] - HT[b]).max())) for b in range(tp.NH) if np.abs(HT_g[t, b] - HT[b]).max() > 0]
    g_bad = [(inv_g[b], float(np.abs(GT_g[t, b] - GT[b]).max() / max(np.abs(GT[b]).max(), 1e-30))) for b in range(tp.NG)
             if np.abs(GT_g[t, b] - GT[b]).max() > 1e-3 * max(np.abs(GT[b]).max(), 1e-30)]
    print(f"tile {t}: mask bit flips per layer {mk_diff}; HT blocks that differ: {len(h_bad)} first {h_bad[:4]}; GT blocks > 1e-3 rel: {len(g_bad)} first {g_bad[:4]}")
