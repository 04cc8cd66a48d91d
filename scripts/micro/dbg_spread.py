import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import gpu_util as G
from test_gpu_train import _mlp_case, _run_native_mlp
from oracle import mipnerf_oracle as orc
from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train
B, N = 8, 32
params, enc, venc, d_raw = _mlp_case(B, N, seed=B * 100 + N)
raw, grads, enc_bf, v_bf = _run_native_mlp(G, params, enc, venc, d_raw)
tp = TrainPlan.build()
flatp = np.concatenate([v.ravel() for v in params.values()])
S = B * N
flat, seen, raw_em = emulate_train(tp, flatp, enc_bf.reshape(S, 96), np.repeat(v_bf, N, axis=0), d_raw.reshape(S, 4), round_bf16=True)
print("raw err", np.abs(raw.reshape(S, 4) - raw_em).max())
off = 0
for k, v in params.items():
    g = grads[k].ravel().astype(np.float64); em = flat[off:off + v.size].astype(np.float64); off += v.size
    print(f"{k:28s} rel {np.linalg.norm(g - em) / max(np.linalg.norm(em), 1e-30):.3e}")
