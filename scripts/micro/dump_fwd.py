import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mipnerf_pl_amd import MipNerf, _lib as L
dev = torch.device("cuda", 0)
B, N = 8, 32
M = B * N
torch.manual_seed(0)
model = MipNerf(num_samples=N, precision="bf16").to(dev)
nctx = model.mlp.native(dev)
enc = (torch.rand(B, N, 96, device=dev) * 2 - 1).to(torch.bfloat16)
venc = torch.zeros(B, 32, device=dev); venc[:, :27] = torch.randn(B, 27, device=dev); venc = venc.to(torch.bfloat16)
sz = nctx.train_sizes(M)
act = torch.zeros(sz[0], dtype=torch.uint8, device=dev); masks = torch.zeros(sz[1], dtype=torch.uint8, device=dev)
raw = torch.empty(B, N, 4, device=dev); rgbs = torch.empty_like(raw)
L.check(L.lib().mipnerf_mlp_forward_train(nctx.handle, M, N, enc.data_ptr(), venc.data_ptr(), rgbs.data_ptr(), raw.data_ptr(),
                                          act.data_ptr(), masks.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
np.savez(sys.argv[1], act=act.cpu().numpy(), masks=masks.cpu().numpy(), raw=raw.cpu().numpy())
