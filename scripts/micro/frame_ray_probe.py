"""Where the fp32 whole-frame error against the reference's frame comes from: device-generated rays (fp32) vs the reference's float64 rays."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import gpu_util as G
import test_gpu_frame as T
from mipnerf_pl_amd import Rays
g = G.load_golden("frame_c5_800x800")
size, focal = int(g["cfg_size"]), float(g["focal"])
R64 = T.rendergen_rays_f64(g["pose"], focal, size)
from mipnerf_pl_amd.datasets import RenderGen
ds = RenderGen(focal, [size, size], scales=1, device=torch.device("cuda:0"))
rd = ds[int(g["cfg_pose"])]
for k in Rays._fields:
    a, b = getattr(rd, k).cpu().numpy().astype(np.float64), getattr(R64, k).astype(np.float64)
    print(k, "max abs", np.abs(a - b).max(), "max rel", (np.abs(a - b) / np.maximum(np.abs(b), 1e-30)).max())
for tag, rays in (("device rays", None), ("reference-exact rays", R64)):
    c, f, vm, d, acc = T.render_reference_frame(G, g, "fp32", rays_np=rays)
    print(tag, "coarse", G.maxdiff(c, g["coarse_rgb"]), "fine", G.maxdiff(f, g["fine_rgb"]), "dist", G.maxdiff(d[0], g["distance"]), "acc", G.maxdiff(acc.reshape(size, size), g["acc"]))
    e = np.abs(f.cpu().numpy() - g["fine_rgb"]).max(-1).ravel()
    print("   fine rgb error quantiles 50/99/99.99/max:", np.quantile(e, [0.5, 0.99, 0.9999, 1.0]), "pixels > 5e-5:", int((e > 5e-5).sum()))
    e = np.abs(d[0].cpu().numpy() - g["distance"]).ravel()
    print("   distance error quantiles:", np.quantile(e, [0.5, 0.99, 0.9999, 1.0]), "> 2e-4:", int((e > 2e-4).sum()))
