"""the unbounded model's training step (4096 rays x (128 + 128) samples) a few times: target of rocprofv3 --kernel-trace --stats.
usage: prof_train360.py [fp32|bf16|bf16_graph ...]   (bf16 = through autograd with torch Adam; bf16_graph = mipnerf_train_step + device-side Adam from one
captured hipGraph; default: all three, as bench.py's fp32.unbounded.train sub-record runs them)"""
import os
import sys
import types

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench  # noqa: E402

args = types.SimpleNamespace(steps=20)
e = types.SimpleNamespace(dev=torch.device("cuda:0"), rank=0, world=1)
print(bench.run_train_unbounded(args, e, which=tuple(sys.argv[1:]) or ("fp32", "bf16", "bf16_graph")))
