"""the unbounded model's bf16 training step through autograd (4096 rays x (128 + 128) samples) a few times: target of rocprofv3 --kernel-trace --stats"""
import os
import sys
import types

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench  # noqa: E402

args = types.SimpleNamespace(steps=20)
e = types.SimpleNamespace(dev=torch.device("cuda:0"), rank=0, world=1)
print(bench.run_train_unbounded(args, e))
