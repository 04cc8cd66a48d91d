"""The reference's multi-GPU training is Lightning's DDPPlugin (train.py:56-60) = torch's DistributedDataParallel around the LightningModule.
`MipNeRFSystem.training_step` hands ordinary per-parameter gradients to autograd (one node, MipNerf.loss_native), so DDP's reducer hooks must
fire and average them -- executed here with two ranks on the one GPU of the box (gloo: RCCL refuses two ranks per device), each rank on its OWN
ray batch as Lightning's DistributedSampler gives it: after `loss.backward()` every rank holds the MEAN of the two ranks' local gradients, and a
`torch.optim.Adam` step keeps the replicas bit-identical."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def test_ddp_averages_the_gradients_of_the_routed_training_step():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "tests", "_ddp_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    import re
    recs = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", out.stdout)]           # (the two ranks' lines may arrive glued together)
    assert sorted(r["rank"] for r in recs) == [0, 1]
    for r in recs:
        # DDP's bucket all-reduce sums in another order than (a + b) / 2: fp32 round-off only
        assert r["grad_rel_err"] <= 1e-6 and r["replicas_equal"] and r["moved"], r
    assert recs[0]["loss"] != recs[1]["loss"]          # the two ranks really saw different batches
