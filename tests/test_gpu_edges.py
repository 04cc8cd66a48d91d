"""Edge sizes of the hot path (the tier's "empty and ragged inputs, maximum sizes"): zero rays, and a batch whose per-sample
buffers pass 4 GiB (every sample / byte offset in the kernels has to be 64-bit).  Ragged sizes (rays not a multiple of the
256-sample tile, N not a multiple of 64) are covered next to the stages they concern (test_gpu_forward.py: fwd_ragged_100x128,
test_gpu_train.py: ragged training shapes, ragged last chunk of a frame)."""
import numpy as np
import pytest
import torch

import synthetic_inputs as si
from oracle import mipnerf_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    assert torch.cuda.is_available()
    return gpu_util


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_zero_rays_return_empty_levels_like_the_reference(G, precision):
    """mip_nerf.py:172-248 on [0, k] rays runs through torch's empty-tensor ops and returns empty per-level tuples; so does the
    native module (inference and under autograd, where backward leaves zero gradients on every parameter)."""
    from mipnerf_pl_amd import Rays
    model = G.make_model(orc.make_params(seed=3), 64, precision)
    rays = G.to_dev(si.synthetic_rays(8, seed=1))
    empty = Rays(*[x[:0] for x in rays])
    with torch.no_grad():
        ret = model(empty, False, True)
    assert len(ret) == 2
    for lvl in ret:
        assert [tuple(t.shape) for t in lvl] == [(0, 3), (0,), (0,), (0, 64), (0, 65)]
        assert all(t.dtype == torch.float32 and t.is_cuda for t in lvl)
    ret = model(empty, True, True)
    loss = sum(t.sum() for lvl in ret for t in lvl[:4])
    loss.backward()
    for n, p in model.named_parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0, n
    # a partition of fewer rays than ranks hands some rank an empty shard; its render is empty, the gather still has every ray
    from mipnerf_pl_amd.parallel import shard_bounds, shard_rays
    three = Rays(*[x[:3] for x in rays])
    pieces = []
    with torch.no_grad():
        whole = model(three, False, True)[1][0]
        for r in range(4):
            sh = shard_rays(three, r, 4)
            assert sh.origins.shape[0] == shard_bounds(3, r, 4)[1] - shard_bounds(3, r, 4)[0]
            pieces.append(model(sh, False, True)[1][0])
    assert pieces[3].shape == (0, 3)
    assert torch.equal(torch.cat(pieces), whole)


@pytest.mark.parametrize("fused_ipe", [1, 0])
def test_buffers_beyond_4_gib_are_indexed_in_64_bit(G, fused_ipe):
    """2,200,037 rays x 128 samples = 281.6 M samples per level: the (r,g,b,sigma) buffer is 4.5 GB, the stand-alone encoding
    buffer (fused_ipe = 0) 54 GB, sample indices pass 2^28 and byte offsets 2^32.  The rays are a 4,099-ray set repeated, so the
    expected value of ray i is the small forward's ray i mod 4099 -- bit for bit (a sample's arithmetic does not depend on where
    its tile lies) -- on EVERY ray, also around the 2^32-byte boundary and in the ragged last tile."""
    P, B, N = 4099, 2_200_037, 128
    model = G.make_model(orc.make_params(seed=5), N, "bf16")
    base = G.to_dev(si.synthetic_rays(P, seed=11, multiscale=True))
    idx = torch.arange(B, device=G.DEV) % P
    from mipnerf_pl_amd import Rays
    big = Rays(*[x[idx].contiguous() for x in base])
    ctx = model.mlp.native(torch.device(G.DEV))
    ctx.set_option(3, fused_ipe)
    try:
        with torch.no_grad():
            small = model(base, False, True)
            out = model(big, False, True)
            torch.cuda.synchronize()
            bad = {}
            for lvl in range(2):
                for nm, a, b in zip(G.NAMES, out[lvl], small[lvl]):
                    ne = (a != b[idx]).reshape(B, -1).any(1)
                    if bool(ne.any()):
                        first = int(torch.nonzero(ne)[0])
                        bad[f"l{lvl}_{nm}"] = (int(ne.sum()), first, float((a - b[idx]).abs().max()))
        G.record(f"beyond_4gib fused_ipe={fused_ipe}", rays=B, samples_per_level=B * N, mismatching_fields=len(bad))
        assert not bad, bad
        assert bool(torch.isfinite(out[1][0]).all())
    finally:
        ctx.set_option(3, 1)
        del out, big
        ctx._ws = None
        torch.cuda.empty_cache()
