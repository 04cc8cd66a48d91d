"""CPU: the plan of the bf16 training kernels (mipnerf_pl_amd/mlp_train_plan.py).  The numpy emulation moves data
exactly like the generated forward-with-save / dgrad kernels and the table-driven wgrad kernel (same pack tables,
T-block layout, ReLU bit packing, partial -> parameter index table); in fp32 it must reproduce the oracle's
gradients, which are pinned against the reference's autograd (tests/golden/mlp_bwd_8x32_trained.npz)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mipnerf_pl_amd.mlp_train_plan import (JOB_FLOATS, TrainPlan, colfeat, emulate_train, frag_sample, pack_mask,
                                           unpack_mask)
from oracle import mipnerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tp():
    return TrainPlan.build()


def _case(golden_dir, S):
    g = dict(np.load(os.path.join(golden_dir, "mlp_bwd_8x32_trained.npz")))
    params = orc.make_params(seed=int(g["param_seed"]), density_gain=float(g["density_gain"]))
    enc = g["enc"].reshape(-1, 96)[:S]
    venc = np.repeat(g["venc"], int(g["num_samples"]), axis=0)[:S]
    view = np.zeros((S, 32), np.float32)
    view[:, :27] = venc
    d_raw = np.concatenate([g["d_rgb"].reshape(-1, 3), g["d_den"].reshape(-1, 1)], -1)[:S]
    return params, enc, venc, view, d_raw


def test_layout_maps_are_bijections(tp):
    for kind in (0, 1):
        assert sorted(colfeat(kind, n) for n in range(32)) == list(range(32))
    assert sorted(frag_sample(f, hi, j) for f in range(2) for hi in range(2) for j in range(8)) == list(range(32))
    assert tp.NH == 72 and tp.NG == 69 and tp.NMASK == 9
    assert len(tp.bchunks) % 64 == 0 and tp.n_bchunks_real == 1100
    rng = np.random.default_rng(0)
    r0, r1 = rng.normal(size=(64, 8)).astype(np.float32), rng.normal(size=(64, 8)).astype(np.float32)
    words = np.zeros((64, 4), np.uint32)
    for t in range(8):
        words[:, t >> 1] |= pack_mask(r0 * (t + 1), r1 - t) << np.uint32(8 * (t & 1))
    for t in range(8):
        assert (unpack_mask(words, t) == (np.concatenate([r0 * (t + 1), r1 - t], 1) > 0)).all()


def test_every_parameter_has_exactly_one_partial(tp):
    ot = tp.wgrad_out_table()
    assert ot.shape[1:] == (8, 9, 64, 16) and ot[0].size == JOB_FLOATS
    idx = ot[ot >= 0]
    offs, total = tp.fwd.param_offsets()
    assert total == 612740 and idx.size == np.unique(idx).size and idx.max() == total + tp.n_scratch - 1
    # parameters produced by the chain-rule post-processing instead of a partial: extra_layer.{weight,bias},
    # view_layers.0.0.weight[:, :256], view_layers.0.0.bias -- everything else (+ the whole scratch region) exactly once
    po = tp.post
    post = np.zeros(total, bool)
    post[offs[po["extra_w"]]:offs[po["extra_w"]] + 256 * 256] = True
    post[offs[po["extra_b"]]:offs[po["extra_b"]] + 256] = True
    vw = np.zeros((128, 283), bool)
    vw[:, :256] = True
    post[offs[po["view_w"]]:offs[po["view_w"]] + 128 * 283] = vw.ravel()
    post[offs[po["view_b"]]:offs[po["view_b"]] + 128] = True
    covered = np.zeros(total + tp.n_scratch, bool)
    covered[idx] = True
    assert covered[total:].all() and np.array_equal(covered[:total], ~post)
    # the dgrad stream uses every weight that has a downstream gradient exactly once: all but layer 0, the 96
    # skip columns of layer 5 and the 27 view columns of the view layer; no bias
    bp = tp.bpack_table()
    used = bp[bp >= 0]
    assert used.size == np.unique(used).size == 557696   # 557,696 dgrad MACs per sample


def test_emulated_dataflow_reproduces_oracle_gradients(tp, golden_dir):
    S = 70     # 3 wave tiles, the last one ragged (clamped duplicates must not contribute)
    params, enc, venc, view, d_raw = _case(golden_dir, S)
    flatp = np.concatenate([v.ravel() for v in params.values()])
    flat, seen, raw = emulate_train(tp, flatp, enc, view, d_raw)
    assert seen.max() == 1 and flat.size == 612740
    rr, dd = orc.mlp_forward(params, enc[:, None, :], venc)
    np.testing.assert_allclose(raw[:, :3], rr[:, 0], atol=1e-5)
    np.testing.assert_allclose(raw[:, 3], dd[:, 0, 0], atol=1e-4)
    og = orc.mlp_backward(params, enc[:, None, :], venc, d_raw[:, None, :3], d_raw[:, None, 3:])
    off = 0
    for k, v in og.items():
        e = np.abs(flat[off:off + v.size] - v.ravel()).max() / max(1e-20, np.abs(v).max())
        off += v.size
        assert e < 1e-5, (k, e)


def test_embedded_tables_equal_python_plan(tp):
    from mipnerf_pl_amd import _lib as L
    h = L.lib()
    for which, ref in ((3, tp.bpack_table()), (4, tp.wgrad_out_table()), (5, tp.job_table())):
        n = h.mipnerf_debug_table(which, None, 0)
        buf = np.zeros(n, np.int32)
        h.mipnerf_debug_table(which, buf.ctypes.data, n)
        assert n == ref.size and np.array_equal(buf, ref.ravel()), which


VARIANT_CASES = [(1, dict(net_width=128, net_width_condition=128), (40, 37, 9, 296)),
                 (2, dict(net_width_condition=256, use_viewdirs=False), (67, 65, 8, 912)),
                 (3, dict(net_depth=6, skip_index=3), None),
                 (5, dict(net_depth_condition=2), (76, 73, 10, None))]        # round 5: two view layers (one more saved activation set / delta set / mask row)


@pytest.mark.parametrize("vi,arch_kw,shape", VARIANT_CASES)
def test_variant_plan_emulates_and_is_embedded(vi, arch_kw, shape):
    """The training plan is parametric in the architecture: the 128-wide variant and the one without view directions (colour
    head on the trunk output, no bottleneck chain-rule step, unused extra_layer / view_layers) reproduce the oracle's gradients
    through the same emulated dataflow, and their three tables are the ones linked into the library."""
    from mipnerf_pl_amd import _lib as L
    from mipnerf_pl_amd.mlp_plan import Arch
    tpv = TrainPlan.build(Arch(**arch_kw))
    assert shape is None or (tpv.NH, tpv.NG, tpv.NMASK) == shape[:3] and shape[3] in (None, tpv.n_bchunks_real)
    views = arch_kw.get("use_viewdirs", True)
    S = 70
    rng = np.random.default_rng(5)
    params = orc.make_params(seed=3, density_gain=2.0, **{k: v for k, v in arch_kw.items() if k != "use_viewdirs"})
    enc = (rng.normal(size=(S, 96)) * 0.5).astype(np.float32)
    venc = rng.normal(size=(S, 27)).astype(np.float32)
    view = np.zeros((S, 32), np.float32)
    view[:, :27] = venc
    d_raw = rng.normal(size=(S, 4)).astype(np.float32)
    flat, seen, raw = emulate_train(tpv, np.concatenate([v.ravel() for v in params.values()]), enc, view, d_raw)
    assert seen.max() == 1
    og = orc.mlp_backward(params, enc[:, None, :], venc if views else None, d_raw[:, None, :3], d_raw[:, None, 3:],
                          skip_index=arch_kw.get("skip_index", 4), net_depth=arch_kw.get("net_depth", 8),
                          net_depth_condition=arch_kw.get("net_depth_condition", 1))
    off = 0
    for k, v in og.items():
        got = flat[off:off + v.size]
        off += v.size
        if not np.any(v):
            assert not np.any(got), k            # unused parameters of MLP.forward(x, None)
            continue
        e = np.abs(got - v.ravel()).max() / np.abs(v).max()
        assert e < 1e-5, (k, e)
    h = L.lib()
    for which, ref in ((3, tpv.bpack_table()), (4, tpv.wgrad_out_table()), (5, tpv.job_table())):
        n = h.mipnerf_debug_table_variant(vi, which, None, 0)
        buf = np.zeros(n, np.int32)
        h.mipnerf_debug_table_variant(vi, which, buf.ctypes.data, n)
        assert n == ref.size and np.array_equal(buf, ref.ravel()), which


def test_generator_schedule_passes_hazard_check(tmp_path):
    """gen_mlp_train.py replays both tile programs and asserts the register-set discipline (every B operand of op i
    was written by op i-1, every transposed register by the op that stores it)."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "mipnerf_pl_amd", "csrc", "gen_mlp_train.py"), str(tmp_path)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    for f in ("mlp_bf16_trainfwd_gen.hip", "mlp_bf16_dgrad_gen.hip", "mlp_bf16_trainfwd_gen_v1.hip", "mlp_bf16_dgrad_gen_v1.hip",
              "mlp_bf16_trainfwd_gen_v2.hip", "mlp_bf16_dgrad_gen_v2.hip", "mlp_bf16_trainfwd_gen_v3.hip", "mlp_bf16_dgrad_gen_v3.hip",
              "mlp_bf16_trainfwd_gen_v5.hip", "mlp_bf16_dgrad_gen_v5.hip"):
        gen = open(os.path.join(tmp_path, f)).read()
        committed = open(os.path.join(REPO, "mipnerf_pl_amd", "csrc", f)).read()
        assert gen == committed, f"{f} is stale: re-run python -m mipnerf_pl_amd.build"


@pytest.mark.parametrize("arch_kw", [dict(net_width=192, net_width_condition=64), dict(net_width=64, net_width_condition=64)])
def test_generators_schedule_narrow_layers_without_hazards(arch_kw):
    """Shapes outside VARIANTS whose layers have a single two-tile panel (64-wide view layer / trunk): the consumer's first slot
    needs the producer's activations, so the producer's epilogue must not be spread into the consumer's panel.  The inference
    generator used to emit such a kernel silently (bf16 rgb off by 0.2 on an ad-hoc 8 x 192 / 64 build); both generators now place
    register writes in front of their first reader and replay the schedule with a hazard check."""
    sys.path.insert(0, os.path.join(REPO, "mipnerf_pl_amd", "csrc"))
    import gen_mlp_bf16 as GI
    import gen_mlp_train as GT
    from mipnerf_pl_amd.mlp_plan import Arch, Plan
    a = Arch(**arch_kw)
    src = GI.gen_kernel(Plan.build(a), 9)
    assert "namespace v9" in src and "epilogue_half" in src
    tpv = TrainPlan.build(a)
    assert "launch_mlp_bf16_trainfwd_v9" in GT.gen_trainfwd(tpv, 9) and "launch_mlp_bf16_dgrad_v9" in GT.gen_dgrad(tpv, 9)


def test_pre_gemm_training_plan_emulation_matches_oracle():
    """round 5: the training form of the two-kernel bf16 MLP (TrainPlan.build(arch, pre_gemm=True): k_pre_gemm + trunk forward-with-save
    starting from the preloaded register set, the standard dgrad stream, weight-gradient jobs over 32-feature column blocks of the
    encoding) emulated in numpy == the oracle's autograd gradients for the 672-wide unbounded-scene architecture; every parameter is
    fed by at most one partial position, the trunk kernel's generated schedule passes the hazard replay (gen_mlp_train asserts it)."""
    import synthetic_inputs as syn
    from mipnerf_pl_amd.mlp_plan import Arch
    from mipnerf_pl_amd.mlp_train_plan import TrainPlan, emulate_train
    arch = Arch(xyz_dim=672, feat_per_deg=42, bf16_kernels=False)
    tp = TrainPlan.build(arch, pre_gemm=True)
    assert tp.pre_gemm and tp.NE == 21 and "enc" not in tp.h_blocks and tp.NH == 69
    enc_jobs = [j for j in tp.jobs if j.b_src == 1]
    assert [len(j.b_blocks) for j in enc_jobs] == [8, 8, 5, 8, 8, 5] and sum(j.biasmap is not None for j in enc_jobs) == 1
    params = syn.make_params(seed=5, density_gain=6.0, xyz_dim=672)
    names = [n for n, _ in arch.param_shapes()]
    flat = np.concatenate([params[n].ravel() for n in names])
    rng = np.random.default_rng(0)
    S = 45                                       # 1.4 wave tiles: the padded samples carry zero deltas
    enc = rng.uniform(-1, 1, (S, 672)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (S, 27)).astype(np.float32)
    view = np.zeros((S, 32), np.float32)
    view[:, :27] = v27
    d_raw = rng.normal(0, 1e-2, (S, 4)).astype(np.float32)
    g, seen, raw = emulate_train(tp, flat, enc, view, d_raw)
    _, nparams = tp.fwd.param_offsets()
    assert int(seen[:nparams].max()) == 1
    rr, dd = orc.mlp_forward(params, enc[:, None, :], v27)
    assert np.abs(raw[:, :3] - rr[:, 0]).max() <= 5e-6 and np.abs(raw[:, 3] - dd[:, 0, 0]).max() <= 2e-5
    og = orc.mlp_backward(params, enc[:, None, :], v27, d_raw[:, None, :3], d_raw[:, None, 3:])
    offs, _ = tp.fwd.param_offsets()
    for i, n in enumerate(names):
        a = g[offs[i]:offs[i] + params[n].size].reshape(params[n].shape)
        assert np.abs(a - og[n]).max() <= 5e-6 * max(np.abs(og[n]).max(), 1e-20), n
    # the job table tells the kernel which jobs read the encoding
    jt = tp.job_table()
    assert [int(r[3]) for r in jt] == [j.b_src for j in tp.jobs]


def test_wgrad_fragment_gather_builds_the_row_major_lds_image():
    """k_mlp_wgrad, encoding jobs on FRAGMENT encodings (kernels_wgrad.hip, wgrad_body<2>): every lane of the two LDS-DMAs of a 32-feature
    block fetches the 16-byte piece that belongs at its place of the row-major [32 samples][64 B] image the transposing reads expect
    (landed lane-linearly the pieces sit multiples of 256 B apart: 4-way bank conflicts).  The address arithmetic of the source, restated:
    fragment layout = [k-step][lane half][sample][8 features] (kernels_360.hip enc360_index), LDS destination = DMA base + lane * 16."""
    src = open(os.path.join(REPO, "mipnerf_pl_amd", "csrc", "kernels_wgrad.hip")).read()
    assert "const char* p0 = base + ((lane >> 1) & 1) * 1024 + (lane & 1) * 512 + (lane >> 2) * 16;" in src
    assert "const char* p1 = p0 + 256;" in src
    frag = np.zeros(2048, np.int32)                       # one block = two k-steps; value = sample * 100 + feature, per bf16 element (2 bytes)
    for ks in range(2):
        for hi in range(2):
            for n in range(32):
                for j in range(8):
                    frag[(ks * 1024 + hi * 512 + n * 16 + j * 2) // 2 * 2] = n * 100 + ks * 16 + hi * 8 + j
    lds = np.full(2048, -1, np.int32)
    for d in range(2):                                    # the second DMA: source + 256 bytes, LDS destination + 1024 bytes
        for lane in range(64):
            p = ((lane >> 1) & 1) * 1024 + (lane & 1) * 512 + (lane >> 2) * 16 + d * 256
            dst = d * 1024 + lane * 16
            lds[dst:dst + 16] = frag[p:p + 16]
    for s in range(32):
        for f in range(32):
            assert lds[s * 64 + f * 2] == s * 100 + f, (s, f)


def test_wgrad_transposing_reads_deliver_the_plans_operand():
    """The operand reads of the encoding jobs (kernels_wgrad.hip lds_frag_enc): two ds_read_b64_tr_b16 per fragment from the row-major
    [32 samples][64 B] LDS image.  Instruction semantics as probed on MI355X (profiles/r02n_ds_read_tr_probe.txt): within a group of 16 lanes,
    lane i receives element (i & 3) of the 8-byte granules addressed by lanes (i >> 2) + 4 k, k = 0 .. 3.  With the source's per-lane address
    every lane (hi, n) must end up with feature column n of the samples Plan.drow(hi, 8 f + j), j = 0 .. 7 -- the B operand the plan's
    emulation feeds the weight-gradient MFMA (emulate_train_tile: ET[b, f, (hi, n), j])."""
    src = open(os.path.join(REPO, "mipnerf_pl_amd", "csrc", "kernels_wgrad.hip")).read()
    assert "const unsigned enc_lane_off = (unsigned)((4 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (4 * ((lane >> 4) & 1) + (lane & 3)) * 8);" in src
    assert "const char* p = blk + enc_lane_off + f * (16 * 64);" in src and "(p + 8 * 64)" in src
    from mipnerf_pl_amd.mlp_plan import Plan
    image = lambda byte: (byte // 64, (byte % 64) // 2)          # LDS byte -> (sample, feature) of the row-major image
    for f in range(2):
        got = {}
        for r in range(2):
            addr = [(4 * (L_ >> 5) + ((L_ & 15) >> 2)) * 64 + (4 * ((L_ >> 4) & 1) + (L_ & 3)) * 8 + f * 1024 + r * 512 for L_ in range(64)]
            for lane in range(64):
                grp, i = lane & ~15, lane & 15
                for k in range(4):
                    src_lane = grp + (i >> 2) + 4 * k
                    got[(lane, 4 * r + k)] = image(addr[src_lane] + 2 * (i & 3))
        for lane in range(64):
            hi, n = lane >> 5, lane & 31
            for j in range(8):
                assert got[(lane, j)] == (Plan.drow(hi, 8 * f + j), n), (f, lane, j, got[(lane, j)])
