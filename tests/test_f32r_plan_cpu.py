"""CPU: the plan of the register-resident fp32 MLP kernel (mipnerf_pl_amd/mlp_f32r_plan.py, csrc/gen_mlp_f32r.py).

The kernel text is generated from this plan and the weight stream / aux table are packed through its index tables, so what is checked
here is the whole dataflow short of the hardware: the numpy emulation of one wavefront (chunk order, k maps of register and natural
blocks, accumulator-init images, per-lane-half thin-head weights) must reproduce the oracle's MLP (models/mip_nerf.py:75-111) for every
architecture variant the kernel is generated for; the static schedule of ring groups and natural blocks must be hazard-free."""
import os
import sys

import numpy as np
import pytest

import synthetic_inputs as syn
from mipnerf_pl_amd.mlp_f32r_plan import GROUP_CHUNKS, MAGIC, NSLOT, RING_SLOTS, F32RPlan, emulate_wave, supported
from mipnerf_pl_amd.mlp_plan import DLAYOUT, NATURAL
from oracle import mipnerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mipnerf_pl_amd", "csrc"))
from gen_mlp_bf16 import VARIANTS  # noqa: E402

SUPPORTED = [vi for vi, a in enumerate(VARIANTS) if supported(a)]


def _params(arch, seed=3):
    shapes = dict(net_depth=arch.net_depth, net_width=arch.net_width, net_depth_condition=arch.net_depth_condition,
                  net_width_condition=arch.net_width_condition, skip_index=arch.skip_index, xyz_dim=arch.xyz_dim)
    params = syn.make_params(seed=seed, density_gain=5.0, **shapes)
    flat = np.concatenate([params[n].ravel() for n, _ in arch.param_shapes()])
    return params, flat


def test_which_variants_get_the_kernel():
    assert SUPPORTED == [vi for vi, a in enumerate(VARIANTS) if a.net_width <= 256 and a.net_width_condition <= 256]
    assert 0 in SUPPORTED and len(SUPPORTED) >= 5
    with pytest.raises(NotImplementedError):
        F32RPlan.build([a for a in VARIANTS if not supported(a)][0])


@pytest.mark.parametrize("vi", SUPPORTED)
def test_emulated_wave_equals_oracle(vi):
    arch = VARIANTS[vi]
    p = F32RPlan.build(arch)
    params, flat = _params(arch)
    rng = np.random.default_rng(vi)
    enc = rng.uniform(-1, 1, (32, arch.xyz_dim)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (32, 27)).astype(np.float32)
    view = np.zeros((32, 32), np.float32)
    view[:, :27] = v27
    rgb, dens = emulate_wave(p, flat, enc, view)
    rr, dd = orc.mlp_forward(params, enc[:, None, :], v27 if arch.use_viewdirs else None, skip_index=arch.skip_index,
                             net_depth=arch.net_depth, net_depth_condition=arch.net_depth_condition)
    np.testing.assert_allclose(rgb, rr[:, 0], atol=3e-6)
    np.testing.assert_allclose(dens, dd[:, 0, 0], atol=2e-5)


@pytest.mark.parametrize("vi", SUPPORTED)
def test_tables_cover_every_parameter_exactly_once(vi):
    arch = VARIANTS[vi]
    p = F32RPlan.build(arch)
    _, total = p.param_offsets()
    seen = np.zeros(total, np.int32)
    for tab in (p.pack_table().ravel(), p.aux_table()[0].ravel()):
        np.add.at(seen, tab[tab >= 0], 1)
    # the two aux halves hold different accumulator rows / different features of the thin heads' rows; only the thin heads' biases
    # are duplicated in both halves
    aux1 = p.aux_table()[1].ravel()
    lay, H = p.aux_layout()
    dup = np.zeros(H, bool)
    for k, o in lay.items():
        if k[0] == "thin_b":
            dup[o:o + 4] = True
    np.add.at(seen, aux1[~dup & (aux1 >= 0)], 1)
    used = np.ones(total, bool)
    if not arch.use_viewdirs:          # extra_layer / view_layers are unused parameters without view directions (mip_nerf.py:99-110)
        offs, _ = p.param_offsets()
        names = [n for n, _ in arch.param_shapes()]
        for i, n in enumerate(names):
            if n.startswith("extra_layer") or n.startswith("view_layers"):
                used[offs[i]:(offs[i + 1] if i + 1 < len(offs) else total)] = False
    assert np.all(seen[used] == 1), (np.count_nonzero(seen[used] != 1), total)
    assert np.all(seen[~used] == 0)


@pytest.mark.parametrize("vi", SUPPORTED)
def test_ring_groups_and_natural_block_schedule(vi):
    p = F32RPlan.build(VARIANTS[vi])
    groups = p.groups()
    assert len(groups) % RING_SLOTS == 0 and RING_SLOTS == 2           # group g lives in slot g % RING_SLOTS, cyclic over tiles
    assert groups[0][0] == 0 and sum(k for _, k in groups) == p.n_real_chunks
    assert all(c0 + k == groups[i + 1][0] for i, (c0, k) in enumerate(groups[:-1]))
    assert all(k <= GROUP_CHUNKS and k % 4 == 0 for _, k in groups)
    # no k-step straddles a group (its chunks are read with offsets inside one ring slot)
    starts = {c0 for c0, _ in groups}
    for oi, op in enumerate(p.ops):
        for ks in range(op.nk):
            c = op.chunk0 + ks * op.quads
            for q in range(1, op.quads):
                assert c + q not in starts
    sched = p.natural_schedule()
    uses = p.natural_uses()
    assert len(sched) == len(uses)
    n_nat = sum(1 for op in p.ops for b in op.blocks if b.kind == NATURAL)
    assert len(uses) == n_nat
    for u in sched:
        assert 0 <= u["slot"] < NSLOT
        assert u["gi"] == u["g_first"] - 2 and u["issue_group"] == u["gi"] % len(groups) and u["prev_tile"] == (u["gi"] < 0)
    # every register block is read from the set the previous MFMA op wrote
    last_out = None
    for op in p.ops:
        for b in op.blocks:
            if b.kind == DLAYOUT:
                assert b.src == last_out
        if op.tiles:
            last_out = op.out


def test_blob_header():
    p = F32RPlan.build(VARIANTS[0])
    blob = np.frombuffer(p.blob(), np.int32)
    assert blob[0] == MAGIC and blob[1] == p.n_real_chunks == 2384 and blob[4] == p.n_real_chunks * 256
    assert blob[3] == 2 * p.aux_layout()[1] and blob[5] == p.param_offsets()[1] and blob[6] == p.n_groups == 76
    assert blob.size == 16 + blob[4] + blob[3]
    # 9,536 MFMAs per wave and tile = exactly the algorithmic 610,304 MAC per sample: the 640 MACs of the thin heads run on
    # the VALU, the view encoding's 27 -> 32 zero padding adds 5 x 128 = 640 to the view layer
    mfmas = sum(len(op.tiles) * op.nk for op in p.ops)
    assert mfmas == 9536 and mfmas * 2048 // 32 == 610304


# ---- replay of the GENERATED tile body (no GPU): register hazards, accumulator images, LDS-DMA bookkeeping ------------------------
import re  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "mipnerf_pl_amd", "csrc"))
import gen_mlp_f32r as gen  # noqa: E402


def _tile_body(vi):
    g = gen.Gen(F32RPlan.build(VARIANTS[vi]), vi)
    src = g.source()
    body = src[src.index("// ---- generated tile body ----"):src.index("// ---- thin heads:")]
    prologue = src[src.index("// prologue:"):src.index("for (; tile < ntiles;")]
    return g, body.splitlines(), prologue.splitlines()


@pytest.mark.parametrize("vi", SUPPORTED)
def test_generated_body_replays_without_hazards(vi):
    """Walk the generated statements in program order, as one wave executes them:
    * every MFMA accumulates into its op's output set, with the A fragment register its k-step's chunk was read into and the B register
      its k-step's operand was written to; every chunk of the stream is read exactly once per tile, from the ring slot its group lives in;
    * a register of a D tile is read as B operand only AFTER the last MFMA of the op that produced it, and an accumulator image (BIAS)
      overwrites a tile only after its last read and before the first MFMA of the op that accumulates into it next;
    * between two ring barriers exactly the LDS-DMA pieces of one ring group (+ 4 per natural block fetched) are issued, every piece
      behind the barrier that frees its slot."""
    g, lines, prologue = _tile_body(vi)
    p = g.p
    groups = p.groups()
    NG = len(groups)
    cur_op, op_of_name = None, {op.name: (oi, op) for oi, op in enumerate(p.ops)}
    last_mfma_into = {}          # (set, tile) -> line number of the last MFMA writing it (current value)
    value_op = {}                # (set, tile) -> op index that produced / is producing the value
    last_read = {}               # (set, tile) -> line number of the last B-operand read
    reads_lda = []
    breg_src = {}                # b0 / b1 -> ("reg", set, tile, r) | ("nat",)
    n_mfma = 0
    dma_ring, dma_nat, begins = 0, 0, 0
    per_interval = []
    for ln, s_ in enumerate(lines):
        s_ = s_.strip()
        m = re.match(r"// (\w+) k-step (\d+) ", s_)
        if m:
            cur_op = op_of_name[m.group(1)]
            continue
        if s_.startswith("GROUP_BEGIN("):
            per_interval.append([0, 0])
            begins += 1
        for m in re.finditer(r"LDA\((\d+)\)", s_):
            reads_lda.append(int(m.group(1)))
        for m in re.finditer(r"(b[01]) = (relu1\()?([XY])\[(\d+)\]\[(\d+)\]", s_):
            key = (m.group(3), int(m.group(4)))
            # the value must be complete: produced by an EARLIER op than the one that reads it
            assert key in value_op and value_op[key] < cur_op[0], (ln, s_)
            assert bool(m.group(2)) == p.ops[value_op[key]].relu, (ln, s_)
            last_read[key] = ln
            breg_src[m.group(1)] = ("reg",) + key
        for m in re.finditer(r"(b[01]) = bq\[", s_):
            breg_src[m.group(1)] = ("nat",)
        m = re.match(r"BIAS\(([XY])\[(\d+)\], (\d+)\);", s_)
        if m:
            key = (m.group(1), int(m.group(2)))
            # nothing may still need the old value: every reader of it lies behind us
            assert last_read.get(key, -1) < ln
            value_op[key] = "bias"
            last_mfma_into[key] = None
        m = re.match(r"MFMA\(([XY])\[(\d+)\], a([01])\[(\d)\], (b[01])\);", s_)
        if m:
            n_mfma += 1
            oi, op = cur_op
            key = (m.group(1), int(m.group(2)))
            assert m.group(1) == op.out and int(m.group(2)) == 4 * int(m.group(3)) + int(m.group(4))
            if value_op.get(key) != oi:
                assert value_op.get(key) == "bias", ("first MFMA of an op into a tile that holds no accumulator image", ln, s_)
                value_op[key] = oi
            last_mfma_into[key] = ln
        if "dma_piece<" in s_:
            dma_ring += 1
            per_interval[-1][0] += 1
        if "dma_piece_v<" in s_:
            dma_nat += 1
            per_interval[-1][1] += 1
    assert n_mfma == sum(len(op.tiles) * op.nk for op in p.ops)
    assert begins == NG
    # every chunk of the stream read exactly once per tile (offsets inside a slot), the first k-step's by the prologue / the previous tile
    want = []
    for gi, (c0, k) in enumerate(groups):
        want += [(gi % RING_SLOTS) * GROUP_CHUNKS * 1024 + i * 1024 for i in range(k)]
    assert sorted(reads_lda) == sorted(want)
    # LDS-DMA bookkeeping: one ring group per barrier interval (cyclically), four pieces per natural block fetched
    assert dma_ring == sum(k // 4 for _, k in groups) and dma_nat == 4 * sum(u["fetch"] for u in p.natural_schedule())
    assert any("GROUP_BEGIN(0);" in x for x in prologue)
