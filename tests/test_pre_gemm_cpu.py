"""CPU: the plan of the two-kernel bf16 MLP of the unbounded-scene model (mipnerf_pl_amd/mlp_pre_plan.py, csrc/gen_pre_gemm.py).

k_pre_gemm (layer 0 + the encoding half of the skip layer, k-step-major) and the trunk kernel (k_mlp_bf16 generated from
Plan.build(arch, pre_gemm=True)) are generated from this plan and their weight streams / bias tables packed through its index
tables.  Checked here, short of the hardware: the numpy emulation of one wavefront through BOTH kernels reproduces the oracle's MLP
(models/mip_nerf.py:75-111); every parameter lands in exactly one table slot; the generated straight-line bodies issue the MFMAs, ring
barriers and B-operand loads the counted waits assume."""
import os
import re
import sys

import numpy as np
import pytest

import synthetic_inputs as syn
from mipnerf_pl_amd.mlp_plan import Plan
from mipnerf_pl_amd.mlp_pre_plan import GROUP, MAGIC, RING_SLOTS, PrePlan, emulate_pre_gemm, emulate_pre_wave, supported
from oracle import mipnerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mipnerf_pl_amd", "csrc"))
import gen_mlp_bf16 as gb  # noqa: E402
import gen_pre_gemm as gp  # noqa: E402

VIS = [vi for vi, a in enumerate(gb.VARIANTS) if supported(a)]


def _params(arch, seed=3):
    params = syn.make_params(seed=seed, density_gain=5.0, xyz_dim=arch.xyz_dim)
    return params, np.concatenate([params[n].ravel() for n, _ in arch.param_shapes()])


def test_which_variants_get_the_two_kernel_form():
    """exactly the fp32-only variants with an encoding too wide for k_mlp_bf16 (today: the 672 off-axis features of MipNerf(unbounded=True))"""
    assert VIS == [vi for vi, a in enumerate(gb.VARIANTS) if a.xyz_dim > 96]
    assert len(VIS) == 1 and gb.VARIANTS[VIS[0]].xyz_dim == 672
    with pytest.raises(NotImplementedError):
        PrePlan.build(gb.VARIANTS[0])


@pytest.mark.parametrize("vi", VIS)
def test_emulated_wave_through_both_kernels_equals_oracle(vi):
    arch = gb.VARIANTS[vi]
    p = PrePlan.build(arch)
    params, flat = _params(arch)
    rng = np.random.default_rng(vi)
    enc = rng.uniform(-1, 1, (32, arch.xyz_dim)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (32, 27)).astype(np.float32)
    view = np.zeros((32, 32), np.float32)
    view[:, :27] = v27
    rr, dd = orc.mlp_forward(params, enc[:, None, :], v27)
    # exact arithmetic of the dataflow (no bf16 rounding): the oracle to fp32 round-off
    rgb, dens = emulate_pre_wave(p, flat, enc, view)
    np.testing.assert_allclose(rgb, rr[:, 0], atol=3e-6)
    np.testing.assert_allclose(dens, dd[:, 0, 0], atol=2e-5)
    # with the kernels' bf16 operand rounding: the bf16 tolerance of the standard model's tests (|raw| up to ~5 here)
    rgb16, dens16 = emulate_pre_wave(p, flat, enc, view, round_bf16=True)
    assert np.max(np.abs(rgb16 - rr[:, 0])) < 2e-2 and np.max(np.abs(dens16 - dd[:, 0, 0])) < 6e-2
    # what the first kernel hands over: X = bf16(relu(W0 enc + b0)), images = W5[:, 256:] enc + b5 in fp32
    x, images = emulate_pre_gemm(p, flat, enc)
    a1 = np.maximum(enc @ params["layers.0.0.weight"].T + params["layers.0.0.bias"], 0)
    p5 = enc @ params[f"layers.{p.skip_layer}.0.weight"][:, arch.net_width:].T + params[f"layers.{p.skip_layer}.0.bias"]
    for n in (0, 7, 31):
        for f in (0, 5, 100, 255):
            t = f // 32
            # feature f of sample n sits in fragment k = kmap^-1: search the dlayout
            hits = [(k, hi, j) for k in range(16) for hi in range(2) for j in range(8) if Plan.kmap(1, k, hi, j) == f]
            assert len(hits) == 1
            k, hi, j = hits[0]
            assert abs(x[k, hi * 32 + n, j] - a1[n, f]) < 2e-5
            regs = [(hi2, r) for hi2 in range(2) for r in range(16) if Plan.drow(hi2, r) == f % 32]
            hi2, r = regs[0]
            assert abs(images[t, hi2 * 32 + n, r] - p5[n, f]) < 2e-5


@pytest.mark.parametrize("vi", VIS)
def test_tables_cover_every_parameter_exactly_once(vi):
    p = PrePlan.build(gb.VARIANTS[vi])
    _, total = p.trunk.param_offsets()
    seen = np.zeros(total, np.int32)
    for tab in (p.pack_table(), p.bias_table(), p.trunk.pack_table(), p.trunk.bias_table()):
        t = tab.ravel()
        np.add.at(seen, t[t >= 0], 1)
    assert np.all(seen == 1)


@pytest.mark.parametrize("vi", VIS)
def test_streams_and_blob(vi):
    p = PrePlan.build(gb.VARIANTS[vi])
    # pre-GEMM stream: whole revolutions of the three-slot ring, no padding, k-step-major; split form: both matrices per k-step
    assert p.n_real_chunks == len(p.chunks) == 2 * p.nk * p.ntiles == 672 and len(p.chunks) % (GROUP * RING_SLOTS) == 0
    if p.split:
        assert p.chunks[:17] == [(0, 0, t) for t in range(8)] + [(1, 0, t) for t in range(8)] + [(0, 1, 0)]
    else:
        assert p.chunks[:9] == [(0, 0, t) for t in range(8)] + [(0, 1, 0)] and p.chunks[336] == (1, 0, 0)
    two = PrePlan.build(gb.VARIANTS[vi], split=not p.split)          # the other work split: the same chunks in another order
    assert sorted(two.chunks) == sorted(p.chunks) and two.chunks != p.chunks
    # trunk: the standard stream minus layer 0 (8 x 42) and the skip layer's encoding k-steps (8 x 42), one whole group of zero padding
    std = Plan.build(gb.VARIANTS[vi])
    assert p.trunk.n_real_chunks == std.n_real_chunks - 2 * 8 * 42 == 1120 and len(p.trunk.chunks) == 1152
    assert [op.name for op in p.trunk.ops][0] == "layer1" and [op.pre for op in p.trunk.ops].count(True) == 1
    assert p.trunk.ops[p.skip_layer - 1].pre and p.trunk.ops[p.skip_layer - 1].nk == 16
    blob = np.frombuffer(p.blob(), np.int32)
    assert blob[0] == MAGIC and blob[1] == blob[2] == 672 and blob[4] == 1152 and blob[5] == 1120 and blob[6] == p.trunk.n_tiles == 70
    assert blob[7] == p.trunk.param_offsets()[1] and blob[8] == 42 and blob[9] == 8
    # round 6: the one-kernel form's tables ride behind (1792 chunks: 2 x 336 k-major + the trunk's 1120, no padding; 78 tiles)
    assert blob[10] == len(p.fused.chunks) == p.fused.n_real_chunks == 1792 and blob[11] == p.fused.n_tiles == 78
    assert blob.size == 16 + 672 * 512 + blob[3] + 1152 * 512 + 70 * 32 + 1792 * 512 + 78 * 32


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("vi", VIS)
def test_generated_gemm_body(vi, split):
    """program order of the generated k_pre_gemm tile body, for both work splits (mlp_pre_plan.SPLIT): the MFMAs of a wave in stream order
    on the accumulator of their tile, with the A register of their slot and the B register of their k-step; the A fragment of every slot
    read from the ring position its chunk lives at; one B-operand load per k-step, into the register the PREVIOUS k-step read, the
    rotation tile-periodic; 21 ring barriers, each with the same number of loads since its predecessor (what the counted vmcnt relies on)"""
    p = PrePlan.build(gb.VARIANTS[vi], split=split)
    src = gp.gen_gemm(p, vi)
    depth = gp.DEPTH_SPLIT if split else gp.DEPTH
    per_wave = [c for c in p.chunks if c[0] == 0] if split else p.chunks          # split: a wave sees one matrix (role 0 here; role 1 = + 8 chunks)
    body = src[src.index("    for (;;) {"):src.index("        if (!has_next) break;")].splitlines()
    mf = [re.match(r"\s+MFMA\(acc(\d), A(\d), EB(\d+)\);", ln) for ln in body]
    mf = [m for m in mf if m]
    assert len(mf) == len(per_wave) == (336 if split else 672)
    steps_per_tile = p.nk if split else 2 * p.nk
    assert steps_per_tile % depth == 0
    for c, m in enumerate(mf):
        ps, ks, t = per_wave[c]
        assert int(m.group(1)) == t and int(m.group(2)) == c % gp.PREFETCH and int(m.group(3)) == (ps * p.nk + ks) % depth
    # every A fragment is read from where its chunk sits in the three-slot ring (the role offset of the split form is in ring_lane)
    ldas = [int(x) for x in re.findall(r"A\d = LDA\((\d+)\);", "\n".join(body))]
    want = []
    for c in range(len(per_wave)):
        gc = p.chunks.index(per_wave[(c + gp.PREFETCH) % len(per_wave)])
        want.append(((gc // GROUP) % RING_SLOTS) * GROUP * 1024 + (gc % GROUP) * 1024)
    assert ldas == want
    events = []
    for ln in body:
        if "RING_BARRIER(" in ln:
            events.append("gb")
        elif re.search(r"LOAD_B(_NT)?\(EB", ln):           # (_NT: pass 1's reads of the two-pass form, the last use of a tile's operands)
            events.append(int(re.search(r"LOAD_B(?:_NT)?\(EB(\d+)\)", ln).group(1)))
        elif re.match(r"\s+MFMA\(acc7", ln):
            events.append("k")
    assert events.count("gb") == len(p.chunks) // GROUP == 21 and events.count("k") == steps_per_tile
    gaps, n = [], 0
    for ev in events:
        if ev == "gb":
            gaps.append(n)
            n = 0
        elif ev != "k":
            n += 1
    per_gap = 2 if split else 4
    assert gaps[1:] == [per_gap] * 20 and gaps[0] in (per_gap - 1, per_gap)
    step = -1
    for ev in events:
        if ev == "k":
            step += 1
        elif ev != "gb":
            assert ev == (step - 1) % depth
    # counted waits: never more than the vector-memory operations known to be younger than the awaited group's DMA (its successor's four
    # chunks + the loads of two barrier intervals), minus the margin
    # the counts assume the successor group's DMA is in flight at EVERY barrier, so the refill may not stop on the last tile
    assert "if (has_next) issue_group" not in "\n".join(body) and 's_waitcnt vmcnt(0)' in src[src.index("if (!has_next) break;"):]
    ks_ = [int(x) for x in re.findall(r"RING_BARRIER\((\d+)\);", src)]
    assert ks_ and max(ks_) <= 4 + 2 * per_gap - gp.VM_MARGIN and min(ks_) >= 4


@pytest.mark.parametrize("vi", VIS)
def test_generated_trunk(vi):
    """the trunk kernel text: X preloaded from 16 fragments, the skip layer's eight accumulators from k_pre_gemm's images (no BIAS for
    them), one extra ring barrier for the zero-padding group, and the register hazard replay of gen_mlp_bf16 passed (it asserts)"""
    p = PrePlan.build(gb.VARIANTS[vi])
    src = gb.gen_kernel(p.trunk, vi)
    assert len(re.findall(r"X\[\d+\] = PRE_LD\(reinterpret_cast<const bf16x8\*>\(prex_lane", src)) == 16
    assert sorted(int(x) for x in re.findall(r"pre_load\(acc\d\d, pre_lane \+ (\d+)\);", src)) == [t * 4096 for t in range(8)]
    skip_first = p.trunk.ops[p.skip_layer - 1].first_tile
    bias_tiles = {int(x) for x in re.findall(r"BIAS\(acc\d\d, (\d+)\);", src)}
    assert bias_tiles == set(range(p.trunk.n_tiles)) - set(range(skip_first, skip_first + 8))
    assert src.count("GROUP_BEGIN(35, 0);") == 1 and len(re.findall(r"\n\s+MFMA\(", src)) == 1120
    assert "ipe_to_lds<" not in src[src.index("k_mlp_bf16("):]


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.parametrize("vi", VIS)
def test_compiled_gemm_kernel_keeps_the_counted_waits_honest(vi, tmp_path):
    """The ring barriers of k_pre_gemm wait with a counted `vmcnt(K)`: correct as long as at least K vector-memory operations are younger than
    the wave's DMA of the awaited group.  The B-operand loads are ordinary C++ loads the compiler schedules, so this looks at what it actually
    emitted -- the shipped object, disassembled: between any two consecutive barriers of the steady state there are exactly 4 (split form: 2)
    B-operand loads and 4 LDS-DMA chunks (=> 4 + 2 x that many younger operations; K leaves 4 of margin), nothing went to scratch, and no
    call is left in the kernel."""
    import shutil
    import subprocess
    lib = os.path.join(REPO, "mipnerf_pl_amd", "csrc", "libmipnerf_hip.so")
    if not (os.path.exists(lib) and os.path.exists(OBJDUMP)):
        pytest.skip("needs the built library (python -m mipnerf_pl_amd.build) and llvm-objdump")
    shutil.copy(lib, tmp_path / "x.so")
    subprocess.run([OBJDUMP, "--offloading", "x.so"], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = []                                                    # one code object per translation unit: the one that defines this variant's kernel
    for f in sorted(os.listdir(tmp_path)):
        syms = subprocess.run([OBJDUMP, "-t", f], cwd=tmp_path, capture_output=True, text=True).stdout if "gfx950" in f else ""
        if f"pre_v{vi}" in syms and "k_pre_gemm" in syms:      # (the trunk kernels' units mention pre_v<i> too)
            co.append(f)
    assert len(co) == 1
    dis = subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
    kernels = re.split(r"\n[0-9a-f]+ <[^>]*k_pre_gemm[^>]*>:\n", dis)[1:]
    assert len(kernels) == 2                                   # fragment / row-major encodings
    for body in kernels:
        assert "scratch_" not in body and "s_swappc" not in body
        iv = body.split("s_barrier")
        assert len(iv) == 24                                   # bias fill + prologue + 21 per tile + tail
        loads = [len(re.findall(r"global_load_dwordx4", x)) for x in iv]
        dma = [len(re.findall(r"global_load_lds_dwordx4", x)) for x in iv]
        split = PrePlan.build(gb.VARIANTS[vi]).split
        per_gap, depth = (2, gp.DEPTH_SPLIT) if split else (4, gp.DEPTH)
        assert loads[3:23] == [per_gap] * 20 and loads[2] in (per_gap - 1, per_gap) and dma[2:24] == [4] * 22 and loads[1] == depth - 1 and dma[1] == 8
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)[^\n]*\n[^\n]*s_barrier", body)      # the wait in front of every ring barrier
        assert len(waits) == 22 and set(waits) == {str(4 + 2 * per_gap - gp.VM_MARGIN)}


@pytest.mark.parametrize("vi", VIS)
def test_one_kernel_form_plan_and_generated_kernel(vi):
    """Round 6, the ONE-kernel form (Plan.build(arch, fused=True)): layer 0 and the skip layer are k-step-major ops (chunks [k-step][tile], all 8
    accumulator tiles live), the skip layer adds the encoding part FIRST -- the order of k_pre_gemm + trunk --, so the numpy emulation of the plan
    gives the two-kernel emulation's bits (fp32 and bf16 rounding); against the oracle MLP to fp32 round-off.  From the generated source: every
    encoding k-step is DMA'd into the wave-private ring exactly once per pass, FUSED_AHEAD k-steps ahead of its first MFMA and never into a slot
    whose previous k-step is still to be read; every ring read sits behind a wait that leaves at most the DMAs issued after its own in flight; the
    ring-group boundaries wait with counted vmcnt too (a full wait would drain the prefetched encoding every 4 k-steps)."""
    from mipnerf_pl_amd.mlp_plan import emulate_wave
    a = gb.VARIANTS[vi]
    p = PrePlan.build(a)
    pf = p.fused
    assert [op.kmajor for op in pf.ops[:6]] == [True, False, False, False, False, True]
    assert [(s_.regset, s_.nk) for s_ in pf.ops[5].segs] == [("encg", 42), ("X", 16)] and pf.ops[0].segs[0].regset == "encg"
    params = orc.make_params(seed=21, density_gain=10.0, xyz_dim=a.xyz_dim)
    flat = np.concatenate([params[n].ravel() for n, _ in a.param_shapes()])
    rng = np.random.default_rng(3)
    enc = rng.uniform(-1, 1, (32, a.xyz_dim)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (32, 27)).astype(np.float32)
    view = np.zeros((32, 32), np.float32)
    view[:, :27] = v27
    for rb in (False, True):
        r1, d1 = emulate_wave(pf, flat, enc, view, rb)
        r2, d2 = emulate_pre_wave(p, flat, enc, view, rb)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2)
    rr, dd = orc.mlp_forward(params, enc[:, None, :], v27)
    r1, d1 = emulate_wave(pf, flat, enc, view, False)
    np.testing.assert_allclose(r1, rr[:, 0], atol=5e-6)
    np.testing.assert_allclose(d1, dd[:, 0, 0], atol=2e-5)
    # ---- the generated kernel text ----
    src = gb.gen_kernel(pf, vi)
    body = src[src.index("for (int tile = blockIdx.x;"):src.index("if (hi == 0 && s < M)")]
    R, D = gb.FUSED_RING, gb.FUSED_AHEAD
    assert len(re.findall(r"\n\s+MFMA\(", body)) == 1792 and "@VM@" not in src and "RING_WAIT" not in src
    events = []                                   # program order: ("dma", seq) | ("read", ring slot byte offset) | ("mfma", b operand)
    for ln in body.splitlines():
        for m in re.finditer(r"ENC_DMA\((\d+), (\w+), (\d+), (\d+)\)|(E\d) = LDB\((\d+)\)|MFMA\((\w+), A\d, ([\w\[\]]+)\)", ln):
            if m.group(1):
                events.append(("dma", int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))))
            elif m.group(5):
                events.append(("read", m.group(5), int(m.group(6))))
            else:
                events.append(("mfma", m.group(7), m.group(8)))
    dmas = [e_ for e_ in events if e_[0] == "dma"]
    assert [d[1] for d in dmas] == list(range(D, 84 + D))                      # 0 .. D-1: issued by the previous tile (or the kernel prologue)
    for _, i, base, goff, loff in dmas:
        assert goff == (i % 42) * 1024 and loff == 2048 + (i % 84 % R) * 1024 and base == ("encb" if i < 84 else "encb_next")
    pro = src[:src.index("for (int tile = blockIdx.x;")]
    assert [int(x) for x in re.findall(r"ENC_DMA\((\d+), encb,", pro)] == list(range(D))
    # ring discipline: walking the events, slot contents and who still has to read them
    ring_reads = [e_ for e_ in events if e_[0] == "read" and e_[2] >= 2048]
    assert len(ring_reads) == 84 and [(r[2] - 2048) // 1024 for r in ring_reads] == [i % R for i in range(84)]
    pos = {("dma", d[1]): k for k, d in enumerate(events) if d[0] == "dma"}
    reads_at = [k for k, e_ in enumerate(events) if e_[0] == "read" and e_[2] >= 2048]
    for i in range(D, 84):
        assert pos[("dma", i)] < reads_at[i]                                   # landed before read is the wait's business; issued before, in any case
        if i >= R:
            assert reads_at[i - R] < pos[("dma", i)]                           # the slot's previous k-step has been read (and consumed: its MFMAs follow the read by two k-steps)
            first_mfma_after_read = next(k for k in range(reads_at[i - R], len(events)) if events[k][0] == "mfma" and events[k][2] == events[reads_at[i - R]][1])
            assert first_mfma_after_read < pos[("dma", i)]
    # counted waits: group boundaries never wait for everything inside a tile, ring reads wait with small counts
    ks_ = [int(x) for x in re.findall(r"GROUP_BEGIN_CNT\(\d+, \d+, (\d+)\)", body)]
    assert len(ks_) == 55 and max(ks_) <= D + 1 and body.count("GROUP_BEGIN(0, 1);") == 1
    waits = [int(x) for x in re.findall(r's_waitcnt vmcnt\((\d+)\)" ::: "memory"\); E\d = LDB', body)]
    assert waits and max(waits) <= D + 4 and min(waits) >= D - 4


@pytest.mark.parametrize("ring,ahead", [(8, 7), (8, 5), (6, 5), (4, 3)])
@pytest.mark.parametrize("vi", VIS)
def test_one_kernel_form_counted_waits_replayed(vi, ring, ahead, monkeypatch):
    """An INDEPENDENT replay of the generated one-kernel form's vector-memory traffic (not gen_mlp_bf16.count_waits' own bookkeeping): walk the
    tile body in program order with the hardware's rule -- a wave's vector-memory operations retire in issue order, `s_waitcnt vmcnt(K)` returns
    when at most K are outstanding -- and require that (i) when a ring-group boundary's wait returns, this wave's four DMAs of that group have
    retired (the barrier behind it then makes that true for all eight waves), (ii) every read of an encoding ring slot happens after the DMA of
    the k-step it expects has retired, and that DMA is the LAST one issued into the slot, (iii) no DMA overwrites a slot whose k-step has not been
    read yet.  Two tiles back to back, so the hand-over of the next tile's first k-steps is replayed too."""
    # (the shipped geometry is 8 slots / 7 ahead; the others exercise the generator's other paths -- e.g. a group boundary whose DMAs an earlier,
    #  stricter ring wait has already retired -- with the same replay)
    monkeypatch.setattr(gb, "FUSED_RING", ring)
    monkeypatch.setattr(gb, "FUSED_AHEAD", ahead)
    p = PrePlan.build(gb.VARIANTS[vi])
    src = gb.gen_kernel(p.fused, vi)
    body = src[src.index("for (int tile = blockIdx.x;"):src.index("if (hi == 0 && s < M)")]
    pro = src[:src.index("for (int tile = blockIdx.x;")]
    R, D, NG = gb.FUSED_RING, gb.FUSED_AHEAD, len(p.fused.chunks) // 32
    tok = re.compile(r"GROUP_BEGIN\(0, \d+\)|GROUP_BEGIN_CNT\((\d+), \d+, (\d+)\)|ENC_DMA\((\d+), (\w+),|s_waitcnt vmcnt\((\d+)\)\" ::: \"memory\"\); |(E\d) = LDB\((\d+)\)")
    outstanding = []                       # issue order; entries ("grp", g, tile) | ("enc", seq, tile)
    slot_holds = {}                        # ring slot -> (tile, seq) of the last DMA issued into it
    retired = set()
    unread = {}                            # ring slot -> (tile, seq) landed or in flight and not read yet
    nreads = 0

    def wait(k):
        while len(outstanding) > k:
            retired.add(outstanding.pop(0))

    def issue(tag, slot=None):
        outstanding.append(tag)
        if slot is not None:
            assert slot not in unread, ("DMA into a ring slot whose k-step was not read", tag, unread[slot])
            slot_holds[slot] = tag
            unread[slot] = tag
    # kernel prologue: group 0 + the first tile's first D encoding k-steps
    for _ in range(4):
        issue(("grp", 0, 0))
    for i in (int(x) for x in re.findall(r"ENC_DMA\((\d+), encb,", pro)):
        issue(("enc", i, 0), i % R)
    for tile in range(2):
        nread_tile = 0
        for m in tok.finditer(body):
            t = m.group(0)
            if t.startswith("GROUP_BEGIN(0"):
                wait(0)
                for _ in range(4):
                    issue(("grp", 1, tile))
            elif t.startswith("GROUP_BEGIN_CNT"):
                g, k = int(m.group(1)), int(m.group(2))
                wait(k)
                assert not any(o[0] == "grp" and o[1] == g and o[2] == tile for o in outstanding), ("group not landed at its boundary", g, k)
                if g + 1 < NG:
                    for _ in range(4):
                        issue(("grp", g + 1, tile))
                else:
                    for _ in range(4):
                        issue(("grp", 0, tile + 1))          # has_next
            elif t.startswith("ENC_DMA"):
                i, base = int(m.group(3)), m.group(4)
                tag = ("enc", i % 84, tile + (1 if base == "encb_next" else 0))
                assert (base == "encb_next") == (i >= 84)
                issue(tag, (i % 84) % R)
            elif t.startswith("s_waitcnt"):
                wait(int(m.group(5)))
            elif int(m.group(7)) >= 2048:                     # a read of the encoding ring: the nread_tile-th k-step of this tile
                slot = (int(m.group(7)) - 2048) // 1024
                want = ("enc", nread_tile, tile)
                assert slot == nread_tile % R and slot_holds.get(slot) == want, (slot, want, slot_holds.get(slot))
                assert want in retired, ("ring slot read before its DMA has retired", want, list(outstanding)[:4])
                del unread[slot]
                nread_tile += 1
        assert nread_tile == 84
        nreads += nread_tile
    assert nreads == 168
