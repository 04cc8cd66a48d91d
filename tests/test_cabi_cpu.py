"""CPU (no GPU): the C-ABI shared library loads, exports every symbol include/mipnerf_hip.h
declares, and its host-side plan expansion equals mipnerf_pl_amd/mlp_plan.py.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mipnerf_pl_amd import _lib as L
from mipnerf_pl_amd.mlp_plan import Plan, emulate_wave
from oracle import mipnerf_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        from mipnerf_pl_amd import build
        build.build(verbose=False)
    return L.lib()


def header_functions():
    src = open(os.path.join(REPO, "include", "mipnerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mipnerf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mipnerf_hip.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(L.SIGNATURES) == names


def _dynamic_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_dynamic_symbol_table_is_exactly_the_c_abi(lib):
    """the libraries export the entry points their headers declare and NOTHING else -- no mip::launch_* C++ launchers,
    kernel handles or table blobs (a linker version script in build.py keeps `mipnerf_*` only)."""
    assert _dynamic_symbols(L.LIB_PATH) == header_functions()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "mipnerf_diag.h")).read(), flags=re.S)
    assert _dynamic_symbols(L.DIAG_LIB_PATH) == sorted(set(re.findall(r"\b(mipnerf_[a-z0-9_]+)\s*\(", src)))


def test_abi_version_and_compiled_arch(lib):
    assert lib.mipnerf_abi_version() == 6
    cfg = L.Config()
    assert lib.mipnerf_compiled_arch(C.byref(cfg)) == 0
    assert (cfg.net_depth, cfg.net_width, cfg.net_depth_condition, cfg.net_width_condition, cfg.skip_index) == (8, 256, 1, 128, 4)
    assert (cfg.max_deg_point, cfg.deg_view) == (16, 4)


def test_error_reporting_without_gpu(lib):
    # argument validation happens before any HIP call
    assert lib.mipnerf_sample_along_rays(0, 64, None, None, None, 0, None, None) == L.E_INVALID
    assert b"sample_along_rays" in lib.mipnerf_last_error()
    with pytest.raises(ValueError):
        L.check(L.E_INVALID, "x")
    with pytest.raises(NotImplementedError):
        L.check(L.E_UNSUPPORTED, "x")


@pytest.mark.parametrize("which,ref", [(0, "pack_table"), (1, "bias_table"), (2, "pack_table_f32")])
def test_host_plan_tables_match_python_plan(lib, which, ref):
    plan = Plan.build()
    want = getattr(plan, ref)().astype(np.int32).ravel()
    n = lib.mipnerf_debug_table(which, None, 0)
    assert n == want.size
    got = np.empty(n, np.int32)
    assert lib.mipnerf_debug_table(which, got.ctypes.data, n) == n
    np.testing.assert_array_equal(got, want)


def test_f32_layer_descriptors(lib):
    plan = Plan.build()
    n = lib.mipnerf_debug_f32net(None, 0)
    buf = np.empty((n, 12), np.int32)
    lib.mipnerf_debug_f32net(buf.ctypes.data, buf.size)
    layers = plan.f32_layers()
    assert n == len(layers)
    chunk0 = 0
    W = plan.arch.net_width
    for i, (row, Ly) in enumerate(zip(buf, layers)):
        assert tuple(row[:5]) == (Ly["x_in0"], Ly["kb0"], Ly["x_in1"], Ly["kb1"], Ly["x_out"])
        assert row[5] == len(Ly["tiles"]) and row[6] == Ly["first_tile"] and row[7] == int(Ly["relu"]) and row[8] == Ly["kind"]
        assert row[9] == chunk0 and row[10] == int(Ly["kind"] == 1)
        chunk0 += len(Ly["tiles"]) * Ly["kb"]
        assert row[11] == 2 * W + 96 + 4
        # double buffering: a layer never writes the buffer it (or its second segment) reads
        assert row[4] != row[0] and (row[3] == 0 or row[2] == 2 * W)
        if i > 0 and Ly["kb0"] * 16 == W:
            assert row[0] == buf[i - 1][4]                     # reads what the previous layer wrote
    assert [r[8] for r in buf] == [0] * 8 + [1, 0, 2]
    assert (buf[5][0], buf[5][1], buf[5][2], buf[5][3]) == (W, 16, 2 * W, 6)     # skip concat: [x | encoding]


def test_register_dataflow_emulation_matches_oracle():
    """The k-permutation / D-layout repacking the bf16 kernel relies on, emulated in numpy in fp32
    (no bf16 rounding), reproduces the oracle MLP to fp32 round-off."""
    plan = Plan.build()
    params = orc.make_params(seed=11, density_gain=10.0)
    names = [n for n, _ in plan.arch.param_shapes()]
    assert names == list(params.keys())
    flat = np.concatenate([params[n].ravel() for n in names])
    rng = np.random.default_rng(0)
    enc = rng.uniform(-1, 1, (32, 96)).astype(np.float32)
    v27 = rng.uniform(-1, 1, (32, 27)).astype(np.float32)
    view = np.zeros((32, 32), np.float32)
    view[:, :27] = v27
    rgb, dens = emulate_wave(plan, flat, enc, view)
    rr, dd = orc.mlp_forward(params, enc[:, None, :], v27)
    np.testing.assert_allclose(rgb, rr[:, 0], atol=5e-6)
    np.testing.assert_allclose(dens, dd[:, 0, 0], atol=2e-5)


def _variants():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "gen_mlp_bf16", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mipnerf_pl_amd", "csrc", "gen_mlp_bf16.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.VARIANTS


def test_architecture_variants_exported(lib):
    """Every MLP shape of gen_mlp_bf16.VARIANTS is carried by the library (mipnerf_variant_arch) and the C++ table
    expansion of each equals mlp_plan.py (bf16 stream incl. its zero padding, bias table, fp32 stream)."""
    variants = _variants()
    assert lib.mipnerf_num_variants() == len(variants) >= 6
    for v, arch in enumerate(variants):
        cfg, has_train = L.Config(), C.c_int(-1)
        assert lib.mipnerf_variant_arch(v, C.byref(cfg), C.byref(has_train)) == 0
        assert (cfg.net_depth, cfg.net_width, cfg.net_depth_condition, cfg.net_width_condition, cfg.skip_index,
                bool(cfg.use_viewdirs)) == (arch.net_depth, arch.net_width, arch.net_depth_condition, arch.net_width_condition,
                                            arch.skip_index, arch.use_viewdirs)
        # bf16 training kernels for every variant that has bf16 kernels and widths <= 256 (mlp_train_plan.py; round 5: two view layers train in
        # bf16 too; the 512-wide trunk has a bf16 inference kernel and trains in fp32)
        # ... or the two-kernel form's training kernels (round 5: the wide-encoding variant of the unbounded-scene model)
        from mipnerf_pl_amd import mlp_pre_plan
        assert has_train.value == int((arch.bf16_kernels and arch.net_width <= 256)
                                       or mlp_pre_plan.supported(arch))
        assert bool(cfg.unbounded) == (arch.feat_per_deg == 42) and (cfg.max_deg_point - cfg.min_deg_point) * arch.feat_per_deg == arch.xyz_dim
        plan = Plan.build(arch)
        for which, ref in ((0, "pack_table"), (1, "bias_table"), (2, "pack_table_f32")):
            want = getattr(plan, ref)().astype(np.int32).ravel()
            n = lib.mipnerf_debug_table_variant(v, which, None, 0)
            got = np.empty(n, np.int32)
            assert lib.mipnerf_debug_table_variant(v, which, got.ctypes.data, n) == n
            if which == 2:      # the C++ table is sized like the bf16 stream; the fp32 stream fills its real chunks
                assert n >= want.size and (got[want.size:] == -1).all()
                got = got[:want.size]
            np.testing.assert_array_equal(got, want)


def test_variant_dataflow_emulation_matches_oracle():
    """Register dataflow of the generated bf16 kernels for the non-default shapes (half-width trunk; no view directions:
    density-only head, colour head on the trunk output), emulated in fp32, against the oracle MLP."""
    for arch in _variants()[1:]:
        plan = Plan.build(arch)
        params = orc.make_params(seed=21, density_gain=10.0, net_width=arch.net_width, net_width_condition=arch.net_width_condition,
                                 net_depth=arch.net_depth, skip_index=arch.skip_index, xyz_dim=arch.xyz_dim,
                                 net_depth_condition=arch.net_depth_condition)
        names = [n for n, _ in arch.param_shapes()]
        assert names == list(params.keys())
        flat = np.concatenate([params[n].ravel() for n in names])
        rng = np.random.default_rng(1)
        enc = rng.uniform(-1, 1, (32, arch.xyz_dim)).astype(np.float32)
        v27 = rng.uniform(-1, 1, (32, 27)).astype(np.float32)
        view = np.zeros((32, 32), np.float32)
        view[:, :27] = v27
        rgb, dens = emulate_wave(plan, flat, enc, view)
        rr, dd = orc.mlp_forward(params, enc[:, None, :], v27 if arch.use_viewdirs else None, skip_index=arch.skip_index,
                                 net_depth=arch.net_depth, net_depth_condition=arch.net_depth_condition)
        np.testing.assert_allclose(rgb, rr[:, 0], atol=5e-6)
        np.testing.assert_allclose(dens, dd[:, 0, 0], atol=2e-5)


def test_diagnostics_live_in_their_own_library(lib):
    """VERDICT r03 hygiene: the drop-in library is the hot path only -- the MFMA ceilings and the hand-off probe are exported by
    libmipnerf_diag.so (include/mipnerf_diag.h), header == exports == ctypes table, and NOT by libmipnerf_hip.so."""
    src = open(os.path.join(REPO, "include", "mipnerf_diag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(mipnerf_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(L.DIAG_SIGNATURES) and len(names) == 3
    d = L.diag_lib()
    assert d is not None
    for n in names:
        assert hasattr(d, n)
        if n != "mipnerf_diag_last_error":
            assert not hasattr(lib, n), f"{n} still exported by the drop-in library"
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "ceiling" not in syms and "handoff" not in syms


def test_diagnostic_entry_points_validate_their_arguments(lib):
    """mipnerf_mfma_ceiling / mipnerf_handoff_probe reject bad arguments before touching the device; a later SUCCESSFUL validation
    does not clear the message."""
    out3, out6 = (C.c_double * 3)(), (C.c_double * 6)()
    lib = L.diag_lib()
    assert lib.mipnerf_mfma_ceiling(0, 3, 1, 1.0, out3, None) == L.E_INVALID          # waves per SIMD must be 1 or 2
    assert lib.mipnerf_mfma_ceiling(4, 2, 1, 1.0, out3, None) == L.E_INVALID          # feeding mode 0 ... 3
    assert lib.mipnerf_mfma_ceiling(0, 2, 1, 100.0, out3, None) == L.E_INVALID        # bounded run time
    assert lib.mipnerf_mfma_ceiling(0, 2, 1, 1.0, None, None) == L.E_INVALID
    assert lib.mipnerf_handoff_probe(1, 0, 0, 4, 131072, 0, 1, out6, None) == L.E_INVALID       # no tiles
    assert lib.mipnerf_handoff_probe(1, 0, 16, 4, 1000, 0, 1, out6, None) == L.E_INVALID        # tile too small
    assert lib.mipnerf_handoff_probe(1, 0, 16, 64, 1 << 22, 0, 1, out6, None) == L.E_INVALID    # ring x tile x 128 pairs > 4 GiB
    assert b"handoff_probe" in lib.mipnerf_diag_last_error()
    assert lib.mipnerf_handoff_probe(1, 5, 16, 4, 131072, 0, 1, out6, None) == L.E_INVALID      # no such protocol
    assert lib.mipnerf_handoff_probe(0, 4, 16, 4, 131072, 0, 1, out6, None) == L.E_INVALID      # fence-free protocols are same-XCD only
    assert lib.mipnerf_handoff_probe(1, 3, 16, 4, 262144, 0, 1, out6, None) == L.E_INVALID      # per-wave protocols: 64 / 128 KiB tiles


def _torch_mlp(params, arch, enc, venc):
    """plain-torch restatement of MLP.forward (mip_nerf.py:75-111) on a state dict -- only to compare a padded with an unpadded net"""
    import torch
    x = enc
    h = x
    for i in range(arch["net_depth"]):
        h = torch.relu(torch.nn.functional.linear(h, params[f"layers.{i}.0.weight"], params[f"layers.{i}.0.bias"]))
        if i % arch["skip_index"] == 0 and i > 0:
            h = torch.cat([h, x], -1)
    dens = torch.nn.functional.linear(h, params["density_layer.weight"], params["density_layer.bias"])
    b = torch.nn.functional.linear(h, params["extra_layer.weight"], params["extra_layer.bias"])
    v = torch.cat([b, venc], -1)
    for i in range(arch["net_depth_condition"]):
        v = torch.relu(torch.nn.functional.linear(v, params[f"view_layers.{i}.0.weight"], params[f"view_layers.{i}.0.bias"]))
    rgb = torch.nn.functional.linear(v, params["color_layer.weight"], params["color_layer.bias"])
    return torch.cat([rgb, dens], -1)


@pytest.mark.parametrize("w,wc,depth,skip,dc", [(200, 72, 8, 4, 1), (100, 40, 8, 4, 1), (40, 24, 6, 3, 1), (130, 90, 8, 4, 2)])
def test_width_padding_keeps_the_function_and_its_gradient(lib, w, wc, depth, skip, dc):
    """model.WidthPadding (host logic behind MLP widths that are not generated shapes): the variant search returns the smallest
    containing generated shape; scattering the true parameters into zeros of that shape gives a network with the SAME outputs,
    and the gather of its gradient IS the gradient of the true parameters (every padded entry's gradient is exactly 0)."""
    import torch
    from mipnerf_pl_amd import model as M
    arch = dict(net_depth=depth, net_width=w, net_depth_condition=dc, net_width_condition=wc, skip_index=skip, num_rgb_channels=3,
                num_density_channels=1, xyz_dim=96, view_dim=27)
    found = M.containing_variant(arch, True, False, False)
    assert found is not None and not found[1]
    padded_arch = found[0]
    assert padded_arch["net_width"] >= w and padded_arch["net_width"] % 32 == 0 and padded_arch["net_width_condition"] >= wc
    assert M.containing_variant(dict(arch, net_width=padded_arch["net_width"], net_width_condition=padded_arch["net_width_condition"]),
                                True, False, False)[1]                       # the containing shape itself is an exact match
    pad = M.WidthPadding(arch, padded_arch)
    layout = M.param_layout(arch)
    gen = torch.Generator().manual_seed(w)
    true = {n: (torch.rand(shp, generator=gen, dtype=torch.float64) - 0.5).requires_grad_() for n, shp, _ in layout}
    assert pad.true_numel == sum(p.numel() for p in true.values()) and len(set(pad.index.tolist())) == pad.true_numel
    flat = torch.zeros(pad.padded_numel, dtype=torch.float64)
    pad.scatter(list(true.values()), flat)
    flat.requires_grad_()
    padded = dict(zip([n for n, _, _ in layout], pad.padded_views(flat)))
    assert [tuple(v.shape) for v in padded.values()] == [shp for _, shp, _ in M.param_layout(padded_arch)]
    enc = torch.randn(50, 96, generator=gen, dtype=torch.float64)
    venc = torch.randn(50, 27, generator=gen, dtype=torch.float64)
    y0 = _torch_mlp(true, arch, enc, venc)
    y1 = _torch_mlp(padded, padded_arch, enc, venc)
    assert float((y0 - y1).abs().max()) <= 1e-13 * float(y0.abs().max())      # summation order of the BLAS only
    wgt = torch.randn(50, 4, generator=gen, dtype=torch.float64)
    (y0 * wgt).sum().backward()
    (y1 * wgt).sum().backward()
    g_true = torch.cat([p.grad.reshape(-1) for p in true.values()])
    assert float((pad.gather(flat.grad) - g_true).abs().max()) <= 1e-13 * float(g_true.abs().max())
    mask = torch.ones(pad.padded_numel, dtype=torch.bool)
    mask[pad.index] = False
    assert float(flat.grad[mask].abs().max()) == 0.0                        # padded entries never move under any optimiser
    # nothing wider than the widest generated shape
    assert M.containing_variant(dict(arch, net_width=600), True, False, False) is None


def test_launch_attribute_caches_are_per_device(lib):
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property: a launcher that remembers "done" once per process would
    launch its > 64-KiB-LDS kernel without the attribute on the second GPU a process touches (MLP.native(device) keeps one context per
    device).  Every cache in the hand-written and the generated sources must be indexed by the current device."""
    csrc = os.path.join(REPO, "mipnerf_pl_amd", "csrc")
    checked = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp", ".py")):
            continue
        src = open(os.path.join(csrc, f)).read()
        if "hipFuncSetAttribute" not in src:
            continue
        checked += 1
        assert not re.search(r"static\s+bool\s+\w*attr\w*", src), f"{f}: per-process attribute cache"
        for m in re.finditer(r"static\s+int\s+(\w*attr\w*)\s*\[", src):
            assert re.search(r"hipGetDevice\s*\(", src), f"{f}: {m.group(1)} is not indexed by the device"
    assert checked >= 10


def test_wrong_result_build_knobs_need_an_explicit_opt_in(monkeypatch):
    """the timing-experiment variables (ablated barriers, transposing reads on untransposed data, ...) give WRONG results; a stale
    one in the environment must not silently build the product library.  build.py refuses them unless MIPNERF_EXPERIMENT_BUILD=1 AND the
    output is not libmipnerf_hip.so; the library such a build produces refuses mipnerf_create() without MIPNERF_ALLOW_EXPERIMENT_LIB=1."""
    from mipnerf_pl_amd import build as b
    for k in b.WRONG_RESULT_KNOBS:
        monkeypatch.delenv(k, raising=False)
    assert b.experiment_flags() == ""
    monkeypatch.setenv("MLP_WGRAD_TR", "1")
    with pytest.raises(RuntimeError, match="WRONG results"):
        b.experiment_flags()
    monkeypatch.setenv("MIPNERF_EXPERIMENT_BUILD", "1")
    with pytest.raises(RuntimeError, match="WRONG results"):        # still the product library's name
        b.experiment_flags()
    monkeypatch.setattr(b, "LIB", os.path.join(b.CSRC, "libmipnerf_hip_exp.so"))
    assert b.experiment_flags() == "MLP_WGRAD_TR=1"
    src = open(os.path.join(REPO, "mipnerf_pl_amd", "csrc", "capi.hip")).read()
    assert "MIPNERF_ALLOW_EXPERIMENT_LIB" in src and "#ifdef MIPNERF_EXPERIMENT_BUILD" in src


def test_one_wave_per_simd_kernel_ring_is_consistent():
    """The 512-wide trunk's bf16 kernel (gen_mlp_bf16.waves_of: 4-wave workgroups, one wave per SIMD) keeps kAhead ring groups in flight with
    counted vmcnt waits in kSlots = kAhead + 2 slots (round 6: the spare slot means a boundary refills the slot of group g - 2, whose reads
    returned long ago, and need not drain the wave's in-flight A-fragment reads).  From the generated source: every chunk is read from the slot
    its group was issued into, every group boundary is there exactly once, group 0 waits for everything (the tile's encoding DMAs are younger than
    the ring's), every other boundary leaves exactly the 4 (kAhead - 1) DMAs of the groups behind it in flight, the group count keeps the ring
    phase tile-invariant, and the refill never depends on has_next."""
    src = open(os.path.join(REPO, "mipnerf_pl_amd", "csrc", "mlp_bf16_gen_v6.hip")).read()
    ahead = int(re.search(r"constexpr int kAhead = (\d+);", src).group(1))
    slots = int(re.search(r"constexpr int kSlots = (\d+);", src).group(1))
    ngroups = int(re.search(r"constexpr int kNumGroups = (\d+);", src).group(1))
    group_bytes = int(re.search(r"constexpr int kGroupBytes = (\d+);", src).group(1))
    ring_bytes = int(re.search(r"constexpr int kRingBytes = (\d+);", src).group(1))
    assert slots == ahead + 2 and ring_bytes == slots * group_bytes and group_bytes == 16 * 1024 and ngroups % slots == 0
    assert "__launch_bounds__(256, 1)" in src and "MIP_OPAQUE_STREAM_BASE" in src
    body = src[src.index("for (int tile = blockIdx.x;"):]
    body = body[:body.index("if (hi == 0 && s < M)")]
    offs = [int(x) for x in re.findall(r"A\d+ = LDA\((\d+)\);", body)]
    assert len(offs) == ngroups * 16
    for c, off in enumerate(offs):                      # chunk c of the stream: group c // 16 lives in slot (c // 16) % slots
        assert off == ((c // 16) % slots) * group_bytes + (c % 16) * 1024, (c, off)
    begins = [(kind, int(g), int(w)) for kind, g, w in re.findall(r"GROUP_BEGIN_DEEP(_NODRAIN)?\((\d+), (\d+)\);", body)]
    assert [g for _, g, _ in begins] == list(range(ngroups))
    assert begins[0][0] == "" and begins[0][2] == 0 and all(k == "_NODRAIN" and w == 4 * (ahead - 1) for k, _, w in begins[1:])
    # the prologue issues groups 0 .. kAhead - 1 into slots 0 .. kAhead - 1; the macros refill slot (g + kAhead) % kSlots with group g + kAhead
    pro = src[:src.index("for (int tile = blockIdx.x;")]
    assert [tuple(map(int, m)) for m in re.findall(r"issue_group<DMA>\(stream, smem, (\d+), (\d+), wave, lane16\);", pro)][-ahead:] == [(g, g) for g in range(ahead)]
    for name in ("GROUP_BEGIN_DEEP(g, WAITCNT)", "GROUP_BEGIN_DEEP_NODRAIN(g, WAITCNT)"):
        macro = src[src.index("#define " + name):]
        macro = macro[:macro.index("} while (0)")]
        assert "has_next" not in macro and "(g) + kAhead - kNumGroups" in macro and "((g) + kAhead) % kSlots" in macro
    nodrain = src[src.index("#define GROUP_BEGIN_DEEP_NODRAIN"):]
    assert "lgkmcnt" not in nodrain[:nodrain.index("} while (0)")]
    tail = src[src.index("if (hi == 0 && s < M)"):]
    assert 's_waitcnt vmcnt(0)' in tail[:tail.index("\n}\n")]


def test_one_wave_per_simd_kernel_ring_replayed():
    """ADVICE r05 (high) was a counted wait that leaned on DMAs which are not in flight on a workgroup's last tile.  An independent replay of the
    512-wide kernel's ring (the rule of the hardware: a wave's vector-memory operations retire in issue order, `vmcnt(K)` leaves K outstanding),
    two tiles back to back with the refill running around the stream: at every group boundary the wave's four DMAs of that group have retired
    when the wait returns, every A-fragment read comes from the slot its group was DMA'd into LAST, and no refill overwrites a slot before all 16
    chunks of the group it held have been read."""
    src = open(os.path.join(REPO, "mipnerf_pl_amd", "csrc", "mlp_bf16_gen_v6.hip")).read()
    ahead = int(re.search(r"constexpr int kAhead = (\d+);", src).group(1))
    slots = int(re.search(r"constexpr int kSlots = (\d+);", src).group(1))
    ngroups = int(re.search(r"constexpr int kNumGroups = (\d+);", src).group(1))
    pro = src[:src.index("for (int tile = blockIdx.x;")]
    body = src[src.index("for (int tile = blockIdx.x;"):]
    body = body[:body.index("if (hi == 0 && s < M)")]
    outstanding, holds, reads_done = [], {}, {}

    def issue(group_id, slot):
        prev = holds.get(slot)
        assert prev is None or reads_done.get(prev, 0) == 16, ("refill over unread chunks", group_id, prev, reads_done.get(prev, 0))
        holds[slot] = group_id
        outstanding.extend([group_id] * 4)
    for g, sl in re.findall(r"issue_group<DMA>\(stream, smem, (\d+), (\d+), wave, lane16\);", pro)[-ahead:]:
        issue((0, int(g)), int(sl))
    tok = re.compile(r"GROUP_BEGIN_DEEP(?:_NODRAIN)?\((\d+), (\d+)\);|A\d+ = LDA\((\d+)\);")
    for tile in range(2):
        chunk = 0
        for m in tok.finditer(body):
            if m.group(3) is None:
                g, k = int(m.group(1)), int(m.group(2))
                while len(outstanding) > k:
                    outstanding.pop(0)
                assert (tile, g) not in outstanding, ("group not landed at its boundary", tile, g, k)
                nxt = g + ahead
                issue((tile + nxt // ngroups, nxt % ngroups), nxt % slots)          # around the stream, unconditionally
            else:
                slot = int(m.group(3)) // (16 * 1024)
                gid = (tile, chunk // 16)
                assert holds.get(slot) == gid and gid not in outstanding, ("A fragment read from a slot that does not hold its group", gid, holds.get(slot))
                reads_done[gid] = reads_done.get(gid, 0) + 1
                chunk += 1
        assert chunk == ngroups * 16


def test_sample_count_limit_is_one_number_everywhere():
    """num_samples <= 1024 = 64 lanes x 16 samples: the header, the ctypes layer, the LDS row constants and the samples-per-lane dispatch of
    every per-ray kernel (buckets 1, 2, 4, 8, 16) say the same."""
    csrc = os.path.join(REPO, "mipnerf_pl_amd", "csrc")
    hdr = open(os.path.join(REPO, "include", "mipnerf_hip.h")).read()
    assert int(re.search(r"#define MIPNERF_MAX_SAMPLES (\d+)", hdr).group(1)) == L.MAX_SAMPLES == 1024
    assert "constexpr int kPdfMaxBins = 1024;" in open(os.path.join(csrc, "raywave.hpp")).read()
    assert "constexpr int kMaxBins = 1024;" in open(os.path.join(csrc, "kernels_resample_grad.hip")).read()
    n = 0
    for f in ("kernels_ray.hip", "kernels_train.hip", "kernels_resample_grad.hip"):
        src = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r"case 5: case 6: case 7: case 8: (MIP_\w+)\(8\); break;\s*\n\s*case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: (MIP_\w+)\(16\); break;\s*\n\s*default: return hipErrorInvalidValue;", src):
            assert m.group(1) == m.group(2)
            n += 1
        assert src.count("(8); break;") == src.count("(16); break;"), f          # no dispatch without the 16-samples-per-lane bucket
    assert n == 7
