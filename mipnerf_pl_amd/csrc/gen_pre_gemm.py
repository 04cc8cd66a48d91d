#!/usr/bin/env python3
"""Generate the two-kernel bf16 MLP of the architecture variants whose encoding is too wide for k_mlp_bf16's wave-private LDS area
(mipnerf_pl_amd/mlp_pre_plan.py; today: the unbounded-scene model's 672 off-axis IPE features):

  pre_gemm_gen_v<i>.hip       k_pre_gemm: layer 0 and the encoding half of the skip layer (models/mip_nerf.py:83-90), k-step-major
  mlp_bf16_pre_gen_v<i>.hip   the trunk (layers 1.., heads): gen_mlp_bf16.gen_kernel on Plan.build(arch, pre_gemm=True)
  _gen_pre_tables_v<i>.bin    PrePlan.blob(): the index tables capi.hip packs both weight streams and bias tables with
  mlp_pre_variants_gen.hpp    launchers + dispatch tables indexed by variant

k_pre_gemm, per workgroup of 8 waves (2 per SIMD) and tile of 256 samples, straight-line code with a pinned schedule:
  * a wave owns 32 samples; two passes (W0, then W_skip[:, 256:]) of 42 k-steps x 8 output tiles = 336 MFMAs each; the 8 accumulator
    tiles of a pass stay in registers (128 VGPRs), so the encoding is read ONCE per pass;
  * B operand of a k-step = one 16-byte vector per lane straight from global memory (fragment layout written by k_cast_ipe_360, or
    the row-major [M, xyz_dim] bf16 buffer of the per-stage API), loaded DEPTH k-steps ahead into rotating registers -- across the
    pass boundary and across tiles;
  * A operands: the weight stream through a THREE-slot LDS ring (global_load_lds), two groups in flight; the ring barriers wait with a
    COUNTED vmcnt (the number of younger B-operand loads is known statically), so a barrier never waits for the prefetched operands;
  * pass 0 ends in bias-free ReLU + bf16 packing (the accumulators started from the bias image) and 16 fragment stores = the trunk's
    register set X; pass 1 stores its 8 accumulator tiles as fp32 images the trunk's skip layer starts from.

A second form of k_pre_gemm (mlp_pre_plan.SPLIT = MLP_PRE_SPLIT=1) gives the two matrices to the two halves of a workgroup on the same 128 samples (one HBM
read of the encoding); it is generated and tested too, and measured 3.6 % slower -- see mlp_pre_plan.py.

Usage: python gen_pre_gemm.py [outdir]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import gen_mlp_bf16 as gb  # noqa: E402
from mipnerf_pl_amd.mlp_pre_plan import GROUP, RING_SLOTS, PrePlan, supported  # noqa: E402

WAVES = 8
CHUNK = 1024
PREFETCH = 4                                              # A fragments in flight (registers A0..A3)
# rotating B-operand registers EB0..: DEPTH - 1 k-steps in flight.  Must divide the 84 k-steps of a tile (12 or 14), so that the register of
# k-step s is the same in every tile (the loads run ahead across the tile boundary)
DEPTH = int(os.environ.get("MLP_PRE_DEPTH", "12"))
DEPTH_SPLIT = int(os.environ.get("MLP_PRE_DEPTH_SPLIT", "14"))   # split form: must divide the 42 k-steps of a tile (6, 7, 14, 21)
VM_MARGIN = 4                                             # see the counted vmcnt below
# cache policy: non-temporal stores of the two outputs (written once, read by the NEXT kernel) and a non-temporal second read of the encoding
# (pass 1 is its last use).  Alternating A/B on one box, whole 8192 x (256 + 256) forward: 8.78 / 8.83 ms without, 8.72 / 8.74 with the stores,
# 8.68 / 8.72 with both (profiles/r04y_pre_gemm_policy_ab.txt); 13 instead of 11 operands in flight: no difference.  Both on by default.
NT_STORES = os.environ.get("MLP_PRE_NT_STORES", "1") == "1"
NT_PASS1 = os.environ.get("MLP_PRE_NT_PASS1", "1") == "1"
# round 6, timing probe for the one-kernel form: k_pre_gemm without its 1,536 B/sample of stores.  WRONG results (build.py: WRONG_RESULT_KNOBS)
ABLATE_STORES = os.environ.get("MLP_PRE_ABLATE_STORES", "0") == "1"


def gen_gemm(p: PrePlan, vi: int) -> str:
    a = p.arch
    nk, nt = p.nk, p.ntiles
    nchunks = len(p.chunks)
    assert nchunks == p.n_real_chunks and nchunks % (GROUP * RING_SLOTS) == 0, "k_pre_gemm streams whole ring revolutions without padding"
    ngroups = nchunks // GROUP
    ring_bytes = RING_SLOTS * GROUP * CHUNK
    bias_bytes = 2 * nt * 128
    lds_bytes = ring_bytes + bias_bytes
    split = p.split
    depth = DEPTH_SPLIT if split else DEPTH
    nsteps = nk if split else 2 * nk     # k-steps per tile and wave
    assert depth < nk and GROUP % nt == 0 and nsteps % depth == 0, "the B-register rotation must be tile-periodic"
    tile_waves = 4 if split else WAVES   # wave tiles (32 samples) per workgroup tile
    nslots = nsteps * nt                 # MFMAs per wave and tile

    def lda(c):
        """A fragment of wave-local slot c.  Split form: the stream is [k-step][matrix][tile]; the matrix (= the wave's half of the workgroup)
        is a wave-uniform 8-KiB offset folded into ring_lane"""
        if split:
            ks, t = divmod(c, nt)
            gc = ks * 2 * nt + t
        else:
            gc = c
        slot = (gc // GROUP) % RING_SLOTS
        return f"A{c % PREFETCH} = LDA({slot * GROUP * CHUNK + (gc % GROUP) * CHUNK});"

    slots_per_group = GROUP // 2 if split else GROUP      # wave-local MFMA slots between two ring barriers

    # ---- program-order event list of one tile body: ("gb", g) ring barrier of group g (issues group g + 2), ("ld", step) B-operand load
    body = []                            # (kind, payload) in emission order; "stmt" entries carry C++ text
    E = lambda kind, x: body.append((kind, x))
    for t in range(nt):
        E("stmt", f"BIAS(acc{t}, {t});")
    E("stmt", "PIN();")
    for c in range(nslots):
        step, t = divmod(c, nt)
        ps = 0 if split else step // nk
        if not split and ps == 1 and step == nk:
            # pass 0's epilogue tile by tile in front of pass 1's first k-step: X fragments 2t, 2t+1, then the accumulator restarts from b_skip
            E("stmt", f"epilogue_half<true, 0>(acc{t}, xo);  STORE_X({2 * t}, xo);")
            E("stmt", f"epilogue_half<true, 8>(acc{t}, xo);  STORE_X({2 * t + 1}, xo);")
            E("stmt", f"BIAS(acc{t}, {nt + t});")
        E("stmt", f"MFMA(acc{t}, A{c % PREFETCH}, EB{step % depth});")
        lc = c + PREFETCH
        if lc % slots_per_group == 0:
            E("gb", (lc // slots_per_group) % ngroups)  # lc == nslots: group 0 of the NEXT tile (its first A loads follow below)
        E("stmt", lda(lc % nslots))                     # past the end: the next tile's first fragments (harmless after the last tile)
        if t == nt - 1:
            # load the operand depth - 1 k-steps ahead (wrapping into the next pass / the next tile) into the register the PREVIOUS
            # k-step read, so the youngest MFMA that used it is 8 slots back
            E("ld", step + depth - 1)
        E("stmt", "PIN();")
    if split:
        # the tile's epilogue, by half of the workgroup (a wave-uniform branch): W0 half -> ReLU, bf16, the trunk's X fragments;
        # skip-layer half -> the accumulator tiles as they are
        E("stmt", "if (role == 0) {")
        for t in range(nt):
            E("stmt", f"    epilogue_half<true, 0>(acc{t}, xo);  STORE_X({2 * t}, xo);")
            E("stmt", f"    epilogue_half<true, 8>(acc{t}, xo);  STORE_X({2 * t + 1}, xo);")
            E("stmt", "    PIN();")
        E("stmt", "} else {")
        for t in range(nt):
            E("stmt", f"    STORE_ACC({t}, acc{t});")
            E("stmt", "    PIN();")
        E("stmt", "}")
    else:
        # pass 1's epilogue: plain stores of the 8 accumulator tiles
        for t in range(nt):
            E("stmt", f"STORE_ACC({t}, acc{t});")
            E("stmt", "PIN();")

    # ---- counted vmcnt of every ring barrier: vm operations known to be younger than the DMA of the group it waits for ----------
    # DMA(g) is issued inside barrier (g - 2) mod ngroups; steady state = the tile body repeated.  The first tile's prologue issues
    # DMA(0), DMA(1) and then DEPTH loads, which is never fewer younger operations than the steady state has, so the steady-state
    # count (a lower bound there too) is used for every tile.
    seq = [(k, x) for k, x in body if k in ("gb", "ld")] * 3
    gb_pos = [i for i, (k, x) in enumerate(seq) if k == "gb"]
    vmk = {}
    for i in gb_pos[2 * ngroups // 2 + 2:]:          # any barrier with two predecessors in the repeated sequence
        g = seq[i][1]
        prev = [j for j in gb_pos if j < i]
        j2 = prev[-2]                                 # the barrier that issued DMA(g)
        assert seq[j2][1] == (g - 2) % ngroups
        younger = sum(1 for j in range(j2 + 1, i) if seq[j][0] == "ld") + 4       # + DMA(g + 1): four chunks per wave
        vmk[g] = min(vmk.get(g, 63), younger)
    assert len(vmk) == ngroups and all(4 <= k <= 63 for k in vmk.values()), vmk
    prologue_younger = 4 + depth - 1
    vmk[0] = min(vmk[0], prologue_younger)
    # margin: the B-operand loads are ordinary C++ loads; should the compiler move a few of them across a barrier, the count stays a
    # lower bound (tests/test_pre_gemm_cpu.py counts the vector-memory instructions between the barriers of the compiled kernel)
    vmk = {g: k - VM_MARGIN for g, k in vmk.items()}

    L = []
    e = L.append
    e("// AUTO-GENERATED by gen_pre_gemm.py from mlp_pre_plan.py -- do not edit by hand.")
    e(f"// k_pre_gemm of architecture variant {vi}: enc[{a.xyz_dim}] x (W0 | W{p.skip_layer}[:, {a.net_width}:]) -> trunk inputs (models/mip_nerf.py:83-90)")
    e("#include <hip/hip_runtime.h>")
    e('#include "kernels.hpp"')
    e('#include "raymath.hpp"')
    e("namespace mip {")
    e(f"namespace pre_v{vi} {{")
    e("typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;")
    e("typedef __attribute__((ext_vector_type(16))) float f32x16;")
    e(f"constexpr int kRingBytes = {ring_bytes};")
    e(f"constexpr int kBiasBytes = {bias_bytes};")
    e(f"constexpr int kLdsBytes = {lds_bytes};")
    e(f"constexpr int kGroupBytes = {GROUP * CHUNK};")
    e(f"constexpr int kNumGroups = {ngroups};")
    e(f"constexpr int kTileSamples = {tile_waves * 32};")
    e(f"constexpr int kXyzDim = {a.xyz_dim};")
    e(f"constexpr int kNk = {nk};")
    e(gb.KERNEL_PREAMBLE.replace("BARRIER_INSN", "s_barrier").replace("WAIT_INSN", "s_waitcnt vmcnt(0) lgkmcnt(0)"))
    e("// ring barrier of group g with a counted vmcnt: K = vector-memory operations known to be younger than this wave's DMA of group g")
    e("// (the DMA of group g + 1 and the B-operand loads issued since), so the prefetched operands are never waited for here")
    e('#define RING_BARRIER(K) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\\n\\ts_barrier" ::"n"(K) : "memory")')
    e("// stores: wave-uniform base + literal offset (SGPRs) + the 32-bit lane offset, so no 64-bit VGPR address per 4 KiB of output stays live")
    if ABLATE_STORES:      # timing probe (WRONG results): the two outputs are computed and dropped -- what a kernel that keeps them on chip would not write
        e('#define ST16(p, v) do { auto v_ = (v); asm volatile("" :: "v"(v_)); } while (0)')
    elif NT_STORES:
        e("#define ST16(p, v) __builtin_nontemporal_store((v), (p))")
    else:
        e("#define ST16(p, v) *(p) = (v)")
    e("#define STORE_X(k, v) ST16(reinterpret_cast<bf16x8*>(xo_base + (k) * 1024 + lane16), (v))")
    e("// sub-vectors of the accumulator, not element-wise copies: the stores read the accumulator registers themselves")
    e("#define STORE_ACC(t, acc)                                                                                                  \\")
    e("    do {                                                                                                                   \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + (t) * 4096 + lane16), __builtin_shufflevector(acc, acc, 0, 1, 2, 3));                  \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + ((t) * 4096 + 1024) + lane16), __builtin_shufflevector(acc, acc, 4, 5, 6, 7));           \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + ((t) * 4096 + 2048) + lane16), __builtin_shufflevector(acc, acc, 8, 9, 10, 11));         \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + ((t) * 4096 + 3072) + lane16), __builtin_shufflevector(acc, acc, 12, 13, 14, 15));       \\")
    e("    } while (0)")
    e("// FRAG: enc is the fragment layout [wave tile][k-step][lane][8] (k_cast_ipe_360 writes it); else row-major [M, xyz_dim] bf16")
    e("// B-operand source of a wave tile = a wave-uniform base (SGPRs, so the loads take the saddr form) + a 32-bit lane offset; 16 bytes per")
    e("// lane and k-step.  Row-major: rows past M are clamped to the last one (their results are never stored).")
    e("template <bool FRAG>")
    e("__device__ __forceinline__ const char* bbase_of(const char* enc, int64_t wt, int64_t M) {")
    e("    if (FRAG) return enc + wt * (int64_t)(kNk * 1024);")
    e("    const int64_t s0 = wt * 32;")
    e("    return enc + (s0 < M ? s0 : M - 1) * (int64_t)(kXyzDim * 2);")
    e("}")
    e("template <bool FRAG>")
    e("__device__ __forceinline__ unsigned boff_of(int64_t wt, int lane, int64_t M) {")
    e("    if (FRAG) return (unsigned)lane * 16u;")
    e("    const int64_t s0 = wt * 32, s = s0 + (lane & 31);")
    e("    const int64_t r0 = s0 < M ? s0 : M - 1, r = s < M ? s : M - 1;")
    e("    return (unsigned)((r - r0) * (kXyzDim * 2) + (lane >> 5) * 16);")
    e("}")
    e("template <bool FRAG>")
    e(f"__global__ void __launch_bounds__({WAVES * 64}, 2)")
    e("k_pre_gemm(const char* __restrict__ stream, const float* __restrict__ bias_tab, const char* __restrict__ enc,")
    e("           char* __restrict__ pre_x, char* __restrict__ pre_acc, int64_t M, int ntiles, int nwg) {")
    e("    constexpr bool DMA = true;")
    e("    extern __shared__ __attribute__((aligned(16))) char smem[];")
    e("    // workgroup / workitem ids through the builtins: wave-uniform values the compiler KNOWS to be uniform (tile bases in SGPRs)")
    e("    const int tid = (int)__builtin_amdgcn_workitem_id_x();")
    e("    const int lane = tid & 63;")
    e("    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);")
    e("    const int hi = lane >> 5, n = lane & 31;")
    e("    const unsigned lane16 = (unsigned)lane * 16u;")
    if split:
        e("    // waves 0-3: W0 (role 0), waves 4-7: the skip layer's encoding half (role 1), on the SAME four wave tiles -- the second reader of a")
        e("    // fragment finds it in L1 / L2.  The role is a wave-uniform offset into the ring (8 chunks per k-step each) and the bias table")
        e("    const int role = wave >> 2, wsub = wave & 3;")
        e(f"    const char* ring_lane = smem + lane16 + role * {nt * CHUNK};")
        e(f"    const char* bias_lane = smem + kRingBytes + hi * 64 + role * {nt * 128};")
        e("#define WT_OF(tl) ((int64_t)(tl) * 4 + wsub)")
    else:
        e("    const char* ring_lane = smem + lane16;")
        e("    const char* bias_lane = smem + kRingBytes + hi * 64;")
        e(f"#define WT_OF(tl) ((int64_t)(tl) * {WAVES} + wave)")
    e(f"    for (int i = tid; i < kBiasBytes / 16; i += {WAVES * 64})")
    e("        reinterpret_cast<float4*>(smem + kRingBytes)[i] = reinterpret_cast<const float4*>(bias_tab)[i];")
    e("    __syncthreads();")
    e("    if (wave >= 4) __builtin_amdgcn_s_setprio(1);")
    e("    int tile = (int)__builtin_amdgcn_workgroup_id_x();")
    e("    if (tile >= ntiles) return;")
    e("    constexpr int kBStep = FRAG ? 1024 : 32;          // bytes between consecutive k-steps of a lane")
    e("    const char* bsrc = bbase_of<FRAG>(enc, WT_OF(tile), M);          // uniform")
    e("    const char* bnext = bsrc;")
    e("    unsigned boff0 = boff_of<FRAG>(WT_OF(tile), lane, M), boff0_next = boff0;")
    e("    bf16x8 " + ", ".join(f"A{i}" for i in range(PREFETCH)) + ", " + ", ".join(f"EB{i}" for i in range(depth)) + ", xo;")
    e("    f32x16 " + ", ".join(f"acc{t}" for t in range(nt)) + ";")
    e("    // prologue: ring groups 0 and 1, the first B operands, the first A fragments")
    e("    issue_group<DMA>(stream, smem, 0, 0, wave, lane16);")
    e("    issue_group<DMA>(stream, smem, 1, 1, wave, lane16);")
    e("    // the B operands are read through ONE running pointer (opaque to the compiler: with literal offsets it materialises a 64-bit base")
    e("    // per 4 KiB of the 42-KiB fragment run, for this tile and the next, and spills)")
    e("    const char* bp = bsrc;")
    e("    unsigned bo = boff0;")
    e('#define LOAD_B(reg) do { reg = *reinterpret_cast<const bf16x8*>(bp + bo); bo += kBStep; asm volatile("" : "+v"(bo)); } while (0)')
    e('#define LOAD_B_NT(reg) do { reg = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(bp + bo)); bo += kBStep; asm volatile("" : "+v"(bo)); } while (0)')
    for d in range(depth - 1):
        e(f"    LOAD_B(EB{d});")
    e(f"    RING_BARRIER({vmk[0]});")
    e("    issue_group<DMA>(stream, smem, 2, 2, wave, lane16);")
    for c in range(PREFETCH):
        e(f"    {lda(c)}")
    e("    for (;;) {")
    e("        const int tnext = tile + nwg;")
    e("        const bool has_next = tnext < ntiles;")
    e("        bnext = has_next ? bbase_of<FRAG>(enc, WT_OF(tnext), M) : bsrc;")
    e("        boff0_next = has_next ? boff_of<FRAG>(WT_OF(tnext), lane, M) : boff0;")
    e("        char* xo_base = pre_x + WT_OF(tile) * 16384;       // uniform")
    e("        char* ao_base = pre_acc + WT_OF(tile) * 32768;")
    for kind, x in body:
        if kind == "stmt":
            e(f"        {x}")
        elif kind == "ld":
            step = x
            if not split and step == nk:
                e("        bo = boff0;                // pass 1 reads the same operands again")
            elif step == nsteps:
                e("        bp = bnext; bo = boff0_next;    // ... and from here on the next tile's")
            last_use = NT_PASS1 and not split and nk <= step < nsteps          # pass 1's reads of the current tile
            e(f"        {'LOAD_B_NT' if last_use else 'LOAD_B'}(EB{step % depth});")
        else:
            g = x
            e(f"        RING_BARRIER({vmk[g]});      // group {g} readable, the slot of group {(g - 1) % ngroups} free")
            g2 = g + 2
            if g2 < ngroups:
                e(f"        issue_group<DMA>(stream, smem, {g2}, {g2 % RING_SLOTS}, wave, lane16);")
            else:
                # around the stream, also on the last tile: the counted waits assume DMA(g + 1) is younger than DMA(g) at EVERY barrier
                # (ADVICE r05); the phantom groups land in free slots and the kernel ends with vmcnt(0)
                e(f"        issue_group<DMA>(stream, smem, {g2 - ngroups}, {g2 % RING_SLOTS}, wave, lane16);")
    e("        if (!has_next) break;")
    e("        tile = tnext;")
    e("        bsrc = bnext;")
    e("        boff0 = boff0_next;")
    e("    }")
    e('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may land after the workgroup has released its LDS')
    e("#undef LOAD_B")
    e("#undef LOAD_B_NT")
    e("#undef WT_OF")
    e("}")
    e(f"}}  // namespace pre_v{vi}")
    e("")
    e("// enc: bf16, fragment layout (frag != 0; ceil(M / 256) * 8 wave tiles) or row-major [M, xyz_dim]; pre_x / pre_acc: 16 KiB / 32 KiB per wave tile")
    e(f"hipError_t launch_pre_gemm_v{vi}(const void* stream_w, const float* bias_tab, const void* enc, int frag, void* pre_x, void* pre_acc,")
    e("                              int64_t M, int grid_limit, hipStream_t st) {")
    e(f"    using namespace pre_v{vi};")
    e("    const int64_t nt64 = (M + kTileSamples - 1) / kTileSamples;")
    e("    if (nt64 < 1 || nt64 > 0x7fffffff) return hipErrorInvalidValue;")
    e("    const int ntiles = (int)nt64;")
    e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
    e("    if (grid < 1) grid = 1;")
    e("    static int attr_done[64] = {};")
    e("    int dev = 0;")
    e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
    e("    if (!attr_done[dev]) {")
    e("        hipError_t er = hipFuncSetAttribute((const void*)k_pre_gemm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        er = hipFuncSetAttribute((const void*)k_pre_gemm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        attr_done[dev] = 1;")
    e("    }")
    e(f"    if (frag) hipLaunchKernelGGL((k_pre_gemm<true>), dim3(grid), dim3({WAVES * 64}), kLdsBytes, st, (const char*)stream_w, bias_tab, (const char*)enc,")
    e("                                 (char*)pre_x, (char*)pre_acc, M, ntiles, grid);")
    e(f"    else hipLaunchKernelGGL((k_pre_gemm<false>), dim3(grid), dim3({WAVES * 64}), kLdsBytes, st, (const char*)stream_w, bias_tab, (const char*)enc,")
    e("                            (char*)pre_x, (char*)pre_acc, M, ntiles, grid);")
    e("    return hipGetLastError();")
    e("}")
    e("}  // namespace mip")
    return "\n".join(L) + "\n"


# ---- third form (round 5 experiment, MLP_PRE_FORM=once): every encoding k-step from HBM ONCE -------------------------------------------------
# 4-wave workgroups at ONE wave per SIMD (the 512-register budget the 512-wide trunk's kernel uses, gen_mlp_bf16.waves_of): a wave owns 32
# samples and ALL 16 output tiles of both matrices (256 accumulator registers, which the compiler keeps in acc VGPRs), so a k-step's B operand
# is loaded once and feeds 16 MFMAs.  Stream order [k-step][matrix][tile] (PrePlan split order), one ring group = one k-step = 16 chunks; every
# wave consumes every chunk, so there is a ring barrier per 16 of its MFMAs (as in the 512-wide trunk's kernel) and the stream is fetched once
# per 128 samples instead of once per 256.  Built (146 VGPRs + 256 acc VGPRs, 0 scratch), parity-green (MIPNERF_LIB=<that build> pytest
# tests/test_gpu_unbounded_bf16.py) and NOT faster: 7.975 / 7.977 / 8.059 ms per 8192 x (256 + 256) forward with a 7-slot ring against
# 7.954 / 7.976 / 7.986 of the default form in three alternating rounds, 8.21-8.30 with a 3-slot ring (profiles/r05t_pre_gemm_once_ab.txt):
# the second read of the encoding is not what bounds k_pre_gemm.  Kept as a knob; the default stays the two-pass form.
ONCE_DEPTH = int(os.environ.get("MLP_PRE_ONCE_DEPTH", "14"))      # B operands in flight + 1; must divide the 42 k-steps
ONCE_SLOTS = int(os.environ.get("MLP_PRE_ONCE_SLOTS", "3"))       # ring slots of 16 KiB; must divide the 42 groups


def gen_gemm_once(p: PrePlan, vi: int) -> str:
    a = p.arch
    nk, nt = p.nk, p.ntiles
    assert p.split, "the once-form consumes the [k-step][matrix][tile] stream"
    W4 = 4                                   # waves per workgroup
    G = 2 * nt                               # chunks per ring group = one k-step of both matrices
    nchunks = len(p.chunks)
    assert nchunks == p.n_real_chunks == nk * G and nk % ONCE_SLOTS == 0 and nk % ONCE_DEPTH == 0 and ONCE_DEPTH < nk
    ngroups, depth, slots = nk, ONCE_DEPTH, ONCE_SLOTS
    ring_bytes = slots * G * CHUNK
    bias_bytes = 2 * nt * 128
    lds_bytes = ring_bytes + bias_bytes
    nacc = 2 * nt
    nslots = nk * nacc

    def lda(c):
        return f"A{c % PREFETCH} = LDA({((c // G) % slots) * G * CHUNK + (c % G) * CHUNK});"

    body = []
    E = lambda kind, x: body.append((kind, x))
    for t in range(nacc):
        E("stmt", f"BIAS(acc{t}, {t});")
    E("stmt", "PIN();")
    for c in range(nslots):
        step, t = divmod(c, nacc)
        E("stmt", f"MFMA(acc{t}, A{c % PREFETCH}, EB{step % depth});")
        lc = c + PREFETCH
        if lc % G == 0:
            E("gb", (lc // G) % ngroups)
        E("stmt", lda(lc % nslots))
        if t == nacc - 1:
            E("ld", step + depth - 1)
        E("stmt", "PIN();")
    for t in range(nt):
        E("stmt", f"epilogue_half<true, 0>(acc{t}, xo);  STORE_X({2 * t}, xo);")
        E("stmt", f"epilogue_half<true, 8>(acc{t}, xo);  STORE_X({2 * t + 1}, xo);")
        E("stmt", "PIN();")
    for t in range(nt):
        E("stmt", f"STORE_ACC({t}, acc{nt + t});")
        E("stmt", "PIN();")
    # counted vmcnt, as in gen_gemm: DMA(g) is issued inside barrier g - (slots - 1); younger = the DMAs of the groups behind it (4 chunk
    # loads per wave each) + the B-operand loads since
    ahead = slots - 1
    seq = [(k, x) for k, x in body if k in ("gb", "ld")] * 3
    gb_pos = [i for i, (k, x) in enumerate(seq) if k == "gb"]
    vmk = {}
    for i in gb_pos[ngroups + 2:]:
        g = seq[i][1]
        prev = [j for j in gb_pos if j < i]
        j2 = prev[-ahead]
        assert seq[j2][1] == (g - ahead) % ngroups
        younger = sum(1 for j in range(j2 + 1, i) if seq[j][0] == "ld") + 4 * (ahead - 1)
        vmk[g] = min(vmk.get(g, 63), younger)
    assert len(vmk) == ngroups, vmk
    vmk = {g: max(0, min(k, 4 * (ahead - 1) + depth - 1) - VM_MARGIN) for g, k in vmk.items()}

    L = []
    e = L.append
    e("// AUTO-GENERATED by gen_pre_gemm.py (MLP_PRE_FORM=once) from mlp_pre_plan.py -- do not edit by hand.")
    e(f"// k_pre_gemm of architecture variant {vi}, every encoding k-step read once: 4-wave workgroups, one wave per SIMD, 16 accumulator tiles per wave")
    e("#include <hip/hip_runtime.h>")
    e('#include "kernels.hpp"')
    e('#include "raymath.hpp"')
    e("namespace mip {")
    e(f"namespace pre_v{vi} {{")
    e("typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;")
    e("typedef __attribute__((ext_vector_type(16))) float f32x16;")
    e("#define MIP_OPAQUE_STREAM_BASE 1")
    e(f"constexpr int kRingBytes = {ring_bytes};")
    e(f"constexpr int kBiasBytes = {bias_bytes};")
    e(f"constexpr int kLdsBytes = {lds_bytes};")
    e(f"constexpr int kGroupBytes = {G * CHUNK};")
    e(f"constexpr int kNumGroups = {ngroups};")
    e(f"constexpr int kTileSamples = {W4 * 32};")
    e(f"constexpr int kXyzDim = {a.xyz_dim};")
    e(f"constexpr int kNk = {nk};")
    e(gb.KERNEL_PREAMBLE.replace("BARRIER_INSN", "s_barrier").replace("WAIT_INSN", "s_waitcnt vmcnt(0) lgkmcnt(0)"))
    e('#define RING_BARRIER(K) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\\n\\ts_barrier" ::"n"(K) : "memory")')
    e("#define ST16(p, v) __builtin_nontemporal_store((v), (p))" if NT_STORES else "#define ST16(p, v) *(p) = (v)")
    e("#define STORE_X(k, v) ST16(reinterpret_cast<bf16x8*>(xo_base + (k) * 1024 + lane16), (v))")
    e("#define STORE_ACC(t, acc)                                                                                                  \\")
    e("    do {                                                                                                                   \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + (t) * 4096 + lane16), __builtin_shufflevector(acc, acc, 0, 1, 2, 3));                  \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + ((t) * 4096 + 1024) + lane16), __builtin_shufflevector(acc, acc, 4, 5, 6, 7));           \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + ((t) * 4096 + 2048) + lane16), __builtin_shufflevector(acc, acc, 8, 9, 10, 11));         \\")
    e("        ST16(reinterpret_cast<f32x4_*>(ao_base + ((t) * 4096 + 3072) + lane16), __builtin_shufflevector(acc, acc, 12, 13, 14, 15));       \\")
    e("    } while (0)")
    e("template <bool FRAG>")
    e("__device__ __forceinline__ const char* bbase_of(const char* enc, int64_t wt, int64_t M) {")
    e("    if (FRAG) return enc + wt * (int64_t)(kNk * 1024);")
    e("    const int64_t s0 = wt * 32;")
    e("    return enc + (s0 < M ? s0 : M - 1) * (int64_t)(kXyzDim * 2);")
    e("}")
    e("template <bool FRAG>")
    e("__device__ __forceinline__ unsigned boff_of(int64_t wt, int lane, int64_t M) {")
    e("    if (FRAG) return (unsigned)lane * 16u;")
    e("    const int64_t s0 = wt * 32, s = s0 + (lane & 31);")
    e("    const int64_t r0 = s0 < M ? s0 : M - 1, r = s < M ? s : M - 1;")
    e("    return (unsigned)((r - r0) * (kXyzDim * 2) + (lane >> 5) * 16);")
    e("}")
    e("template <bool FRAG>")
    e(f"__global__ void __launch_bounds__({W4 * 64}, 1)")
    e("k_pre_gemm(const char* __restrict__ stream, const float* __restrict__ bias_tab, const char* __restrict__ enc,")
    e("           char* __restrict__ pre_x, char* __restrict__ pre_acc, int64_t M, int ntiles, int nwg) {")
    e("    constexpr bool DMA = true;")
    e("    extern __shared__ __attribute__((aligned(16))) char smem[];")
    e("    const int tid = (int)__builtin_amdgcn_workitem_id_x();")
    e("    const int lane = tid & 63;")
    e("    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);")
    e("    const int hi = lane >> 5, n = lane & 31;")
    e("    const unsigned lane16 = (unsigned)lane * 16u;")
    e("    const char* ring_lane = smem + lane16;")
    e("    const char* bias_lane = smem + kRingBytes + hi * 64;")
    e(f"#define WT_OF(tl) ((int64_t)(tl) * {W4} + wave)")
    e(f"    for (int i = tid; i < kBiasBytes / 16; i += {W4 * 64})")
    e("        reinterpret_cast<float4*>(smem + kRingBytes)[i] = reinterpret_cast<const float4*>(bias_tab)[i];")
    e("    __syncthreads();")
    e("    int tile = (int)__builtin_amdgcn_workgroup_id_x();")
    e("    if (tile >= ntiles) return;")
    e("    constexpr int kBStep = FRAG ? 1024 : 32;")
    e("    const char* bsrc = bbase_of<FRAG>(enc, WT_OF(tile), M);")
    e("    const char* bnext = bsrc;")
    e("    unsigned boff0 = boff_of<FRAG>(WT_OF(tile), lane, M), boff0_next = boff0;")
    e("    bf16x8 " + ", ".join(f"A{i}" for i in range(PREFETCH)) + ", " + ", ".join(f"EB{i}" for i in range(depth)) + ", xo;")
    e("    f32x16 " + ", ".join(f"acc{t}" for t in range(nacc)) + ";")
    for g in range(ahead):
        e(f"    issue_group<DMA>(stream, smem, {g}, {g}, wave, lane16);")
    e("    const char* bp = bsrc;")
    e("    unsigned bo = boff0;")
    e('#define LOAD_B(reg) do { reg = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(bp + bo)); bo += kBStep; asm volatile("" : "+v"(bo)); } while (0)')
    for d in range(depth - 1):
        e(f"    LOAD_B(EB{d});")
    e(f"    RING_BARRIER({vmk[0]});")
    e(f"    issue_group<DMA>(stream, smem, {ahead}, {ahead % slots}, wave, lane16);")
    for c in range(PREFETCH):
        e(f"    {lda(c)}")
    e("    for (;;) {")
    e("        const int tnext = tile + nwg;")
    e("        const bool has_next = tnext < ntiles;")
    e("        bnext = has_next ? bbase_of<FRAG>(enc, WT_OF(tnext), M) : bsrc;")
    e("        boff0_next = has_next ? boff_of<FRAG>(WT_OF(tnext), lane, M) : boff0;")
    e("        char* xo_base = pre_x + WT_OF(tile) * 16384;")
    e("        char* ao_base = pre_acc + WT_OF(tile) * 32768;")
    for kind, x in body:
        if kind == "stmt":
            e(f"        {x}")
        elif kind == "ld":
            if x == nk:
                e("        bp = bnext; bo = boff0_next;    // from here on the next tile's operands")
            e(f"        LOAD_B(EB{x % depth});")
        else:
            g = x
            e(f"        RING_BARRIER({vmk[g]});      // group {g} readable, the slot of group {(g - 1) % ngroups} free")
            g2 = g + ahead
            if g2 < ngroups:
                e(f"        issue_group<DMA>(stream, smem, {g2}, {g2 % slots}, wave, lane16);")
            else:
                e(f"        issue_group<DMA>(stream, smem, {g2 - ngroups}, {g2 % slots}, wave, lane16);      // around the stream, also on the last tile (see gen_gemm)")
    e("        if (!has_next) break;")
    e("        tile = tnext;")
    e("        bsrc = bnext;")
    e("        boff0 = boff0_next;")
    e("    }")
    e('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
    e("#undef LOAD_B")
    e("#undef WT_OF")
    e("}")
    e(f"}}  // namespace pre_v{vi}")
    e("")
    e(f"hipError_t launch_pre_gemm_v{vi}(const void* stream_w, const float* bias_tab, const void* enc, int frag, void* pre_x, void* pre_acc,")
    e("                              int64_t M, int grid_limit, hipStream_t st) {")
    e(f"    using namespace pre_v{vi};")
    e("    // whole 256-sample tiles of the trunk kernel: an even number of 128-sample tiles (pre_x / pre_acc are sized for them)")
    e("    const int64_t nt64 = ((M + 255) / 256) * 2;")
    e("    if (nt64 < 1 || nt64 > 0x7fffffff) return hipErrorInvalidValue;")
    e("    const int ntiles = (int)nt64;")
    e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
    e("    if (grid < 1) grid = 1;")
    e("    static int attr_done[64] = {};")
    e("    int dev = 0;")
    e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
    e("    if (!attr_done[dev]) {")
    e("        hipError_t er = hipFuncSetAttribute((const void*)k_pre_gemm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        er = hipFuncSetAttribute((const void*)k_pre_gemm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        attr_done[dev] = 1;")
    e("    }")
    e(f"    if (frag) hipLaunchKernelGGL((k_pre_gemm<true>), dim3(grid), dim3({W4 * 64}), kLdsBytes, st, (const char*)stream_w, bias_tab, (const char*)enc,")
    e("                                 (char*)pre_x, (char*)pre_acc, M, ntiles, grid);")
    e(f"    else hipLaunchKernelGGL((k_pre_gemm<false>), dim3(grid), dim3({W4 * 64}), kLdsBytes, st, (const char*)stream_w, bias_tab, (const char*)enc,")
    e("                            (char*)pre_x, (char*)pre_acc, M, ntiles, grid);")
    e("    return hipGetLastError();")
    e("}")
    e("}  // namespace mip")
    return "\n".join(L) + "\n"


def variants_header(vis, n):
    L = ["// AUTO-GENERATED by gen_pre_gemm.py -- do not edit by hand.", "#pragma once", '#include "kernels.hpp"', "namespace mip {",
         "// two-kernel bf16 MLP of the variants whose encoding is too wide for k_mlp_bf16's wave-private LDS area (mlp_pre_plan.py)",
         "typedef hipError_t (*LaunchPreGemmFn)(const void* stream_w, const float* bias_tab, const void* enc, int frag, void* pre_x, void* pre_acc,",
         "                                      int64_t M, int grid_limit, hipStream_t st);",
         "typedef hipError_t (*LaunchBf16PreFn)(const void* stream_w, const float* bias_tab, const void* pre_x, const void* pre_acc, const void* viewenc,",
         "                                      float* rgb_sigma, float* raw_out, int64_t M, int num_samples, float density_bias, float rgb_padding,",
         "                                      int grid_limit, const float* dnoise, float dnoise_scale, hipStream_t st);"]
    for vi in vis:
        L.append(f"hipError_t launch_pre_gemm_v{vi}(const void*, const float*, const void*, int, void*, void*, int64_t, int, hipStream_t);")
        L.append(f"hipError_t launch_mlp_bf16_pre_v{vi}(const void*, const float*, const void*, const void*, const void*, float*, float*, int64_t, int, float,")
        L.append("                                   float, int, const float*, float, hipStream_t);")
        L.append(f"hipError_t launch_mlp_bf16_fused_v{vi}(const void*, const float*, const void*, const void*, const void*, float*, float*, int64_t, int, float,")
        L.append("                                     float, int, const float*, float, hipStream_t);")
        L.append(f'extern "C" const unsigned char mip_pre_tables_v{vi}[];')
    f = lambda fmt: ", ".join(fmt.format(vi) if vi in vis else "nullptr" for vi in range(n))
    L.append(f"static const LaunchPreGemmFn kLaunchPreGemm[{n}] = {{{f('launch_pre_gemm_v{}')}}};")
    L.append(f"static const LaunchBf16PreFn kLaunchBf16Pre[{n}] = {{{f('launch_mlp_bf16_pre_v{}')}}};")
    L.append("// one-kernel form (round 6): same launcher shape as the trunk's; pre_x = the encoding's fragment buffer, pre_acc = nullptr")
    L.append(f"static const LaunchBf16PreFn kLaunchBf16Fused[{n}] = {{{f('launch_mlp_bf16_fused_v{}')}}};")
    L.append(f"static const unsigned char* const kPreTableBlobs[{n}] = {{{f('mip_pre_tables_v{}')}}};")
    L.append("}  // namespace mip")
    return "\n".join(L) + "\n"


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else HERE
    vis = [vi for vi, a in enumerate(gb.VARIANTS) if supported(a)]
    for vi in vis:
        once = os.environ.get("MLP_PRE_FORM", "") == "once"
        p = PrePlan.build(gb.VARIANTS[vi], split=True) if once else PrePlan.build(gb.VARIANTS[vi])
        with open(os.path.join(outdir, f"pre_gemm_gen_v{vi}.hip"), "w") as f:
            f.write(gen_gemm_once(p, vi) if once else gen_gemm(p, vi))
        with open(os.path.join(outdir, f"mlp_bf16_pre_gen_v{vi}.hip"), "w") as f:
            f.write(gb.gen_kernel(p.trunk, vi))
        # round 6: the ONE-kernel form of the same model (Plan.build(arch, fused=True)): layer 0 and the skip layer as k-step-major ops of the trunk
        # kernel, their encoding streamed through a wave-private LDS ring; its pack / bias tables ride behind the two-kernel form's in the blob
        with open(os.path.join(outdir, f"mlp_bf16_fused_gen_v{vi}.hip"), "w") as f:
            f.write(gb.gen_kernel(p.fused, vi))
        with open(os.path.join(outdir, f"_gen_pre_tables_v{vi}.bin"), "wb") as f:
            f.write(p.blob())
        print(f"variant {vi}: pre-GEMM {p.n_real_chunks} chunks ({p.nk} k-steps x {p.ntiles} tiles x 2 passes), trunk {p.trunk.n_real_chunks} "
              f"(+{len(p.trunk.chunks) - p.trunk.n_real_chunks} pad) chunks, {p.trunk.n_tiles} tiles")
    with open(os.path.join(outdir, "mlp_pre_variants_gen.hpp"), "w") as f:
        f.write(variants_header(vis, len(gb.VARIANTS)))


if __name__ == "__main__":
    main()
