#!/usr/bin/env python3
"""Generate mlp_bf16_gen.hip (the register-resident bf16 MFMA MLP kernel) and
mlp_plan_gen.hpp (plan descriptors for the host-side packer) from mlp_plan.Plan.

The kernel is straight-line code: every chunk of the weight stream has a compile-time LDS
ring offset, every activation register a literal index (runtime-indexed vector arrays would
be demoted to scratch), and the software pipeline is explicit:

  * A fragments (weights) are read from the LDS ring PREFETCH chunks ahead of the MFMA that
    consumes them, into rotating registers A0..A3;
  * the encodings (B operands of layer 0, the skip layer and the view layer) are DMA'd once
    per tile into a wave-private LDS area and read back two k-steps ahead (E0..E2), so they
    never occupy registers across layers;
  * two accumulator pairs alternate between panels, so the bias/ReLU/bf16 epilogue of panel
    p-1 and the bias loads of panel p+1 overlap with the MFMAs of panel p;
  * every slot ends with sched_barrier(0) so hipcc keeps exactly this schedule (left alone it
    hoists hundreds of LDS reads / address computations and spills ~1.6 KB per lane).

Usage: python gen_mlp_bf16.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mipnerf_pl_amd.mlp_plan import CHAIN, Plan, Arch  # noqa: E402

WAVES = int(os.environ.get("MLP_WAVES", "8"))   # wavefronts per workgroup (32 samples each); 4 => two 128-sample workgroups per CU
GROUP = 4 * WAVES     # chunks per ring slot (wave w DMAs chunks 4w..4w+3 of a group)
SLOTS = 2             # ring slots
WG_PER_CU = 8 // WAVES
CHUNK_BYTES = 1024
PREFETCH = int(os.environ.get("MLP_PREFETCH", "4"))   # A-fragment prefetch distance in chunks (registers A0..)
ABLATE_BARRIER = os.environ.get("MLP_ABLATE_BARRIER", "0") == "1"   # timing experiments only (wrong results)
ABLATE_WAIT = os.environ.get("MLP_ABLATE_WAIT", "0") == "1"         # timing experiments only (wrong results)
ABLATE_LDA = int(os.environ.get("MLP_ABLATE_LDA", "0"))              # N > 0: read only every (N+1)-th A fragment (wrong results)
# static priority 1 for the second-dispatched half of the workgroup (the arbitration loser on every segment): 0.4722-0.4756 vs
# 0.4762-0.4773 ms per launch in three alternating A/B pairs (+0.4 %, profiles/r02k_setprio_ab.log); MLP_SETPRIO=0 turns it off
SETPRIO = os.environ.get("MLP_SETPRIO", "1") == "1"
# The integrated positional encoding of the NEXT tile computed piecewise in the shadow of this tile's MFMAs (after the last layer
# that reads the encoding: its LDS area is free from there on) instead of in a VALU-only phase at the start of every tile.
IPE_SHADOW = os.environ.get("MLP_IPE_SHADOW", "0") == "1"    # measured: -0.56 % cycles, +0.25 % time (profiles/r03f_ipe_shadow_ab.txt): off
IPE_SHADOW_STRIDE = int(os.environ.get("MLP_IPE_SHADOW_STRIDE", "4"))    # one piece every STRIDE slots
# trunk kernels of the two-kernel form: pre_x / pre_acc (read once, 1.5 KB per sample) with the non-temporal policy, so that they do not push the weight
# stream out of the L2: 7.803 / 7.812 / 7.819 vs 7.835 / 7.837 / 7.862 ms per forward in three alternating pairs (-0.4 %, profiles/r04z_trunk_nt_loads_ab.txt)
PRE_NT = os.environ.get("MLP_PRE_NT_LOADS", "1") == "1"
# round 6, timing probe for the one-kernel form: the trunk without its 1,536 B/sample of pre_x / pre_acc loads -- the values come
# from LDS instead (stale weights as X, the bias table as accumulator images).  WRONG results (build.py: WRONG_RESULT_KNOBS)
ABLATE_PRELOADS = os.environ.get("MLP_TRUNK_ABLATE_PRELOADS", "0") == "1"
NE = 3                # rotating registers for LDS-resident B operands (E0..E2)
# one-kernel form of a wide encoding (Plan.fused): wave-private LDS ring of encoding k-steps (1 KiB each, global_load_lds from the fragment buffer
# k_cast_ipe_360 writes) and how many k-steps ahead of its MFMAs a k-step's DMA is issued (its B-operand read happens two k-steps ahead)
FUSED_RING = int(os.environ.get("MLP_FUSED_RING", "8"))
FUSED_AHEAD = int(os.environ.get("MLP_FUSED_AHEAD", "7"))
ENC_WAVE_BYTES = 8192  # wave-private LDS: 6 KiB encoding + 2 KiB view encoding


# Architectures the library is generated for.  Variant 0 is the reference's shipped configuration (configs/lego.yaml).  Every
# variant gets the plan tables, a bf16 inference kernel (this file) and -- when mlp_train_plan.py covers the shape and
# gen_mlp_train.py's schedule passes its hazard check -- bf16 training kernels + tables; otherwise it trains in fp32 mode.  To
# support another MLP shape add it here and rebuild (python -m mipnerf_pl_amd.build): units, tables and the dispatch headers
# (mlp_variants_gen.hpp, mlp_train_variants_gen.hpp) follow from this list.
VARIANTS = [
    Arch(),                                                            # 8 x 256, view layer 128, view directions
    Arch(net_width=128, net_width_condition=128),                      # half-width trunk (proves the plan is parametric)
    Arch(net_width_condition=256, use_viewdirs=False),                 # MLP.forward(x, None): colour head on the trunk output
    Arch(net_depth=6, skip_index=3),                                   # another depth / skip period (20 parameter tensors)
    # SURVEY 8(f)-4, the unbounded-scene model: MipNerf(unbounded=True) feeds the 2 x 21 x 16 = 672 off-axis IPE features of the
    # contracted Gaussians (kernels_360.hip) to the same 8 x 256 trunk (first layer 672 -> 256, skip concat 256 + 672).  bf16_kernels=False:
    # 42 k-steps per sample do not fit THIS generator's 8-KiB wave-private encoding area -- its bf16 form is the two-kernel one (gen_pre_gemm.py:
    # k_pre_gemm + a trunk kernel generated here with pre_gemm=True), inference and training.
    Arch(xyz_dim=672, feat_per_deg=42, bf16_kernels=False),
    # two view layers (mlp_net_depth_condition = 2, mip_nerf.py:62-69).  Round 5: bf16 INFERENCE kernel too -- its stream is 39 ring
    # groups + one whole group of zero padding (the ring phase must be tile-invariant: an even number of groups), which the generator
    # handles since the trunk kernels of round 4 (an extra GROUP_BEGIN at the tile end).  bf16 TRAINING kernels too: the plan saves one more
    # activation set / delta set / mask row per extra view layer (mlp_train_plan.py).
    Arch(net_depth_condition=2),
    # a 512-wide trunk with a 256-wide view layer.  Round 5: bf16 INFERENCE kernel too -- 4-wave workgroups at one wave per SIMD (waves_of below).
    # Training: fp32 forward + GEMM backward.
    Arch(net_width=512, net_width_condition=256),
]


def _ops_table(plan: Plan, name: str):
    L = [f"static const OpDesc {name}[{len(plan.ops)}] = {{"]
    f32 = plan.f32_layers()
    for op, fl in zip(plan.ops, f32):
        segs = ", ".join(f"{{{s.kind}, {s.nk}, {s.col0}, {s.ncols}}}" for s in op.segs)
        if len(op.segs) == 1:
            segs += ", {0, 0, 0, 0}"
        tiles = ", ".join(f"{{{t.wt}, {t.bt}, {t.row0}, {t.nrows}, {t.ld}}}" for t in op.tiles)
        assert len(op.tiles) <= 17
        kind = 1 if op.name == "head" else (2 if op.out == "rgb" else 0)
        L.append(f"  /* {op.name} */ {{{len(op.segs)}, {{{segs}}}, {len(op.tiles)}, {{{tiles}}}, {op.first_tile}, {fl['x_in']}, {int(op.relu)}, {kind}}},")
    L.append("};")
    return L


def gen_plan_header(plans) -> str:
    plan = plans[0]
    a = plan.arch
    L = []
    L.append("// AUTO-GENERATED by gen_mlp_bf16.py from mlp_plan.py -- do not edit.")
    L.append("#pragma once")
    L.append("#include <stdint.h>")
    L.append("namespace mip { namespace plan {")
    L.append("// ---- variant 0 (the shipped architecture; the bf16 training kernels exist for it only) ----")
    L.append(f"constexpr int kNetDepth = {a.net_depth}, kNetWidth = {a.net_width}, kNetDepthCond = {a.net_depth_condition},")
    L.append(f"              kNetWidthCond = {a.net_width_condition}, kSkipIndex = {a.skip_index}, kNumRgb = {a.num_rgb},")
    L.append(f"              kNumDensity = {a.num_density}, kXyzDim = {a.xyz_dim}, kViewDim = {a.view_dim};")
    shapes = a.param_shapes()
    L.append(f"constexpr int kNumParamTensors = {len(shapes)};")
    L.append("constexpr int kParamNumel[kNumParamTensors] = {" +
             ", ".join(str(int(np.prod(s))) for _, s in shapes) + "};")
    L.append(f"constexpr int kNumChunks = {len(plan.chunks)}, kNumTiles = {plan.n_tiles}, kNumOps = {len(plan.ops)};")
    L.append(f"constexpr int kGroupChunks = {GROUP}, kRingSlots = {SLOTS}, kNumGroups = {len(plan.chunks) // GROUP};")
    L.append(f"constexpr bool kChainOrder = {'true' if CHAIN else 'false'};   // weight-stream order (mlp_plan.CHAIN)")
    L.append("struct SegDesc { int kind, nk, col0, ncols; };            // kind 0 natural, 1 dlayout")
    L.append("struct TileDesc { int wt, bt, row0, nrows, ld; };")
    L.append("// kind 0 hidden, 1 head (last tile = density), 2 colour")
    L.append("struct OpDesc { int nsegs; SegDesc segs[2]; int ntiles; TileDesc tiles[17]; int first_tile; int xcol_in; int relu; int kind; };")
    L.append("// ---- every architecture the library was generated for (gen_mlp_bf16.VARIANTS) ----")
    L.append("struct PlanDesc {")
    L.append("    int net_depth, net_width, net_depth_cond, net_width_cond, skip_index, num_rgb, num_density, xyz_dim, view_dim, use_viewdirs;")
    L.append("    int feat_per_deg;   // encoding features per frequency: 6 (axis-aligned IPE) or 42 (off-axis IPE of the unbounded-scene model)")
    L.append("    int num_param_tensors; int param_numel[32];")
    L.append("    int num_chunks, num_real_chunks, num_tiles, num_ops;")
    L.append("    const OpDesc* ops;")
    L.append("    int variant;        // index of the generated bf16 inference kernel (launch_mlp_bf16 / launch_mlp_bf16_v<variant>)")
    L.append("};")
    for vi, pl in enumerate(plans):
        L += _ops_table(pl, "kOps" if vi == 0 else f"kOps_v{vi}")
    L.append(f"constexpr int kNumVariants = {len(plans)};")
    L.append("static const PlanDesc kPlans[kNumVariants] = {")
    for vi, pl in enumerate(plans):
        b = pl.arch
        shp = b.param_shapes()
        assert len(shp) <= 32
        numel = ", ".join(str(int(np.prod(x))) for _, x in shp)
        L.append(f"  {{{b.net_depth}, {b.net_width}, {b.net_depth_condition}, {b.net_width_condition}, {b.skip_index}, {b.num_rgb}, "
                 f"{b.num_density}, {b.xyz_dim}, {b.view_dim}, {int(b.use_viewdirs)}, {b.feat_per_deg}, {len(shp)}, {{{numel}}}, {len(pl.chunks)}, "
                 f"{pl.n_real_chunks}, {pl.n_tiles}, {len(pl.ops)}, {'kOps' if vi == 0 else f'kOps_v{vi}'}, {vi}}},")
    L.append("};")
    L.append("}}  // namespace mip::plan")
    return "\n".join(L) + "\n"


def build_schedule(plan: Plan):
    """Flatten the plan into panels and per-chunk slots.

    slot c: MFMA of chunk c with accumulator `acc` and B operand `b`; b is either a literal
    register ('reg', 'X[3]') or ('lds', byte offset in the wave-private area)."""
    a = plan.arch
    nenc_lds = 0 if (plan.pre_gemm or plan.fused) else a.xyz_dim // 16
    panels, slots = [], []
    ring_seq = 0           # fused plans: running index of the encoding k-steps streamed through the wave-private ring (layer 0: 0.., skip layer: nk..)
    for op in plan.ops:
        if op.kmajor:
            # one WIDE panel: all output tiles accumulate at once (accW0..), chunks in [k-step][tile] order; the encoding k-steps come
            # through the wave-private LDS ring ('ring', sequence index), the activation k-steps from registers as everywhere
            nt = len(op.tiles)
            first = len(slots)
            for ks in range(op.nk):
                seg, ksl = plan.seg_of(op, ks)
                if seg.regset == "encg":
                    b = ("ring", ring_seq)
                    ring_seq += 1
                else:
                    assert seg.regset in ("X", "Y")
                    b = ("reg", f"{seg.regset}[{seg.reg0 + ksl}]")
                for t in range(nt):
                    slots.append(dict(acc=f"accW{t}", b=b, panel=len(panels), ks=ks, first_of_ks=(t == 0)))
            panels.append(dict(op=op, t0=None, t1=None, tiles=list(range(nt)), wide=True, first=first, n=len(slots) - first, pair=None, spk=nt))
            continue
        for (t0, t1) in plan.panels(op):
            pair = len(panels) & 1
            first = len(slots)
            spk = 1 if t1 is None else 2

            def bop(ks):
                seg, ksl = plan.seg_of(op, ks)
                if seg.regset in ("X", "Y"):
                    return ("reg", f"{seg.regset}[{seg.reg0 + ksl}]")
                if seg.regset == "enc":
                    return ("lds", ksl * 1024)
                return ("lds", nenc_lds * 1024 + ksl * 1024)   # view: after the encoding k-steps
            if CHAIN:
                for w in range(spk):
                    for ks in range(op.nk):
                        slots.append(dict(acc=f"acc{pair}{w}", b=bop(ks), panel=len(panels), ks=ks, first_of_ks=True))
            else:
                for ks in range(op.nk):
                    for w in range(spk):
                        slots.append(dict(acc=f"acc{pair}{w}", b=bop(ks), panel=len(panels), ks=ks, first_of_ks=(w == 0)))
            panels.append(dict(op=op, t0=t0, t1=t1, first=first, n=len(slots) - first, pair=pair, spk=1 if CHAIN else spk))
    return panels, slots


import re  # noqa: E402
_WRITES = re.compile(r"epilogue_half<[^>]*>\(\w+, ([XY]\[\d+\])\);\s*/\*op(\d+)\*/")


def _check_hazards(plan, panels, slots, b_expr, side):
    """Replay the tile in program order: every activation register X[k] / Y[k] read by a slot of op i must have been written
    last by the previous op that writes activation registers at all (same discipline as gen_mlp_train.check_hazards).  Without
    this, a view layer or trunk narrower than two panels generated a kernel whose consumer read the registers before the
    producer's epilogue had written them (found with an ad-hoc 8 x 192 / 64 shape: bf16 rgb off by 0.2, fp32 fine)."""
    last = {f"X[{k}]": -1 for k in range(16)} if plan.pre_gemm else {}     # trunk plans: X arrives preloaded ("op -1")
    for c, sl in enumerate(slots):
        b = b_expr[c]
        if b and b[0] in "XY":
            op = panels[sl["panel"]]["op"]
            cur = plan.ops.index(op)
            writers = [o for o in set(last.values()) if o < cur]
            assert writers and last.get(b) == max(writers), ("slot reads a register not produced by the previous writing op",
                                                             c, op.name, b, last.get(b))
        for stmt in side[c]:
            m = _WRITES.search(stmt)
            if m:
                last[m.group(1)] = int(m.group(2))


def epilogue_pieces(plan, pn):
    """C++ statements finishing panel `pn`, split into small pieces (one per slot)."""
    op, pair = pn["op"], pn["pair"]
    relu = "true" if op.relu else "false"
    pieces = []
    for which, t in ([(t, t) for t in pn["tiles"]] if pn.get("wide") else ((0, pn["t0"]), (1, pn["t1"]))):
        if t is None:
            continue
        acc = f"accW{t}" if pn.get("wide") else f"acc{pair}{which}"
        if op.out in ("X", "Y"):
            if op.name == "head" and t == len(op.tiles) - 1:
                pieces.append(f"raw_density = {acc}[0];   // row 0 of the density tile (lanes hi=0)")
            else:
                pieces.append(f"epilogue_half<{relu}, 0>({acc}, {op.out}[{2 * t}]);  /*op{plan.ops.index(op)}*/")
                pieces.append(f"epilogue_half<{relu}, 8>({acc}, {op.out}[{2 * t + 1}]);  /*op{plan.ops.index(op)}*/")
        else:
            pieces.append(f"raw_r = {acc}[0]; raw_g = {acc}[1]; raw_b = {acc}[2];   // rows 0..2 of the colour tile")
    return pieces


def bias_pieces(plan, pn):
    op, pair = pn["op"], pn["pair"]
    if pn.get("wide"):
        return [f"BIAS(accW{t}, {op.first_tile + t});" for t in pn["tiles"]]
    if op.pre:      # accumulator images written by k_pre_gemm (gen_pre_gemm.py): 4 KiB per tile and wave, four lane-linear 1-KiB loads
        out = [f"pre_load(acc{pair}0, pre_lane + {pn['t0'] * 4096});"]
        if pn["t1"] is not None:
            out.append(f"pre_load(acc{pair}1, pre_lane + {pn['t1'] * 4096});")
        return out
    out = [f"BIAS(acc{pair}0, {op.first_tile + pn['t0']});"]
    if pn["t1"] is not None:
        out.append(f"BIAS(acc{pair}1, {op.first_tile + pn['t1']});")
    return out


KERNEL_PREAMBLE = r"""
#ifndef MIP_CHUNKS_PER_WAVE
#define MIP_CHUNKS_PER_WAVE 4      // 1-KiB chunks a wave moves per ring group (issue_group)
#endif
constexpr int kChunksPerWave = MIP_CHUNKS_PER_WAVE;
#define LDA(off) (*reinterpret_cast<const bf16x8*>(ring_lane + (off)))
#define LDB(off) (*reinterpret_cast<const bf16x8*>(enc_lane + (off)))
#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), acc, 0, 0, 0)
#define BIAS(acc, g) acc = *reinterpret_cast<const f32x16*>(bias_lane + (g) * 128)
#define PIN() __builtin_amdgcn_sched_barrier(0)

// ReLU as ONE v_max_f32 the compiler knows.  The generated MLP units are built with -fno-honor-nans -mno-amdgpu-ieee
// (build.py): in IEEE mode llvm.maxnum gets a canonicalising v_max in front of it for sNaN quieting.  NOT inline asm: hipcc's hazard recogniser does not see the
// operands of an asm statement, so an asm VALU result can be allocated onto a register an in-flight MFMA still reads or
// writes (dead A fragments, unused accumulator rows) with no wait states -- observed as corrupted ReLU masks when a
// v_pk_min_u16 asm landed on rows 7..10 of a retiring accumulator (see gen_mlp_train.py).
__device__ __forceinline__ float relu1(float x) {
    return __builtin_fmaxf(x, 0.0f);
}

// Half of the bias/ReLU epilogue of one 32x32 D tile: accumulator registers R0..R0+7 ARE one
// k-step B operand of the next layer (mlp_plan.py "dlayout"); fp32 -> bf16 is RNE.
template <bool RELU, int R0>
__device__ __forceinline__ void epilogue_half(const f32x16& acc, bf16x8& o) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float v = RELU ? relu1(acc[R0 + r]) : acc[R0 + r];
        o[r] = (__bf16)v;
    }
}

// (PRE_LD: read-once data of the trunk kernels -- plain or non-temporal loads, build knob MLP_PRE_NT_LOADS)
#ifndef PRE_LD
#define PRE_LD(ptr) (*(ptr))
#endif
// Trunk kernels of the two-kernel form (mlp_pre_plan.py): one accumulator tile as k_pre_gemm stored it -- registers 4q .. 4q+3 of every
// lane form one lane-linear 1-KiB row, four rows per tile.
typedef __attribute__((ext_vector_type(4))) float f32x4_;
__device__ __forceinline__ void pre_load(f32x16& acc, const char* p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4_ v = PRE_LD(reinterpret_cast<const f32x4_*>(p + q * 1024));
        acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
    }
}

// Stage one ring group (32 chunks = 32 KiB): wave w moves chunks 4w..4w+3.  The global address
// is kept as (wave-uniform 64-bit base in SGPRs) + (32-bit lane offset) so the compiler selects
// the saddr form and does not materialise one 64-bit VGPR address per chunk; the four chunks of
// a wave share one base and differ by the instruction's immediate offset, which the hardware
// adds to BOTH the global address and the LDS (M0) destination.
template <bool DMA>
__device__ __forceinline__ void issue_group(const char* __restrict__ stream, char* smem, int group, int slot,
                                            int wave, unsigned lane16) {
#ifdef MIP_OPAQUE_STREAM_BASE
    // long streams (the 512-wide trunk: 288 groups): without this the compiler hoists one loop-invariant 64-bit base PER GROUP out of the tile
    // loop and spills ~530 SGPRs to VGPR lanes; an opaque copy makes each base two scalar adds at its point of use
    asm volatile("" : "+s"(stream));
#endif
    const char* gbase = stream + ((size_t)group * kGroupBytes + (size_t)wave * (kChunksPerWave * 1024));   // uniform
    char* lbase = smem + slot * kGroupBytes + wave * (kChunksPerWave * 1024);                              // uniform
    if (DMA) {
#pragma unroll
      for (int blk = 0; blk < kChunksPerWave / 4; ++blk, gbase += 4096, lbase += 4096) {
        // Issued through inline asm on purpose: hipcc models the builtin as a FLAT access with a pending
        // LDS side effect, which degrades EVERY later `s_waitcnt lgkmcnt(N)` to lgkmcnt(0) and puts a
        // vmcnt(0) in front of the bias reads for as long as the DMA is in flight (i.e. always).  Its
        // completion is waited for by hand in GROUP_BEGIN (vmcnt(0) + s_barrier).
        // LDS destination = M0 base + imm + lane*16: every chunk lands lane-linear.
        const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbase;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
            "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(lane16), "s"(gbase), "s"(lds_addr)
            : "memory");
      }
    } else {
        // opaque copy: stops LLVM from re-associating (stream + lane) + constant into one hoisted 64-bit
        // VGPR address per chunk (38 groups x 4 chunks of them would be spilled to scratch)
        asm volatile("" : "+v"(lane16));
#pragma unroll
        for (int i = 0; i < kChunksPerWave; ++i)
            *reinterpret_cast<float4*>(lbase + i * 1024 + lane16) =
                *reinterpret_cast<const float4*>(gbase + i * 1024 + lane16);
    }
}

// Stage this wave's B operands that come from memory (integrated positional encoding, NENC
// k-steps, and the padded view encoding, 2 k-steps) into its private LDS area: k-step k is one
// lane-linear 1-KiB fragment (lane (hi, n) holds features k*16 + hi*8 .. +7 of sample n).
template <bool DMA, int NENC, int FIRST>
__device__ __forceinline__ void issue_encodings(const __bf16* ep, const __bf16* vp, char* encw, unsigned lane16) {
#pragma unroll
    for (int k = FIRST; k < NENC + 2; ++k) {
        const __bf16* src = k < NENC ? ep + k * 16 : vp + (k - NENC) * 16;
        if (DMA) {
            const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(encw + k * 1024);
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(lds_addr)
                : "memory");
        } else {
            *reinterpret_cast<float4*>(encw + k * 1024 + lane16) = *reinterpret_cast<const float4*>(src);
        }
    }
}

// Fused path: the integrated positional encoding of this lane's sample (cast_rays + conical_frustum_to_gaussian +
// integrated_pos_enc, models/mip.py:50-103, 322-350) computed in registers and written straight into the wave-private
// LDS fragments -- no [M,96] round trip through HBM, no separate kernel.  Same device functions and operation order as
// k_cast_ipe (kernels_ray.hip; this unit is compiled with -ffp-contract=off too), so both paths give the same bits.
// Lane (hi, n) owns features f = ks*16 + hi*8 + j of sample n for the three "sin" k-steps ks = 0..2; feature f + 48 (the
// "cos" half, k-step ks + 3) has the same (degree, axis), so y and the damping factor are shared.
struct RayIn {
    const float* t;          // [B, N+1]
    const float* origins;    // [B, 3]
    const float* dirs;
    const float* radii;      // [B]
    int min_deg;
    int disable_integration;
};
template <int NENC>
__device__ __forceinline__ void ipe_to_lds(const RayIn& R, int64_t sc, int num_samples, int hi, char* enc_lane_w) {
    static_assert(NENC == 6, "generated for 16 degrees = 96 features");
    const int64_t b = sc / num_samples;
    const int i = (int)(sc - b * num_samples);
    const float d[3] = {R.dirs[b * 3], R.dirs[b * 3 + 1], R.dirs[b * 3 + 2]};
    const float o[3] = {R.origins[b * 3], R.origins[b * 3 + 1], R.origins[b * 3 + 2]};
    const float t0 = R.t[b * (num_samples + 1) + i], t1 = R.t[b * (num_samples + 1) + i + 1];
    Gauss3 g = conical_frustum_to_gaussian(t0, t1, d, o, R.radii[b]);
    if (R.disable_integration) g.cov[0] = g.cov[1] = g.cov[2] = 0.0f;
    // scalars + compile-time picks: a select between array ELEMENTS would be turned into a dynamically indexed stack array
    const float mx = g.mean[0], my = g.mean[1], mz = g.mean[2], cx = g.cov[0], cy = g.cov[1], cz = g.cov[2];
    auto pick = [](int a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); };
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        bf16x8 fs, fc;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f0 = ks * 16 + j, f1 = f0 + 8;              // lane-half 0 / 1 (compile-time after unrolling)
            const int l = hi ? f1 / 3 : f0 / 3;
            const float m = hi ? pick(f1 % 3, mx, my, mz) : pick(f0 % 3, mx, my, mz);
            const float cv = hi ? pick(f1 % 3, cx, cy, cz) : pick(f0 % 3, cx, cy, cz);
            const float scale = (float)(1u << (l + R.min_deg));
            const float y = m * scale;
            const float yv = cv * (scale * scale);
            const float damp = exp_fast(-0.5f * yv);
            fs[j] = (__bf16)(damp * sin_fast(y));
            fc[j] = (__bf16)(damp * sin_fast(y + kHalfPiF));
            if (j == 3) __builtin_amdgcn_sched_barrier(0);
        }
        *reinterpret_cast<bf16x8*>(enc_lane_w + ks * 1024) = fs;
        *reinterpret_cast<bf16x8*>(enc_lane_w + (ks + 3) * 1024) = fc;
        __builtin_amdgcn_sched_barrier(0);     // one k-step at a time: keeps the 48 fp64 range reductions from all being live at once
    }
}

// The same encoding in PIECES (gen_mlp_bf16.IPE_SHADOW): the generated tile body spreads them over the MFMA slots that follow the
// last layer reading the encoding, computing the NEXT tile's fragments while this tile's layers 6.. run.  Expression for
// expression the body of ipe_to_lds above, so both routes write the same bits.
struct IpeNext { float mx, my, mz, cx, cy, cz; };
__device__ __forceinline__ void ipe_next_gauss(const RayIn& R, int64_t sc, int num_samples, IpeNext& q) {
    const int64_t b = sc / num_samples;
    const int i = (int)(sc - b * num_samples);
    const float d[3] = {R.dirs[b * 3], R.dirs[b * 3 + 1], R.dirs[b * 3 + 2]};
    const float o[3] = {R.origins[b * 3], R.origins[b * 3 + 1], R.origins[b * 3 + 2]};
    const float t0 = R.t[b * (num_samples + 1) + i], t1 = R.t[b * (num_samples + 1) + i + 1];
    Gauss3 g = conical_frustum_to_gaussian(t0, t1, d, o, R.radii[b]);
    if (R.disable_integration) g.cov[0] = g.cov[1] = g.cov[2] = 0.0f;
    q.mx = g.mean[0]; q.my = g.mean[1]; q.mz = g.mean[2]; q.cx = g.cov[0]; q.cy = g.cov[1]; q.cz = g.cov[2];
}
template <int KS, int J>
__device__ __forceinline__ void ipe_next_yd(const RayIn& R, const IpeNext& q, int hi, float& y, float& damp) {
    auto pick = [](int a, float x, float yy, float z) { return a == 0 ? x : (a == 1 ? yy : z); };
    constexpr int f0 = KS * 16 + J, f1 = f0 + 8;              // lane-half 0 / 1
    const int l = hi ? f1 / 3 : f0 / 3;
    const float m = hi ? pick(f1 % 3, q.mx, q.my, q.mz) : pick(f0 % 3, q.mx, q.my, q.mz);
    const float cv = hi ? pick(f1 % 3, q.cx, q.cy, q.cz) : pick(f0 % 3, q.cx, q.cy, q.cz);
    const float scale = (float)(1u << (l + R.min_deg));
    y = m * scale;
    const float yv = cv * (scale * scale);
    damp = exp_fast(-0.5f * yv);
}
template <int J>
__device__ __forceinline__ void ipe_next_sc(float y, float damp, bf16x8& fs, bf16x8& fc) {
    fs[J] = (__bf16)(damp * sin_fast(y));
    fc[J] = (__bf16)(damp * sin_fast(y + kHalfPiF));
}
template <int KS>
__device__ __forceinline__ void ipe_next_store(const bf16x8& fs, const bf16x8& fc, char* enc_lane_w) {
    *reinterpret_cast<bf16x8*>(enc_lane_w + KS * 1024) = fs;
    *reinterpret_cast<bf16x8*>(enc_lane_w + (KS + 3) * 1024) = fc;
}

// Ring-group boundary, executed when the A-fragment LOAD cursor enters group g:
// (1) this wave's share of group g has landed (vmcnt) and all its LDS reads of group g-1 have
//     returned (lgkmcnt); (2) barrier: both now hold for every wave, so group g is readable and
//     the slot of group g-1 is free; (3) prefetch group g+1 (or group 0 of the next tile) there.
#define GROUP_BEGIN(g, nslot)                                                                     \
    do {                                                                                          \
        if (DMA) asm volatile("WAIT_INSN\n\tBARRIER_INSN" ::: "memory");          \
        else __syncthreads();                                                                     \
        if ((g) + 1 < kNumGroups) issue_group<DMA>(stream, smem, (g) + 1, (nslot), wave, lane16); \
        else if (has_next) issue_group<DMA>(stream, smem, 0, (nslot), wave, lane16);              \
    } while (0)
"""

DEEP_RING_MACRO = r"""// One wave per SIMD (the 512-wide trunk, gen_mlp_bf16.waves_of): a ring group is 16 chunks = 16 of a wave's MFMAs = ~0.5 us, less than the
// L2 -> LDS latency of the group behind it, and there is no second wave to cover the wait.  These kernels keep kAhead groups in flight in a
// ring of kAhead + 1 slots: entering group g waits, with a COUNTED vmcnt, for this wave's four DMAs of group g only (the 4 (kAhead - 1) DMAs
// of the groups behind it stay in flight), and refills the slot of group g - 1 with group g + kAhead.  Group 0 of a tile waits for
// everything: the tile's encodings were issued behind the ring DMAs and are read right after.
// The refill runs AROUND the stream unconditionally: on a workgroup's last tile the last kAhead groups prefetch groups 0 .. kAhead - 1 once
// more (48 KiB of L2 reads nobody consumes) so that the counted wait holds at every boundary -- with a refill that stopped at the last
// tile only 4 (kNumGroups - 1 - g) DMAs would be younger than group g's and vmcnt(4 (kAhead - 1)) would pass before group g has landed
// (ADVICE r05).  The kernel ends with vmcnt(0): no LDS-DMA may land after the workgroup has released its LDS.
#define GROUP_BEGIN_DEEP(g, WAITCNT)                                                                                     \
    do {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WAITCNT) : "memory");                          \
        if ((g) + kAhead < kNumGroups) issue_group<DMA>(stream, smem, (g) + kAhead, ((g) + kAhead) % kSlots, wave, lane16);      \
        else issue_group<DMA>(stream, smem, (g) + kAhead - kNumGroups, ((g) + kAhead) % kSlots, wave, lane16);                  \
    } while (0)
// kSlots = kAhead + 2 (MLP_WIDE_SPARE_SLOT=1): group g + kAhead goes into the slot of group g - 2, whose LDS reads returned long ago (every
// MFMA slot waits for all but its last PREFETCH - 1 reads), so the boundary need not drain this wave's in-flight A-fragment reads of group g - 1
#define GROUP_BEGIN_DEEP_NODRAIN(g, WAITCNT)                                                                             \
    do {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WAITCNT) : "memory");                                     \
        if ((g) + kAhead < kNumGroups) issue_group<DMA>(stream, smem, (g) + kAhead, ((g) + kAhead) % kSlots, wave, lane16);      \
        else issue_group<DMA>(stream, smem, (g) + kAhead - kNumGroups, ((g) + kAhead) % kSlots, wave, lane16);                  \
    } while (0)
"""


FUSED_MACROS = r"""// ---- one-kernel form of a wide encoding (Plan.fused) ----
// One k-step of the encoding (1 KiB fragment of this wave tile: 16 bytes per lane) from global memory into the wave-private LDS ring.
// Wave-private: the wave that issues the DMA is the wave that reads the slot, so its own counted vmcnt orders the read behind the landing
// (no barrier).  Uniform source base + the 32-bit lane offset (saddr form), like issue_group.
__device__ __forceinline__ void enc_dma(const char* gsrc, char* ldst, unsigned lane16) {
    asm volatile("" : "+s"(gsrc));      // opaque: one scalar add per DMA instead of a hoisted 64-bit base per k-step
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldst;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(gsrc), "s"(lds_addr)
        : "memory");
}
#define ENC_DMA(i, base, goff, loff) enc_dma((base) + (goff), encw + (loff), lane16)
// a wave-uniform pointer the compiler can keep in SGPRs (the tile index is loop-carried: its divergence analysis gives up on it)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)p), hi_ = __builtin_amdgcn_readfirstlane((unsigned)((size_t)p >> 32));
    return reinterpret_cast<const char*>(((size_t)hi_ << 32) | lo);
}
// Ring-group boundary with a COUNTED vmcnt: K = the vector-memory operations this wave has issued since its DMAs of group g (the encoding
// DMAs of the k-steps in between), which stay in flight; everything older -- group g included -- has landed when the wait returns.
#define GROUP_BEGIN_CNT(g, nslot, K)                                                              \
    do {                                                                                          \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K) : "memory");          \
        if ((g) + 1 < kNumGroups) issue_group<DMA>(stream, smem, (g) + 1, (nslot), wave, lane16); \
        else if (has_next) issue_group<DMA>(stream, smem, 0, (nslot), wave, lane16);              \
    } while (0)
"""


def count_waits(lines, start, cpw, ngroups):
    """Fused form: fill in the counted vmcnt of every GROUP_BEGIN_CNT and RING_WAIT of the tile body lines[start:] (program order = text
    order: the body is straight-line code between sched_barriers, its DMAs are asm volatile).  Model: the wave's vector-memory operations
    retire in issue order; a wait for operation x with K younger operations issued since is `vmcnt(K)`.  GROUP_BEGIN(0) at the tile start
    waits for everything (the previous tile's stores, this tile's view encoding, the first encoding k-steps)."""
    log = []                 # tags of the operations issued since the last full wait, in issue order
    unknown = False          # a conditionally issued DMA is in the log: no counted wait may follow before the next full wait
    tok = re.compile(r"GROUP_BEGIN\(0, \d+\)|GROUP_BEGIN_CNT\((\d+), (\d+), @VM@\)|ENC_DMA\((\d+),|RING_WAIT\((\d+)\); ")
    kmax = 0

    def wait_for(tag):
        nonlocal log, kmax
        assert not unknown, "counted wait behind a conditionally issued DMA"
        if tag not in log:
            return None                      # issued before the last full wait: landed
        last = max(i for i, t in enumerate(log) if t == tag)
        k = len(log) - 1 - last
        assert k <= 56, k
        kmax = max(kmax, k)
        log = log[last + 1:]
        return k
    for li in range(start, len(lines)):
        line = lines[li]
        out, pos = [], 0
        for m in tok.finditer(line):
            out.append(line[pos:m.start()])
            pos = m.end()
            t = m.group(0)
            if t.startswith("GROUP_BEGIN(0"):
                log, unknown = [], False     # full wait
                log += [("grp", 1)] * cpw
                out.append(t)
            elif t.startswith("GROUP_BEGIN_CNT"):
                g = int(m.group(1))
                k = wait_for(("grp", g))
                if k is None:                 # an earlier, stricter wait (a ring read behind a younger encoding DMA) has retired this group's DMAs already:
                    k = len(log)              # nothing to wait for -- whatever is in flight may stay in flight
                    assert k <= 56
                out.append(f"GROUP_BEGIN_CNT({g}, {m.group(2)}, {k})")
                if g + 1 < ngroups:
                    log += [("grp", g + 1)] * cpw
                else:
                    unknown = True            # the tile's last boundary issues the NEXT tile's group 0 only when there is one
            elif t.startswith("ENC_DMA"):
                log.append(("enc", int(m.group(3))))
                out.append(t)
            else:                             # RING_WAIT(i);
                k = wait_for(("enc", int(m.group(4))))
                out.append("" if k is None else f'asm volatile("s_waitcnt vmcnt({k})" ::: "memory"); ')
        out.append(line[pos:])
        lines[li] = "".join(out)
    return kmax


def _shadow_fits(plan: Plan) -> bool:
    """Three ops behind the last encoding reader with >= 2 x 18 MFMA slots each (gen_kernel places one encoding k-step per op)."""
    last = max(i for i, op in enumerate(plan.ops) if any(sg.regset == "enc" for sg in op.segs))
    tail = plan.ops[last + 1:last + 4]
    return len(tail) == 3 and all(op.nk * len(op.tiles) // 2 >= 18 for op in tail)


def waves_of(arch: Arch) -> int:
    """Wavefronts per workgroup.  A layer's input AND output activations live in registers (2 x width/16 k-step fragments of 4 VGPRs): up to 256
    wide that is 128 of the 256 registers a wave has at two waves per SIMD; a 512-wide trunk needs 256 for the activations alone, so its kernel
    runs ONE wave per SIMD (4-wave workgroups of 128 samples, one per CU) with the 512-register budget of that occupancy (arch + acc VGPRs)."""
    return WAVES if max(arch.net_width, arch.net_width_condition) <= 256 else 4


def gen_kernel(plan: Plan, variant: int = 0) -> str:
    pre = plan.pre_gemm        # trunk of the two-kernel form (mlp_pre_plan.py): X preloaded, skip-layer accumulators from k_pre_gemm
    fused = plan.fused         # one-kernel form of the same variants: layer 0 and the skip layer as k-step-major ops over a streamed encoding
    ENC_WAVE_BYTES = (2 + FUSED_RING) * 1024 if fused else globals()["ENC_WAVE_BYTES"]       # fused: 2 KiB view encoding + the encoding ring
    assert not fused or (1 < FUSED_AHEAD <= FUSED_RING - 1)
    WAVES = waves_of(plan.arch)                 # (shadow the module defaults: everything below is per kernel)
    wide = max(plan.arch.net_width, plan.arch.net_width_condition) > 256      # one wave per SIMD, 512-register budget
    # chunks a wave DMAs per ring group.  The one-wave-per-SIMD kernels can take 8 (MLP_WIDE_CPW): a group is then 32 of a wave's MFMAs, so the
    # workgroup meets at a barrier half as often -- with one wave per SIMD nobody covers for a wave parked at the barrier
    PREFETCH = int(os.environ.get("MLP_WIDE_PREFETCH", globals()["PREFETCH"])) if wide else globals()["PREFETCH"]      # A-fragment read distance (chunks)
    CPW = int(os.environ.get("MLP_WIDE_CPW", "4")) if wide else 4
    assert CPW in (4, 8)
    GROUP = CPW * WAVES
    WG_PER_CU = 1 if wide else 8 // WAVES
    nreg = max(plan.arch.net_width, plan.arch.net_width_condition) // 16      # k-step fragments of one activation register set
    # Round 6 A/B on the 512-wide trunk (profiles/r06c_w512_group_prefetch_ab.txt, r06d_w512_spare_slot_ab.txt; ms per 524,288 samples): default of round 5
    # (3 groups in flight, 4 slots) 2.085-2.121; 32-chunk groups (MLP_WIDE_CPW=8: half the barriers) 2.14; A fragments 6 / 8 chunks ahead 2.10 / 2.11;
    # a spare slot (no LDS drain at the boundaries) with 2 / 4 groups in flight 2.088-2.093 / 2.096-2.104.  None of the ring's knobs is worth more
    # than 1.3 %: at 128 samples per workgroup the kernel pulls its 4.5-MiB weight stream through L2 -> LDS at 9.6 TB/s chip-wide, which is what bounds it.
    AHEAD = int(os.environ.get("MLP_WIDE_AHEAD", "2")) if wide else 1         # ring groups in flight (GROUP_BEGIN_DEEP); the 8-wave kernels: 1
    SPARE = wide and os.environ.get("MLP_WIDE_SPARE_SLOT", "1") == "1"           # one more ring slot than groups in flight: no LDS drain at the group boundaries
    SLOTS = (AHEAD + (2 if SPARE else 1)) if wide else globals()["SLOTS"]
    sfx = f"_pre_v{variant}" if pre else (f"_fused_v{variant}" if fused else ("" if variant == 0 else f"_v{variant}"))
    nchunks = len(plan.chunks)
    assert nchunks % GROUP == 0, "stream must be a whole number of ring groups"
    ngroups = nchunks // GROUP
    assert ngroups % SLOTS == 0, "tile-to-tile ring phase must be stable"
    a = plan.arch
    nenc = 0 if (pre or fused) else a.xyz_dim // 16
    nk_enc = a.xyz_dim // 16      # (fused: encoding k-steps of one pass over the fragment run of a wave tile)
    assert (nenc + 2) * 1024 <= ENC_WAVE_BYTES
    nbias_bytes = plan.n_tiles * 128
    ring_bytes = SLOTS * GROUP * CHUNK_BYTES
    enc_off = (ring_bytes + nbias_bytes + 1023) // 1024 * 1024
    lds_bytes = enc_off + WAVES * ENC_WAVE_BYTES
    assert lds_bytes <= 160 * 1024
    panels, slots = build_schedule(plan)
    nreal = len(slots)
    # padding: inside the last ring group, or exactly one whole group of zeros, begun by an extra GROUP_BEGIN at the tile end (trunk plans; two view layers)
    # (4-wave kernels: the stream is padded to mlp_plan.RING_MULTIPLE = 64 chunks = four of their groups; every unentered padding group gets its GROUP_BEGIN)
    assert nreal == plan.n_real_chunks and (nchunks - nreal < GROUP or (nchunks - nreal == GROUP and nreal % GROUP == 0) or wide)
    lines = []
    e = lines.append
    e("// AUTO-GENERATED by gen_mlp_bf16.py from mlp_plan.py -- do not edit by hand.")
    e("// Register-resident bf16 MFMA MLP of Mip-NeRF (reference: models/mip_nerf.py:75-111 +")
    e("// activations 236-238).  See mlp_plan.py for the dataflow; DESIGN.md for the roofline.")
    e("#include <hip/hip_runtime.h>")
    e('#include "kernels.hpp"')
    e('#include "raymath.hpp"')
    e("namespace mip {")
    if variant:
        a_ = plan.arch
        e(f"// architecture variant {variant}: depth {a_.net_depth} width {a_.net_width} cond {a_.net_depth_condition}x{a_.net_width_condition} "
          f"use_viewdirs={int(a_.use_viewdirs)}" + (" -- TRUNK of the two-kernel form (layers 1.., mlp_pre_plan.py)" if pre else "")
          + (" -- ONE-kernel form of the wide encoding (Plan.fused)" if fused else ""))
        e(f"namespace v{variant}{'pre' if pre else ('fused' if fused else '')} {{")
    e("typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;")
    e("typedef __attribute__((ext_vector_type(16))) float f32x16;")
    if pre and ABLATE_PRELOADS:
        e("extern __shared__ __attribute__((aligned(16))) char smem_probe[];")
        e("typedef __attribute__((ext_vector_type(4))) float f32x4p_;")
        e("__device__ __forceinline__ bf16x8 pre_fake(const bf16x8* p) { return *reinterpret_cast<const bf16x8*>(smem_probe + ((unsigned)(size_t)p & 0x7ff0u)); }")
        e(f"__device__ __forceinline__ f32x4p_ pre_fake(const f32x4p_* p) {{ return *reinterpret_cast<const f32x4p_*>(smem_probe + {ring_bytes} + ((unsigned)(size_t)p & 0x1ff0u)); }}")
        e("#define PRE_LD(ptr) pre_fake(ptr)")
    elif pre and PRE_NT:
        e("#define PRE_LD(ptr) __builtin_nontemporal_load(ptr)")
    # (fused: 29 instead of 136 SGPRs spilled to VGPR lanes and -0.3 % per forward in three alternating pairs, profiles/r06g_fused360_ab.txt;
    #  the ring geometry -- 8 slots / 7 ahead, 8 / 5, 6 / 5 -- is worth nothing: 6.60-6.61 ms all)
    if wide or (fused and os.environ.get("MLP_FUSED_OPAQUE", "1") == "1"):
        e("#define MIP_OPAQUE_STREAM_BASE 1")
    e(f"constexpr int kRingBytes = {ring_bytes};")
    e(f"constexpr int kBiasBytes = {nbias_bytes};")
    e(f"constexpr int kEncOff = {enc_off};")
    e(f"constexpr int kEncWaveBytes = {ENC_WAVE_BYTES};")
    e(f"constexpr int kLdsBytes = {lds_bytes};")
    e(f"constexpr int kGroupBytes = {GROUP * CHUNK_BYTES};")
    if CPW != 4:
        e(f"#define MIP_CHUNKS_PER_WAVE {CPW}")
    e(f"constexpr int kNumGroups = {ngroups};")
    e(f"constexpr int kTileSamples = {WAVES * 32};")
    if wide:
        e(f"constexpr int kAhead = {AHEAD};")
        e(f"constexpr int kSlots = {SLOTS};")
    e(KERNEL_PREAMBLE.replace("BARRIER_INSN", "s_nop 0" if ABLATE_BARRIER else "s_barrier")
      .replace("WAIT_INSN", "s_waitcnt lgkmcnt(0)" if ABLATE_WAIT else "s_waitcnt vmcnt(0) lgkmcnt(0)"))
    if wide:
        e(DEEP_RING_MACRO)
    if fused:
        e(FUSED_MACROS)
    e("template <bool DMA, bool IPE>")
    e(f"__global__ void __launch_bounds__({WAVES * 64}, {1 if wide else 2})")
    e("k_mlp_bf16(const char* __restrict__ stream, const float* __restrict__ bias_tab,")
    if pre or fused:
        e("           const char* __restrict__ pre_x, const char* __restrict__ pre_acc,       // fused: pre_x = the encoding's fragment buffer, pre_acc unused")
    e("           const __bf16* __restrict__ enc, const __bf16* __restrict__ viewenc, float4* __restrict__ rgb_sigma,")
    e("           float4* __restrict__ raw_out, int64_t M, int num_samples, int ntiles, float density_bias,")
    e("           float rgb_padding, RayIn rin, const float* __restrict__ dnoise, float dnoise_scale) {")
    e("    extern __shared__ __attribute__((aligned(16))) char smem[];")
    e("    const int tid = threadIdx.x;")
    e("    const int lane = tid & 63;")
    e("    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);")
    e("    const int hi = lane >> 5, n = lane & 31;")
    e("    const unsigned lane16 = (unsigned)lane * 16u;")
    e("    const char* ring_lane = smem + lane16;")
    e("    const char* bias_lane = smem + kRingBytes + hi * 64;")
    e("    char* encw = smem + kEncOff + wave * kEncWaveBytes;     // wave-private (uniform base)")
    e("    const char* enc_lane = encw + lane16;")
    e("    for (int i = tid; i < kBiasBytes / 16; i += blockDim.x)")
    e("        reinterpret_cast<float4*>(smem + kRingBytes)[i] = reinterpret_cast<const float4*>(bias_tab)[i];")
    e("    __syncthreads();")
    if SETPRIO:
        e("    if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // MI355X_MICROARCH.md, two waves per SIMD, item 4")
    shadow = IPE_SHADOW and nenc == 6 and _shadow_fits(plan)
    if wide:
        e("    if ((int)blockIdx.x < ntiles) {")
        for g in range(AHEAD):
            e(f"        issue_group<DMA>(stream, smem, {g}, {g}, wave, lane16);")
        e("    }")
    elif fused:
        e("    if ((int)blockIdx.x < ntiles) {")
        e("        issue_group<DMA>(stream, smem, 0, 0, wave, lane16);")
        e(f"        const char* encb = uniform_ptr(pre_x + ((int64_t)blockIdx.x * {WAVES} + wave) * {nk_enc * 1024});      // the first tile's first encoding k-steps")
        e("        PROLOGUE_ENC_DMAS")
        e("    }")
    else:
        e("    if ((int)blockIdx.x < ntiles) issue_group<DMA>(stream, smem, 0, 0, wave, lane16);")
    if shadow:
        e("    if (IPE && (int)blockIdx.x < ntiles) {      // the first tile's encoding; every later one is computed in the previous tile's shadow")
        e("        const int64_t s_first = (int64_t)blockIdx.x * kTileSamples + wave * 32 + n;")
        e(f"        ipe_to_lds<{nenc}>(rin, s_first < M ? s_first : M - 1, num_samples, hi, encw + lane16);")
        e("    }")
    e("    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {")
    e("        const bool has_next = tile + (int)gridDim.x < ntiles;")
    e("        const int64_t s = (int64_t)tile * kTileSamples + wave * 32 + n;")
    e("        const int64_t sc = s < M ? s : M - 1;")
    e("        const int64_t ray = sc / num_samples;")
    if shadow:
        e("        const int64_t s_next = (int64_t)(tile + (int)gridDim.x) * kTileSamples + wave * 32 + n;")
        e("        const int64_t scn = s_next < M ? s_next : M - 1;     // past the end: a valid sample, its encoding is never used")
        e("        IpeNext ipn;")
        e("        float ipe_y = 0.0f, ipe_d = 0.0f;")
        e("        bf16x8 ipe_fs, ipe_fc;")
    if pre or fused:
        e("        issue_encodings<DMA, 0, 0>(nullptr, viewenc + ray * 32 + hi * 8, encw, lane16);")
    else:
        e("        if (IPE) {")
        e(f"            issue_encodings<DMA, {nenc}, {nenc}>(nullptr, viewenc + ray * 32 + hi * 8, encw, lane16);")
        if not shadow:
            e(f"            ipe_to_lds<{nenc}>(rin, sc, num_samples, hi, encw + lane16);")
        e("        } else {")
        e(f"            issue_encodings<DMA, {nenc}, 0>(enc + sc * {a.xyz_dim} + hi * 8, viewenc + ray * 32 + hi * 8, encw, lane16);")
        e("        }")
    e(f"        bf16x8 X[{nreg}], Y[{nreg}], " + ", ".join(f"A{i}" for i in range(PREFETCH)) + ", E0, E1, E2;")
    if pre:
        e("        // what k_pre_gemm left for this wave tile: X = bf16(relu(layer 0)) as 16 lane-linear fragments, the skip layer's accumulator images")
        e(f"        const int64_t wt = (int64_t)tile * {WAVES} + wave;")
        e("        const char* prex_lane = pre_x + wt * 16384 + lane16;")
        e("        const char* pre_lane = pre_acc + wt * 32768 + lane16;")
        for k in range(16):
            e(f"        X[{k}] = PRE_LD(reinterpret_cast<const bf16x8*>(prex_lane + {k * 1024}));")
    e("        f32x16 acc00, acc01, acc10, acc11;")
    if fused:
        e("        f32x16 " + ", ".join(f"accW{t}" for t in range(a.net_width // 32)) + ";      // the k-step-major ops: all output tiles of a layer at once")
        e(f"        const char* encb = uniform_ptr(pre_x + ((int64_t)tile * {WAVES} + wave) * {nk_enc * 1024});                     // this wave tile's fragment run")
        e(f"        const char* encb_next = uniform_ptr(has_next ? pre_x + ((int64_t)(tile + (int)gridDim.x) * {WAVES} + wave) * {nk_enc * 1024} : encb);")
    e("        float raw_density = 0.0f, raw_r = 0.0f, raw_g = 0.0f, raw_b = 0.0f;")

    def lda(c):
        slot = (c // GROUP) % SLOTS
        off = slot * GROUP * CHUNK_BYTES + (c % GROUP) * CHUNK_BYTES
        if ABLATE_LDA and (c // PREFETCH) % (ABLATE_LDA + 1) != 0 and c >= PREFETCH:
            return f"// (ablated) A{c % PREFETCH} = LDA({off});"
        return f"A{c % PREFETCH} = LDA({off});"

    # ---- assign E registers to LDS-resident B operands and place their loads -----------------
    side = {c: [] for c in range(nreal)}
    prologue = []
    ecount = 0
    b_expr = [None] * nreal
    cur_e = None
    for c, sl in enumerate(slots):
        kind, val = sl["b"]
        if kind == "reg":
            b_expr[c] = val
            continue
        if sl["first_of_ks"]:
            cur_e = f"E{ecount % NE}"
            ecount += 1
            spk = panels[sl["panel"]]["spk"]
            at = c - 2 * spk          # two k-steps ahead; the register's previous user is 3 k-steps back
            if kind == "ring":        # k-step `val` of the streamed encoding: its DMA must have landed (counted wait, filled in below)
                stmt = f"RING_WAIT({val}); {cur_e} = LDB({2048 + (val % FUSED_RING) * 1024});"
            else:
                stmt = f"{cur_e} = LDB({val});"
            if at < 0:
                prologue.append(stmt)
            else:
                side[at].append(stmt)
        b_expr[c] = cur_e
    # sanity: an E load at slot `at` (emitted after that slot's MFMA) must not precede a pending user
    last_use = {}
    for c in range(nreal):
        if slots[c]["b"][0] in ("lds", "ring"):
            last_use[b_expr[c]] = c
        for stmt in side[c]:
            m_e = re.search(r"(?:^|; )(E\d) = LDB", stmt)
            if m_e:
                reg = m_e.group(1)
                # every user of the PREVIOUS value of reg must be <= c
                prev_users = [u for u in range(c + 1, nreal) if slots[u]["b"][0] in ("lds", "ring") and b_expr[u] == reg]
                first_new = prev_users[0] if prev_users else None
                assert first_new is None or first_new > c
                assert last_use.get(reg, -1) <= c

    # ---- epilogue of the previous panel / bias of the next panel, spread over this panel ----------
    for pi, pn in enumerate(panels):
        work = []
        nxt = panels[pi + 1] if pi + 1 < len(panels) else None
        if pn.get("wide"):
            # A wide panel keeps 8 accumulator tiles (128 registers) live: its predecessor's epilogue and its own bias loads run between
            # the predecessor's last MFMA and its first one (a bubble of ~50 VALU / LDS instructions per wide op, two per tile), and the
            # bias loads of the panel behind it wait for its last MFMA (the input set is dead from there on)
            if pi > 0:
                side[pn["first"] - 1] += epilogue_pieces(plan, panels[pi - 1]) + bias_pieces(plan, pn)
            if nxt is not None:
                side[pn["first"] + pn["n"] - 1] += bias_pieces(plan, nxt)
        else:
            if pi > 0:
                work += epilogue_pieces(plan, panels[pi - 1])
            if nxt is not None and not nxt.get("wide"):
                work += bias_pieces(plan, nxt)
        n = pn["n"]
        for wi, stmt in enumerate(work):
            at = min(2 + wi, n - 1)     # start two slots in: the previous panel's last MFMAs have drained
            # ... but never behind the first slot of THIS panel that reads the register the statement writes (narrow layers: a
            # one-panel producer's activations are needed from slot 0 on); side[c] is emitted after the MFMA of slot c
            m = _WRITES.search(stmt)
            if m:
                readers = [c for c in range(pn["first"], pn["first"] + n) if b_expr[c] == m.group(1)]
                if readers:
                    at = min(at, readers[0] - pn["first"] - 1)
            side[pn["first"] + at].append(stmt)       # at == -1: right behind the last MFMA of the producing panel
    _check_hazards(plan, panels, slots, b_expr, side)

    # ---- fused form: DMA of every encoding k-step into the wave-private ring, FUSED_AHEAD k-steps ahead of its MFMAs ------------------
    # ring sequence i = 0 .. nring - 1 over the tile (layer 0, then the skip layer); slot i % FUSED_RING; source = fragment i % nk_enc of
    # the wave tile.  DMA(i) goes behind the first MFMA of sequence i - FUSED_AHEAD (by then sequence i - FUSED_RING has been consumed);
    # the first FUSED_AHEAD of a tile are issued by the PREVIOUS tile right behind its last ring k-step (kernel prologue for the first tile).
    enc_prologue = []
    if fused:
        first_slot = {}
        for c, sl in enumerate(slots):
            if sl["b"][0] == "ring" and sl["first_of_ks"]:
                first_slot[sl["b"][1]] = c
        nring = len(first_slot)
        assert nring % nk_enc == 0 and nring >= FUSED_RING
        for i in range(nring):
            stmt = f"ENC_DMA({i}, encb, {(i % nk_enc) * 1024}, {2048 + (i % FUSED_RING) * 1024});"
            if i < FUSED_AHEAD:
                enc_prologue.append(stmt)
            else:
                side[first_slot[i - FUSED_AHEAD]].append(stmt)
        last_ring = max(c for c, sl in enumerate(slots) if sl["b"][0] == "ring")
        for i in range(FUSED_AHEAD):      # the next tile's first k-steps (a harmless re-read of this tile's on the last tile)
            side[last_ring].append(f"ENC_DMA({nring + i}, encb_next, {i * 1024}, {2048 + (i % FUSED_RING) * 1024});")

    # ---- the next tile's integrated positional encoding, in pieces, behind the last reader of this tile's encoding ----
    # One k-step (8 feature pairs + its LDS store) per op, in the FIRST HALF of that op's slots: there most of the op's output
    # registers are still dead, so the pieces' temporaries (y, damping, two fragments) fit the 256-VGPR budget without spills;
    # only the six Gaussian moments stay live from the first piece to the last.
    if shadow:
        last_enc = max(c for c, sl in enumerate(slots) if sl["b"][0] == "lds" and sl["b"][1] < nenc * 1024)
        tail_ops = []
        for pn in panels:
            op = pn["op"]
            if pn["first"] > last_enc + 2 and (not tail_ops or tail_ops[-1][0] is not op):
                tail_ops.append([op, pn["first"], 0])
            if tail_ops and tail_ops[-1][0] is op:
                tail_ops[-1][2] = pn["first"] + pn["n"] - tail_ops[-1][1]
        per_k = []
        for ks in range(3):
            ps = []
            for j in range(8):
                ps.append(f"ipe_next_yd<{ks}, {j}>(rin, ipn, hi, ipe_y, ipe_d);")
                ps.append(f"ipe_next_sc<{j}>(ipe_y, ipe_d, ipe_fs, ipe_fc);")
            ps.append(f"ipe_next_store<{ks}>(ipe_fs, ipe_fc, encw + lane16);")
            per_k.append(ps)
        per_k[0].insert(0, "ipe_next_gauss(rin, scn, num_samples, ipn);")
        regions = [(first, n // 2) for _, first, n in tail_ops[:3]]
        assert len(regions) == 3 and all(r[1] >= len(per_k[0]) for r in regions), \
            "IPE_SHADOW needs three ops behind the skip layer with >= 36 MFMA slots each (set MLP_IPE_SHADOW=0 for this shape)"
        for (first, n), ps in zip(regions, per_k):
            stride = max(1, min(IPE_SHADOW_STRIDE, n // len(ps)))
            for pi_, stmt in enumerate(ps):
                side[first + pi_ * stride].append("if (IPE) { " + stmt + " }")

    # ---- tile prologue --------------------------------------------------------------------------
    def group_begin(g, note=""):
        if wide:          # group 0: a full wait (the tile's encoding DMAs are younger than the ring's and are read next)
            if SPARE and g != 0:
                return f"GROUP_BEGIN_DEEP_NODRAIN({g}, {CPW * (AHEAD - 1)});{note}"
            return f"GROUP_BEGIN_DEEP({g}, {0 if g == 0 else CPW * (AHEAD - 1)});{note}"
        if fused and g != 0:
            return f"GROUP_BEGIN_CNT({g}, {(g + 1) % SLOTS}, @VM@);{note}"      # counted vmcnt, filled in by count_waits below
        return f"GROUP_BEGIN({g}, {(g + 1) % SLOTS});{note}"
    body_start = len(lines)
    e(f"        {group_begin(0)}")
    for c in range(PREFETCH):
        e(f"        {lda(c)}")
    for stmt in prologue:
        e(f"        {stmt}")
    for stmt in bias_pieces(plan, panels[0]):
        e(f"        {stmt}")
    e("        PIN();")
    cur_op = None
    for c, sl in enumerate(slots):
        pn = panels[sl["panel"]]
        if pn["op"] is not cur_op:
            cur_op = pn["op"]
            e(f"        // ---- {cur_op.name}: K = {cur_op.nk} k-steps, {len(cur_op.tiles)} out tiles, relu={int(cur_op.relu)} -> {cur_op.out}")
        e(f"        MFMA({sl['acc']}, A{c % PREFETCH}, {b_expr[c]});")
        lc = c + PREFETCH
        if lc < nreal:
            if lc % GROUP == 0:
                g = lc // GROUP
                e(f"        {group_begin(g)}")
            e(f"        {lda(lc)}")
        for stmt in side[c]:
            e(f"        {stmt}")
        e("        PIN();")
    for g in range((nreal + GROUP - 1) // GROUP, ngroups):
        pad_note = "      // a whole group of zero padding: nothing reads it, but its barrier issues the next tile's group 0"
        e(f"        {group_begin(g, pad_note)}")
    for stmt in epilogue_pieces(plan, panels[-1]):
        e(f"        {stmt}")
    if fused:
        kmax = count_waits(lines, body_start, CPW, ngroups)      # fill in the counted waits of the tile body
        text = "\n".join(lines)
        assert "@VM@" not in text and "RING_WAIT" not in text
        for i, ln in enumerate(lines):
            if "PROLOGUE_ENC_DMAS" in ln:
                lines[i] = "\n".join("        " + st for st in enc_prologue)
    e("        if (hi == 0 && s < M) {")
    e("            // mip_nerf.py:232-233: raw_density += density_noise * randn (randomized training only), BEFORE the activation")
    e("            const float noisy_density = dnoise ? raw_density + dnoise_scale * dnoise[s] : raw_density;")
    e("            rgb_sigma[s] = make_float4(rgb_activation(raw_r, rgb_padding), rgb_activation(raw_g, rgb_padding),")
    e("                                       rgb_activation(raw_b, rgb_padding), density_activation(noisy_density, density_bias));")
    e("            if (raw_out) raw_out[s] = make_float4(raw_r, raw_g, raw_b, raw_density);")
    e("        }")
    e("    }")
    if wide or fused:
        e('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the refill runs around the stream: no LDS-DMA may land after the workgroup has released its LDS')
    e("}")
    e("")
    if variant:
        e(f"}}  // namespace v{variant}{'pre' if pre else ('fused' if fused else '')}")
        e(f"using namespace v{variant}{'pre' if pre else ('fused' if fused else '')};")
    else:
        e("int mlp_bf16_lds_bytes() { return kLdsBytes; }")
    e("")
    if pre or fused:
        e("// pre_x / pre_acc: the two outputs of launch_pre_gemm for the same M (16 KiB + 32 KiB per wave tile of 32 samples)")
        e("// (one-kernel form: pre_x = the encoding's fragment buffer -- whole 256-sample tiles, as k_cast_ipe_360 writes it --, pre_acc = nullptr)")
        e(f"hipError_t launch_mlp_bf16{sfx}(const void* stream_w, const float* bias_tab, const void* pre_x, const void* pre_acc, const void* viewenc,")
        e("                           float* rgb_sigma, float* raw_out, int64_t M, int num_samples, float density_bias,")
        e("                           float rgb_padding, int grid_limit, const float* dnoise, float dnoise_scale, hipStream_t st) {")
        e("    const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);")
        e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
        e("    if (grid < 1) grid = 1;")
        e("    static int attr_done[64] = {};")
        e("    int dev = 0;")
        e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
        e("    if (!attr_done[dev]) {")
        e("        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_bf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
        e("        if (er != hipSuccess) return er;")
        e("        attr_done[dev] = 1;")
        e("    }")
        e("    const RayIn rin = {nullptr, nullptr, nullptr, nullptr, 0, 0};")
        e("    hipLaunchKernelGGL((k_mlp_bf16<true, false>), dim3(grid), dim3(%d), kLdsBytes, st, (const char*)stream_w, bias_tab," % (WAVES * 64))
        e("                       (const char*)pre_x, (const char*)pre_acc, (const __bf16*)nullptr, (const __bf16*)viewenc, (float4*)rgb_sigma,")
        e("                       (float4*)raw_out, M, num_samples, ntiles, density_bias, rgb_padding, rin, dnoise, dnoise_scale);")
        e("    return hipGetLastError();")
        e("}")
        e("}  // namespace mip")
        return "\n".join(lines) + "\n"
    e(f"hipError_t launch_mlp_bf16{sfx}(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,")
    e("                           float* rgb_sigma, float* raw_out, int64_t M, int num_samples, float density_bias,")
    e("                           float rgb_padding, int grid_limit, bool dma, const RayInputs* rays, const float* dnoise,")
    e("                           float dnoise_scale, hipStream_t st) {")
    e("    const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);")
    e(f"    grid_limit *= {WG_PER_CU};      // workgroups per CU (LDS: {lds_bytes} B each)")
    e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
    e("    if (grid < 1) grid = 1;")
    e("    static int attr_done[64] = {};")
    e("    int dev = 0;")
    e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
    e("    if (!attr_done[dev]) {")
    e("        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_bf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        er = hipFuncSetAttribute((const void*)k_mlp_bf16<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        er = hipFuncSetAttribute((const void*)k_mlp_bf16<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        attr_done[dev] = 1;")
    e("    }")
    e("    RayIn rin = {nullptr, nullptr, nullptr, nullptr, 0, 0};")
    e("    if (rays) rin = RayIn{rays->t, rays->origins, rays->dirs, rays->radii, rays->min_deg, rays->disable_integration};")
    e("#define MIP_LAUNCH(D, I) hipLaunchKernelGGL((k_mlp_bf16<D, I>), dim3(grid), dim3(%d), kLdsBytes, st, (const char*)stream_w, \\" % (WAVES * 64))
    e("        bias_tab, (const __bf16*)enc, (const __bf16*)viewenc, (float4*)rgb_sigma, (float4*)raw_out, M, num_samples, ntiles, \\")
    e("        density_bias, rgb_padding, rin, dnoise, dnoise_scale)")
    e("    if (rays) MIP_LAUNCH(true, true);")
    e("    else if (dma) MIP_LAUNCH(true, false);")
    e("    else MIP_LAUNCH(false, false);")
    e("#undef MIP_LAUNCH")
    e("    return hipGetLastError();")
    e("}")
    e("}  // namespace mip")
    return "\n".join(lines) + "\n"


def gen_variants_header(n):
    """Declarations + dispatch table of the per-variant inference launchers (capi.hip indexes it with PlanDesc::variant)."""
    L = ["// AUTO-GENERATED by gen_mlp_bf16.py from VARIANTS -- do not edit by hand.", "#pragma once", '#include "kernels.hpp"',
         "namespace mip {",
         "typedef hipError_t (*LaunchBf16Fn)(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,",
         "                                   float* rgb_sigma, float* raw_out, int64_t M, int num_samples, float density_bias,",
         "                                   float rgb_padding, int grid_limit, bool dma, const RayInputs* rays, const float* dnoise,",
         "                                   float dnoise_scale, hipStream_t st);"]
    for vi in range(1, n):
        if not VARIANTS[vi].bf16_kernels:
            continue
        L.append(f"hipError_t launch_mlp_bf16_v{vi}(const void*, const float*, const void*, const void*, float*, float*, int64_t, int, float, float,")
        L.append("                               int, bool, const RayInputs*, const float*, float, hipStream_t);")
    names = ", ".join("nullptr /* fp32 only */" if not VARIANTS[vi].bf16_kernels else ("launch_mlp_bf16" if vi == 0 else f"launch_mlp_bf16_v{vi}")
                      for vi in range(n))
    L.append(f"static const LaunchBf16Fn kLaunchBf16[{n}] = {{{names}}};")
    L.append("}  // namespace mip")
    return "\n".join(L) + "\n"


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else HERE
    plans = [Plan.build(a) for a in VARIANTS]
    with open(os.path.join(outdir, "mlp_plan_gen.hpp"), "w") as f:
        f.write(gen_plan_header(plans))
    with open(os.path.join(outdir, "mlp_variants_gen.hpp"), "w") as f:
        f.write(gen_variants_header(len(plans)))
    for vi, plan in enumerate(plans):
        if not plan.arch.bf16_kernels:
            print(f"variant {vi}: plan tables only (fp32 kernels), {plan.n_real_chunks} chunks, {plan.n_tiles} tiles")
            continue
        with open(os.path.join(outdir, "mlp_bf16_gen.hip" if vi == 0 else f"mlp_bf16_gen_v{vi}.hip"), "w") as f:
            f.write(gen_kernel(plan, vi))
        print(f"generated variant {vi}: {plan.n_real_chunks} (+{len(plan.chunks) - plan.n_real_chunks} pad) chunks, {plan.n_tiles} tiles -> {outdir}")


if __name__ == "__main__":
    main()
