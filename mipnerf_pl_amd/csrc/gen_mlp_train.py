#!/usr/bin/env python3
"""Generate the bf16 TRAINING kernels of the Mip-NeRF MLP from mlp_train_plan.TrainPlan:

  mlp_bf16_trainfwd_gen.hip  k_mlp_bf16_trainfwd: the inference schedule (gen_mlp_bf16.py) + per output tile the
                             ReLU bit mask and the transposed activations (T-blocks) the weight-gradient kernel
                             consumes (see mlp_train_plan.py);
  mlp_bf16_dgrad_gen.hip     k_mlp_bf16_dgrad: delta_j = (W_{j+1}^T delta_{j+1}) * relu' with delta resident in
                             registers across all layers, W^T streamed through the same LDS ring, T-blocks of
                             every delta written for the weight-gradient kernel.

Both are straight-line code per 32-sample wave tile with compile-time ring offsets, exactly like the
inference kernel; the extra work of a tile (mask, two selection MFMAs, bf16 packing, 2 x 1 KiB stores) is
spread over the MFMA slots of the following panel.

Usage: python gen_mlp_train.py [outdir]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from mipnerf_pl_amd.mlp_plan import Plan  # noqa: E402
from mipnerf_pl_amd.mlp_train_plan import GROUP, SLOTS, TrainPlan  # noqa: E402
from gen_mlp_bf16 import KERNEL_PREAMBLE, SETPRIO  # noqa: E402

WAVES = 8
CHUNK_BYTES = 1024
PREFETCH = int(os.environ.get("MLP_TRAIN_PREFETCH", "4"))
ABLATE_TMFMA = os.environ.get("MLP_TRAIN_ABLATE_TMFMA", "0") == "1"    # timing experiment: no transposing MFMAs (wrong T-blocks)
# timing experiment: comma-separated forward op indices whose saved-activation T-blocks are neither transposed nor stored
# ("store every other layer"; the weight-gradient kernel would recompute them: MLP_WGRAD_RECOMPUTE_PROBE).  Wrong gradients.
SKIP_STORES = {int(x) for x in os.environ.get("MLP_TRAIN_SKIP_STORES", "0").split(",") if x.strip() and os.environ.get("MLP_TRAIN_SKIP_STORES", "0") != "0"}
NE = 3

TRAIN_PREAMBLE = r"""
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define TMFMA0(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), f32x16{0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, 0, 0, 0)
#define LDM(off) (*reinterpret_cast<const u32x4*>(priv_lane + (off)))

// Half of a T-block store: accumulator registers R0..R0+7 of a selection-MFMA result are 8 samples of one
// feature column (exact bf16 values); 64 lanes x 16 B = one lane-linear 1-KiB operand fragment.
template <int R0>
__device__ __forceinline__ void store_tfrag(const f32x16& acc, char* blk, unsigned lane16) {
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (__bf16)acc[R0 + r];
#ifdef MLP_TRAIN_ABLATE_STORES      // timing experiment: everything but the store itself (results are wrong)
    if (lane16 != 0xFFFFFFFFu) { asm volatile("" :: "v"(o)); return; }
#endif
    // streamed once, read back by another kernel after >2 GB of other traffic: keep it out of the L2's way
    // (measured on MI355X: -5 % forward-with-save, -7 % dgrad vs plain stores)
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(blk + (R0 / 8) * 1024 + lane16));
}

// 16 ReLU bits of one output tile from its two (post-ReLU, bf16) k-step registers: bit p = reg 2p > 0,
// bit 16+p = reg 2p+1 > 0 (mlp_train_plan.pack_mask).
// Plain 32-bit integer VALU on the packed pairs (never inline asm next to MFMAs: see relu1 in gen_mlp_bf16.py).  The
// halves are non-negative bf16 (v_max_f32 returns +0 for max(-0, +0)), so h + 0x7FFF has bit 15 set iff h != 0 and
// never carries into the neighbouring half.
__device__ __forceinline__ unsigned tile_mask(const bf16x8& a, const bf16x8& b) {
    const u32x4 da = __builtin_bit_cast(u32x4, a), db = __builtin_bit_cast(u32x4, b);
    unsigned m = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) m |= ((da[p] + 0x7FFF7FFFu) >> (15 - p)) & (0x00010001u << p);
#pragma unroll
    for (int p = 0; p < 4; ++p) m |= ((db[p] + 0x7FFF7FFFu) >> (11 - p)) & (0x00010001u << (4 + p));
    return m;
}

// dgrad epilogue of half a tile: fp32 -> bf16 (RNE), then AND with the expanded ReLU bits.
// `mw` = mask dword of the tile pair, SH = 8 * (tile & 1); pair p of the tile sits at bits (p, 16 + p) + SH.
template <bool MASK, int R0, int SH>
__device__ __forceinline__ void depilogue_half(const f32x16& acc, unsigned mw, bf16x8& o) {
    bf16x8 v;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (__bf16)acc[R0 + r];
    if (MASK) {
        u32x4 d = __builtin_bit_cast(u32x4, v);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned fl = (mw >> (SH + R0 / 2 + p)) & 0x00010001u;
            d[p] &= fl * 0xFFFFu;
        }
        v = __builtin_bit_cast(bf16x8, d);
    }
    o = v;
}

// Ring-group boundary with a COUNTED wait.  vmcnt retires in issue order (loads and stores share the counter on
// gfx9-family parts), and this wave issued exactly K stores (all unconditional) after its 4 DMAs of group g, so
// vmcnt(K) means "my share of group g has landed" without draining the T-block stores still in flight.
#define GROUP_BEGIN_K(g, nslot, K)                                                               \
    do {                                                                                          \
        asm volatile("s_waitcnt vmcnt(" #K ") lgkmcnt(0)\n\ts_barrier" ::: "memory");              \
        if ((g) + 1 < kNumGroups) issue_group<DMA>(stream, smem, (g) + 1, (nslot), wave, lane16); \
        else if (has_next) issue_group<DMA>(stream, smem, 0, (nslot), wave, lane16);              \
    } while (0)

// one 1-KiB lane-linear DMA (global -> wave-private LDS), per-lane 64-bit source address
__device__ __forceinline__ void dma_1k(const void* src_lane, char* lds_dst) {
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src_lane), "s"(lds_addr)
        : "memory");
}
"""


class Prog:
    """A straight-line tile program: slots (one MFMA each), per-slot side statements, panels."""

    def __init__(self):
        self.slots = []       # dict(acc, b, panel, ks, first_of_ks, zero)
        self.panels = []      # dict(first, n, pair, spk, post (list of stmts run during the next panel), pre)


_WRITES_REG = __import__("re").compile(r"epilogue_half<[^>]*>\(\w+, (?:[\w\[\]]+, )?([XY]\[\d+\])\)")


def _slot_b(sl):
    """B-operand expression of a slot: 'X[3]' / 'Y[0]' / 'R' / an E register (forward slots carry ('reg', expr) until assign_lds_b)."""
    if "bexpr" in sl:
        return sl["bexpr"]
    b = sl.get("b")
    return b[1] if isinstance(b, tuple) and b[0] == "reg" else None


def place_sides(prog, nchunks, which=""):
    """Spread each panel's `post` statements (and the next-next panel's `pre`) over the next panel's slots: the
    register work (epilogue, transposing MFMAs, masks) right away, one statement per slot; the T-block STORES evenly
    over the rest of the panel (all 8 waves of a workgroup run in lockstep, so back-to-back stores reach the CU's store
    path as one 32-KiB burst); then `pre` (accumulator re-initialisation of the panel after next, which must follow
    the stores that read those accumulators).  Measured on MI355X (scripts/prof_train.py): removing the store
    instructions altogether saves 0.18 ms of a 0.77 ms launch, but neither spreading them, nor counted vmcnt waits,
    nor the block layout (scripts/micro/wpattern.hip: the pattern alone sustains 4.9 TB/s) changes the launch time --
    the kernel sits on the power envelope and 3.5 TB/s of HBM writes take their share of it."""
    spread = os.environ.get("MLP_TRAIN_SPREAD_STORES", "1") in ("1", which)
    side = {c: [] for c in range(nchunks)}
    for pi, pn in enumerate(prog.panels):
        post = prog.panels[pi - 1]["post"] if pi > 0 else []
        pre = prog.panels[pi + 1]["pre"] if pi + 1 < len(prog.panels) else []
        n = pn["n"]
        first = pn["first"]
        if not spread:
            for wi, stmt in enumerate(post + pre):
                side[first + min(2 + wi, n - 1)].append(stmt)
            continue
        compute = [st for st in post if count_stores(st) == 0]
        stores = [st for st in post if count_stores(st) > 0]
        pos = 2
        for st in compute:
            at = min(pos, n - 1)
            # a statement that writes an activation register goes in front of this panel's first reader of it (one-panel
            # producers feed slot 0 of their consumer); the statements that follow only read that register
            m = _WRITES_REG.search(st)
            if m:
                readers = [c for c in range(first, first + n) if _slot_b(prog.slots[c]) == m.group(1)]
                if readers:
                    at = min(at, readers[0] - first - 1)
            side[first + at].append(st)
            pos += 1
        lo = min(pos + 1, n - 1)                 # >= 2 slots after the last transposing MFMA
        hi = max(lo, n - 2)
        last = lo
        for i, st in enumerate(stores):
            at = lo if len(stores) == 1 else lo + (hi - lo) * i // (len(stores) - 1)
            side[first + at].append(st)
            last = at
        for wi, st in enumerate(pre):
            side[first + min(max(last, pos) + wi, n - 1)].append(st)
    return side


def count_stores(stmt: str) -> int:
    """VMEM store instructions a side statement issues unconditionally (one global_store_dwordx4 each)."""
    return stmt.count("store_tfrag<") + stmt.count("reinterpret_cast<u32x4*>(mask_wave")


def emit_tile_body(e, prog, side, nchunks_total, prologue_lines, final_lines, lda,
                   counted=os.environ.get("MLP_TRAIN_COUNTED_VMCNT", "0") == "1"):
    """MFMA slots with A prefetch PREFETCH chunks ahead and ring-group boundaries where the load cursor
    enters a new group.  nchunks_total >= len(slots): trailing (padding) groups are still cycled through so
    that the ring phase is tile-invariant.  Group boundaries g >= 1 wait with vmcnt(K), K = stores this wave
    issued since the DMA of group g (see GROUP_BEGIN_K); the tile's first boundary drains everything (it also
    needs the tile-private DMAs issued just before it)."""
    nslots = len(prog.slots)
    e("        GROUP_BEGIN(0, 1);")
    since = 0

    def group_begin(g):
        nonlocal since
        if counted:
            assert since <= 60
            e(f"        GROUP_BEGIN_K({g}, {(g + 1) % SLOTS}, {since});")
        else:
            e(f"        GROUP_BEGIN({g}, {(g + 1) % SLOTS});")
        since = 0
    for c in range(min(PREFETCH, nslots)):
        e(f"        {lda(c)}")
    for ln in prologue_lines:
        e(f"        {ln}")
        since += count_stores(ln)
    e("        PIN();")
    cur_op = None
    for c, sl in enumerate(prog.slots):
        if sl.get("opname") != cur_op:
            cur_op = sl.get("opname")
            e(f"        // ---- {cur_op}")
        if sl["zero"]:
            e(f"        TMFMA0({sl['acc']}, A{c % PREFETCH}, {sl['bexpr']});")
        else:
            e(f"        MFMA({sl['acc']}, A{c % PREFETCH}, {sl['bexpr']});")
        lc = c + PREFETCH
        if lc < nslots:
            if lc % GROUP == 0:
                group_begin(lc // GROUP)
            e(f"        {lda(lc)}")
        for stmt in side[c]:
            e(f"        {stmt}")
            since += count_stores(stmt)
        e("        PIN();")
    # groups the load cursor never entered (stream padding): keep the ring protocol going
    first_unentered = (nslots - 1) // GROUP + 1
    for g in range(first_unentered, nchunks_total // GROUP):
        group_begin(g)
    for ln in final_lines:
        e(f"        {ln}")


def check_hazards(prog, side, prologue=(), preloaded=False):
    """Register-set discipline: op i reads the activation registers written by the previous op that writes any.
    Epilogue statements carry an `/*opN*/` tag; replay the tile in program order and assert that every B operand X[k] / Y[k]
    read by a slot of op i was last written by that op, and every register read by a transposing MFMA / tile_mask was
    written by the op that issues the statement."""
    import re
    last = {f"X[{k}]": -1 for k in range(16)} if preloaded else {}     # trunk plans (pre-GEMM form): X arrives preloaded ("op -1")
    wr = re.compile(r"epilogue_half<[^>]*>\(\w+, (?:[\w\[\]]+, )?([XY]\[\d+\])\);\s*/\*op(\d+)\*/")
    rd = re.compile(r"(?:TMFMA0|MFMA)\(\w+, ([XY]\[\d+\]), P[12]\);\s*/\*op(\d+)\*/")
    opidx = {}
    for sl in prog.slots:
        opidx.setdefault(sl["opname"], len(opidx))

    def run(stmts):
        for st in stmts:
            m = wr.search(st)
            if m:
                last[m.group(1)] = int(m.group(2))
            m = rd.search(st)
            if m:
                assert last.get(m.group(1)) == int(m.group(2)), ("transpose reads a stale register", st, last.get(m.group(1)))
    run(prologue)
    for c, sl in enumerate(prog.slots):
        b = sl["bexpr"]
        if b[0] in "XY":
            # the producer is the latest earlier op that writes activation registers at all (an op in between that only
            # produces a head value -- the density row when there is no bottleneck -- is transparent)
            cur = opidx[sl["opname"]]
            writers = [o for o in set(last.values()) if o < cur]
            assert writers and last.get(b) == max(writers), ("slot reads a register not produced by the previous writing op",
                                                             c, sl["opname"], b, last.get(b))
        run(side[c])
    return True


# =================================================================================================================
#  forward-with-save
# =================================================================================================================
def build_fwd_prog(tp: TrainPlan):
    plan = tp.fwd
    a = plan.arch
    prog = Prog()
    for oi, op in enumerate(plan.ops):
        hb, ml = tp.fwd_out[oi]
        relu = "true" if op.relu else "false"
        for (t0, t1) in plan.panels(op):
            pair = len(prog.panels) & 1
            first = len(prog.slots)
            spk = 1 if t1 is None else 2
            for ks in range(op.nk):
                seg, ksl = plan.seg_of(op, ks)
                if seg.regset in ("X", "Y"):
                    b = ("reg", f"{seg.regset}[{seg.reg0 + ksl}]")
                elif seg.regset == "enc":
                    b = ("lds", ksl * 1024)
                else:
                    b = ("lds", (0 if plan.pre_gemm else a.xyz_dim // 16) * 1024 + ksl * 1024)
                for w in range(spk):
                    prog.slots.append(dict(acc=f"acc{pair}{w}", b=b, panel=len(prog.panels), ks=ks,
                                           first_of_ks=(w == 0), zero=False, opname=op.name))
            post, post_t, post_late = [], [], []
            for which, t in ((0, t0), (1, t1)):
                if t is None:
                    continue
                acc = f"acc{pair}{which}"
                if op.out in ("X", "Y"):
                    if op.name == "head" and t == len(op.tiles) - 1:
                        post.append(f"raw_density = {acc}[0];")
                        continue
                    post.append(f"epilogue_half<{relu}, 0>({acc}, {op.out}[{2 * t}]);  /*op{oi}*/")
                    post.append(f"epilogue_half<{relu}, 8>({acc}, {op.out}[{2 * t + 1}]);  /*op{oi}*/")
                    if hb is not None and oi not in SKIP_STORES:
                        if not ABLATE_TMFMA:
                            post_t.append(f"TMFMA0({acc}, {op.out}[{2 * t}], P1);  /*op{oi}*/")
                            post_t.append(f"MFMA({acc}, {op.out}[{2 * t + 1}], P2);  /*op{oi}*/")
                        post_late.append(f"store_tfrag<0>({acc}, ht_wave + {(hb + t) * 2048}, lane16);")
                        post_late.append(f"store_tfrag<8>({acc}, ht_wave + {(hb + t) * 2048}, lane16);")
                    if ml is not None:
                        if t % 2 == 0:
                            post_late.append(f"mq{t // 2} = tile_mask({op.out}[{2 * t}], {op.out}[{2 * t + 1}]);")
                        else:
                            post_late.append(f"mq{t // 2} |= tile_mask({op.out}[{2 * t}], {op.out}[{2 * t + 1}]) << 8;")
                        last_tile = len(op.tiles) - 1
                        if t == last_tile:
                            nq = (last_tile + 2) // 2
                            vals = ", ".join(f"mq{q}" if q < nq else "0u" for q in range(4))
                            post_late.append(f"*reinterpret_cast<u32x4*>(mask_wave + {ml * 1024} + lane16) = u32x4{{{vals}}};")
                else:
                    post.append(f"raw_r = {acc}[0]; raw_g = {acc}[1]; raw_b = {acc}[2];")
            if op.pre:      # skip layer of a trunk plan: the accumulators start from k_pre_gemm's partial sums (bias included), 4 KiB per tile
                pre = [f"pre_load(acc{pair}0, pre_lane + {t0 * 4096});"]
                if t1 is not None:
                    pre.append(f"pre_load(acc{pair}1, pre_lane + {t1 * 4096});")
            else:
                pre = [f"BIAS(acc{pair}0, {op.first_tile + t0});"]
                if t1 is not None:
                    pre.append(f"BIAS(acc{pair}1, {op.first_tile + t1});")
            prog.panels.append(dict(first=first, n=len(prog.slots) - first, pair=pair, spk=spk,
                                    post=post + post_t + post_late, pre=pre))
    return prog


def assign_lds_b(prog, nchunks):
    """E-register rotation for LDS-resident B operands (as gen_mlp_bf16.py)."""
    side_extra = {c: [] for c in range(nchunks)}
    prologue = []
    ecount = 0
    cur_e = None
    for c, sl in enumerate(prog.slots):
        kind, val = sl["b"]
        if kind == "reg":
            sl["bexpr"] = val
            continue
        if sl["first_of_ks"]:
            cur_e = f"E{ecount % NE}"
            ecount += 1
            spk = prog.panels[sl["panel"]]["spk"]
            at = c - 2 * spk
            stmt = f"{cur_e} = LDB({val});"
            if at < 0:
                prologue.append(stmt)
            else:
                side_extra[at].append(stmt)
        sl["bexpr"] = cur_e
    return side_extra, prologue


def file_header(e, ns, consts):
    e("// AUTO-GENERATED by gen_mlp_train.py from mlp_train_plan.py -- do not edit by hand.")
    e("#include <hip/hip_runtime.h>")
    e('#include "kernels.hpp"')
    e('#include "raymath.hpp"')
    e(f"namespace mip {{ namespace {ns} {{")
    e("typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;")
    e("typedef __attribute__((ext_vector_type(16))) float f32x16;")
    for k, v in consts.items():
        e(f"constexpr int {k} = {v};")
    e(KERNEL_PREAMBLE.replace("BARRIER_INSN", "s_barrier").replace("WAIT_INSN", "s_waitcnt vmcnt(0) lgkmcnt(0)"))
    e(TRAIN_PREAMBLE)
    e("constexpr bool DMA = true;")


SELECTORS = [
    "bf16x8 P1, P2;   // selection matrices of the transposing MFMAs: P1[k][n] = (n == k), P2[k][n] = (n == 16 + k)",
    "#pragma unroll",
    "for (int j = 0; j < 8; ++j) {",
    "    P1[j] = (__bf16)((n == hi * 8 + j) ? 1.0f : 0.0f);",
    "    P2[j] = (__bf16)((n == 16 + hi * 8 + j) ? 1.0f : 0.0f);",
    "}",
]


def gen_trainfwd(tp: TrainPlan, variant: int = 0) -> str:
    if tp.pre_gemm:
        return gen_trainfwd_pre(tp, variant)
    sfx = f"_v{variant}" if variant else ""
    plan = tp.fwd
    a = plan.arch
    nchunks = len(plan.chunks)
    nreal = plan.n_real_chunks        # the stream is padded with zero chunks to whole ring groups (mlp_plan.RING_MULTIPLE)
    assert nchunks % GROUP == 0 and (nchunks // GROUP) % SLOTS == 0
    nenc = a.xyz_dim // 16
    enc_wave_bytes = 8192
    assert (nenc + 2) * 1024 <= enc_wave_bytes
    nbias_bytes = plan.n_tiles * 128
    ring_bytes = SLOTS * GROUP * CHUNK_BYTES
    enc_off = (ring_bytes + nbias_bytes + 1023) // 1024 * 1024
    lds_bytes = enc_off + WAVES * enc_wave_bytes
    assert lds_bytes <= 160 * 1024
    prog = build_fwd_prog(tp)
    # padding: inside the last ring group, or exactly one whole group of zeros (two view layers: 39 groups + 1), begun by an extra GROUP_BEGIN at the tile end
    assert len(prog.slots) == nreal and (nchunks - nreal < GROUP or (nchunks - nreal == GROUP and nreal % GROUP == 0))
    side_e, prologue_e = assign_lds_b(prog, nreal)
    side = place_sides(prog, nreal, "fwd")
    for c in range(nreal):
        side[c] = side_e[c] + side[c]
    check_hazards(prog, side)
    lines = []
    e = lines.append
    file_header(e, "trainfwd" + sfx, dict(kRingBytes=ring_bytes, kBiasBytes=nbias_bytes, kEncOff=enc_off,
                                    kEncWaveBytes=enc_wave_bytes, kLdsBytes=lds_bytes,
                                    kGroupBytes=GROUP * CHUNK_BYTES, kNumGroups=nchunks // GROUP,
                                    kTileSamples=WAVES * 32, kNH=tp.NH, kNMask=tp.NMASK))
    e("template <bool IPE>")
    e(f"__global__ void __launch_bounds__({WAVES * 64})")
    e("k_mlp_bf16_trainfwd(const char* __restrict__ stream, const float* __restrict__ bias_tab,")
    e("                    const __bf16* __restrict__ enc, const __bf16* __restrict__ viewenc, float4* __restrict__ rgb_sigma,")
    e("                    float4* __restrict__ raw_out, char* __restrict__ HT, char* __restrict__ masks, int64_t M,")
    e("                    int num_samples, int ntiles, float density_bias, float rgb_padding, RayIn rin,")
    e("                    const float* __restrict__ dnoise, float dnoise_scale) {")
    e("    extern __shared__ __attribute__((aligned(16))) char smem[];")
    e("    const int tid = threadIdx.x;")
    e("    const int lane = tid & 63;")
    e("    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);")
    e("    const int hi = lane >> 5, n = lane & 31;")
    e("    const unsigned lane16 = (unsigned)lane * 16u;")
    e("    const char* ring_lane = smem + lane16;")
    e("    const char* bias_lane = smem + kRingBytes + hi * 64;")
    e("    char* encw = smem + kEncOff + wave * kEncWaveBytes;")
    e("    const char* enc_lane = encw + lane16;")
    for ln in SELECTORS:
        e("    " + ln)
    e("    for (int i = tid; i < kBiasBytes / 16; i += blockDim.x)")
    e("        reinterpret_cast<float4*>(smem + kRingBytes)[i] = reinterpret_cast<const float4*>(bias_tab)[i];")
    e("    __syncthreads();")
    if SETPRIO:
        e("    if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // as the inference kernel (gen_mlp_bf16.SETPRIO)")
    e("    if ((int)blockIdx.x < ntiles) issue_group<DMA>(stream, smem, 0, 0, wave, lane16);")
    e("    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {")
    e("        const bool has_next = tile + (int)gridDim.x < ntiles;")
    e("        const int64_t wt = (int64_t)tile * 8 + wave;                 // wave tile (uniform)")
    e("        const int64_t s = wt * 32 + n;")
    e("        const int64_t sc = s < M ? s : M - 1;")
    e("        const int64_t ray = sc / num_samples;")
    e("        char* ht_wave = HT + wt * (int64_t)(kNH * 2048);")
    e("        char* mask_wave = masks + wt * (int64_t)(kNMask * 1024);")
    e("        if (IPE) {      // encoding computed here (registers -> LDS), see ipe_to_lds in gen_mlp_bf16.py")
    e(f"            issue_encodings<DMA, {nenc}, {nenc}>(nullptr, viewenc + ray * 32 + hi * 8, encw, lane16);")
    e(f"            ipe_to_lds<{nenc}>(rin, sc, num_samples, hi, encw + lane16);")
    e("        } else {")
    e(f"            issue_encodings<DMA, {nenc}, 0>(enc + sc * {a.xyz_dim} + hi * 8, viewenc + ray * 32 + hi * 8, encw, lane16);")
    e("        }")
    e("        bf16x8 X[16], Y[16], " + ", ".join(f"A{i}" for i in range(PREFETCH)) + ", E0, E1, E2;")
    e("        f32x16 acc00, acc01, acc10, acc11;")
    e("        unsigned mq0 = 0, mq1 = 0, mq2 = 0, mq3 = 0;")
    e("        float raw_density = 0.0f, raw_r = 0.0f, raw_g = 0.0f, raw_b = 0.0f;")

    def lda(c):
        slot = (c // GROUP) % SLOTS
        off = slot * GROUP * CHUNK_BYTES + (c % GROUP) * CHUNK_BYTES
        return f"A{c % PREFETCH} = LDA({off});"
    # tile prologue: transposed encodings (inputs of layer 0 / the skip layer / the view layer for wgrad)
    pro = []
    e0 = tp.h_blocks["enc"][0]
    accs = ["acc10", "acc11"]
    k = 0
    for b in range(nenc // 2):
        acc = accs[k % 2]
        k += 1
        pro.append(f"E0 = LDB({2 * b * 1024}); E1 = LDB({(2 * b + 1) * 1024});")
        pro.append(f"TMFMA0({acc}, E0, P1); MFMA({acc}, E1, P2);")
        pro.append(f"store_tfrag<0>({acc}, ht_wave + {(e0 + b) * 2048}, lane16); store_tfrag<8>({acc}, ht_wave + {(e0 + b) * 2048}, lane16);")
    if "view" in tp.h_blocks:          # use_viewdirs=False has no view layer, hence no view-feature T-block
        acc = accs[k % 2]
        vb = tp.h_blocks["view"][0]
        pro.append(f"E0 = LDB({nenc * 1024}); E1 = LDB({(nenc + 1) * 1024});")
        pro.append(f"TMFMA0({acc}, E0, P1); MFMA({acc}, E1, P2);")
        pro.append(f"store_tfrag<0>({acc}, ht_wave + {vb * 2048}, lane16); store_tfrag<8>({acc}, ht_wave + {vb * 2048}, lane16);")
    pro += prologue_e
    pro += prog.panels[0]["pre"]
    final = list(prog.panels[-1]["post"])
    final += [
        "if (hi == 0 && s < M) {",
        "    const float noisy_density = dnoise ? raw_density + dnoise_scale * dnoise[s] : raw_density;   // mip_nerf.py:232-233",
        "    rgb_sigma[s] = make_float4(rgb_activation(raw_r, rgb_padding), rgb_activation(raw_g, rgb_padding),",
        "                               rgb_activation(raw_b, rgb_padding), density_activation(noisy_density, density_bias));",
        "    raw_out[s] = make_float4(raw_r, raw_g, raw_b, raw_density);",
        "}",
    ]
    emit_tile_body(e, prog, side, nchunks, pro, final, lda)
    e("    }")
    e("}")
    e("}  // namespace trainfwd" + sfx)
    e("")
    e(f"int mlp_trainfwd_lds_bytes{sfx}() {{ return trainfwd{sfx}::kLdsBytes; }}")
    e(f"hipError_t launch_mlp_bf16_trainfwd{sfx}(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,")
    e("                                    float* rgb_sigma, float* raw_out, void* HT, void* masks, int64_t M, int num_samples,")
    e("                                    float density_bias, float rgb_padding, int grid_limit, const RayInputs* rays,")
    e("                                    const float* dnoise, float dnoise_scale, hipStream_t st) {")
    e(f"    using namespace trainfwd{sfx};")
    e("    const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);")
    e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
    e("    if (grid < 1) grid = 1;")
    e("    static int attr_done[64] = {};")
    e("    int dev = 0;")
    e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
    e("    if (!attr_done[dev]) {")
    e("        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_bf16_trainfwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        er = hipFuncSetAttribute((const void*)k_mlp_bf16_trainfwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        attr_done[dev] = 1;")
    e("    }")
    e("    RayIn rin = {nullptr, nullptr, nullptr, nullptr, 0, 0};")
    e("    if (rays) rin = RayIn{rays->t, rays->origins, rays->dirs, rays->radii, rays->min_deg, rays->disable_integration};")
    e("#define MIP_LAUNCH(I) hipLaunchKernelGGL((k_mlp_bf16_trainfwd<I>), dim3(grid), dim3(%d), kLdsBytes, st, (const char*)stream_w, \\" % (WAVES * 64))
    e("        bias_tab, (const __bf16*)enc, (const __bf16*)viewenc, (float4*)rgb_sigma, (float4*)raw_out, (char*)HT, (char*)masks, M, \\")
    e("        num_samples, ntiles, density_bias, rgb_padding, rin, dnoise, dnoise_scale)")
    e("    if (rays) MIP_LAUNCH(true); else MIP_LAUNCH(false);")
    e("#undef MIP_LAUNCH")
    e("    return hipGetLastError();")
    e("}")
    e("}  // namespace mip")
    return "\n".join(lines) + "\n"


def gen_trainfwd_pre(tp: TrainPlan, variant: int) -> str:
    """TRUNK forward-with-save of the two-kernel bf16 form (TrainPlan.build(arch, pre_gemm=True); round 5): k_pre_gemm (gen_pre_gemm.py,
    unchanged) has computed layer 0 and the encoding half of the skip layer; this kernel starts from the preloaded register set
    X = bf16(relu(layer 0)) -- whose eight T-blocks and ReLU mask row it writes first (x1 > 0 <=> the pre-activation was positive) --,
    runs layers 1 .. D-1, head, view layer and colour like k_mlp_bf16_trainfwd (same panels, same epilogues, same T-block / mask
    stores), and initialises the skip layer's accumulators from k_pre_gemm's fp32 partial sums.  The encoding's own T-blocks do not
    exist: the weight-gradient kernel reads the encoding fragments (WJob.b_src = 1)."""
    sfx = f"_pre_v{variant}"
    plan = tp.fwd
    a = plan.arch
    assert plan.pre_gemm and a.net_width == 256
    nchunks = len(plan.chunks)
    nreal = plan.n_real_chunks
    assert nchunks % GROUP == 0 and (nchunks // GROUP) % SLOTS == 0
    enc_wave_bytes = 8192            # only the two view-encoding k-steps live there
    nbias_bytes = plan.n_tiles * 128
    ring_bytes = SLOTS * GROUP * CHUNK_BYTES
    enc_off = (ring_bytes + nbias_bytes + 1023) // 1024 * 1024
    lds_bytes = enc_off + WAVES * enc_wave_bytes
    assert lds_bytes <= 160 * 1024
    prog = build_fwd_prog(tp)
    assert len(prog.slots) == nreal and (nchunks - nreal < GROUP or (nchunks - nreal == GROUP and nreal % GROUP == 0))
    side_e, prologue_e = assign_lds_b(prog, nreal)
    side = place_sides(prog, nreal, "fwd")
    for c in range(nreal):
        side[c] = side_e[c] + side[c]
    check_hazards(prog, side, preloaded=True)
    lines = []
    e = lines.append
    file_header(e, "trainfwd" + sfx, dict(kRingBytes=ring_bytes, kBiasBytes=nbias_bytes, kEncOff=enc_off,
                                    kEncWaveBytes=enc_wave_bytes, kLdsBytes=lds_bytes,
                                    kGroupBytes=GROUP * CHUNK_BYTES, kNumGroups=nchunks // GROUP,
                                    kTileSamples=WAVES * 32, kNH=tp.NH, kNMask=tp.NMASK))
    e(f"__global__ void __launch_bounds__({WAVES * 64})")
    e("k_mlp_bf16_trainfwd_pre(const char* __restrict__ stream, const float* __restrict__ bias_tab, const char* __restrict__ pre_x,")
    e("                        const char* __restrict__ pre_acc, const __bf16* __restrict__ viewenc, float4* __restrict__ rgb_sigma,")
    e("                        float4* __restrict__ raw_out, char* __restrict__ HT, char* __restrict__ masks, int64_t M,")
    e("                        int num_samples, int ntiles, float density_bias, float rgb_padding,")
    e("                        const float* __restrict__ dnoise, float dnoise_scale) {")
    e("    extern __shared__ __attribute__((aligned(16))) char smem[];")
    e("    const int tid = threadIdx.x;")
    e("    const int lane = tid & 63;")
    e("    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);")
    e("    const int hi = lane >> 5, n = lane & 31;")
    e("    const unsigned lane16 = (unsigned)lane * 16u;")
    e("    const char* ring_lane = smem + lane16;")
    e("    const char* bias_lane = smem + kRingBytes + hi * 64;")
    e("    char* encw = smem + kEncOff + wave * kEncWaveBytes;")
    e("    const char* enc_lane = encw + lane16;")
    for ln in SELECTORS:
        e("    " + ln)
    e("    for (int i = tid; i < kBiasBytes / 16; i += blockDim.x)")
    e("        reinterpret_cast<float4*>(smem + kRingBytes)[i] = reinterpret_cast<const float4*>(bias_tab)[i];")
    e("    __syncthreads();")
    if SETPRIO:
        e("    if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // as the inference kernel (gen_mlp_bf16.SETPRIO)")
    e("    if ((int)blockIdx.x < ntiles) issue_group<DMA>(stream, smem, 0, 0, wave, lane16);")
    e("    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {")
    e("        const bool has_next = tile + (int)gridDim.x < ntiles;")
    e("        const int64_t wt = (int64_t)tile * 8 + wave;                 // wave tile (uniform)")
    e("        const int64_t s = wt * 32 + n;")
    e("        const int64_t sc = s < M ? s : M - 1;")
    e("        const int64_t ray = sc / num_samples;")
    e("        char* ht_wave = HT + wt * (int64_t)(kNH * 2048);")
    e("        char* mask_wave = masks + wt * (int64_t)(kNMask * 1024);")
    e("        issue_encodings<DMA, 0, 0>(nullptr, viewenc + ray * 32 + hi * 8, encw, lane16);")
    e("        bf16x8 X[16], Y[16], " + ", ".join(f"A{i}" for i in range(PREFETCH)) + ", E0, E1, E2;")
    e("        // what k_pre_gemm left for this wave tile: X = bf16(relu(layer 0)) as 16 lane-linear fragments, the skip layer's accumulator images")
    e("        const char* prex_lane = pre_x + wt * 16384 + lane16;")
    e("        const char* pre_lane = pre_acc + wt * 32768 + lane16;")
    for k in range(16):
        e(f"        X[{k}] = PRE_LD(reinterpret_cast<const bf16x8*>(prex_lane + {k * 1024}));")
    e("        f32x16 acc00, acc01, acc10, acc11;")
    e("        unsigned mq0 = 0, mq1 = 0, mq2 = 0, mq3 = 0;")
    e("        float raw_density = 0.0f, raw_r = 0.0f, raw_g = 0.0f, raw_b = 0.0f;")

    def lda(c):
        slot = (c // GROUP) % SLOTS
        off = slot * GROUP * CHUNK_BYTES + (c % GROUP) * CHUNK_BYTES
        return f"A{c % PREFETCH} = LDA({off});"
    # tile prologue: T-blocks and ReLU mask of x1 (from the preloaded registers), T-block of the view features
    pro = []
    accs = ["acc10", "acc11"]
    x1 = tp.h_blocks["x1"][0]
    nW = a.net_width // 32
    for t in range(nW):
        acc = accs[t % 2]
        pro.append(f"TMFMA0({acc}, X[{2 * t}], P1); MFMA({acc}, X[{2 * t + 1}], P2);")
        pro.append(f"store_tfrag<0>({acc}, ht_wave + {(x1 + t) * 2048}, lane16); store_tfrag<8>({acc}, ht_wave + {(x1 + t) * 2048}, lane16);")
        if t % 2 == 0:
            pro.append(f"mq{t // 2} = tile_mask(X[{2 * t}], X[{2 * t + 1}]);")
        else:
            pro.append(f"mq{t // 2} |= tile_mask(X[{2 * t}], X[{2 * t + 1}]) << 8;")
    vals = ", ".join(f"mq{q}" if q < (nW + 1) // 2 else "0u" for q in range(4))
    pro.append(f"*reinterpret_cast<u32x4*>(mask_wave + 0 + lane16) = u32x4{{{vals}}};      // mask row 0 = layer 0")
    acc = accs[nW % 2]
    vb = tp.h_blocks["view"][0]
    pro.append("E0 = LDB(0); E1 = LDB(1024);")
    pro.append(f"TMFMA0({acc}, E0, P1); MFMA({acc}, E1, P2);")
    pro.append(f"store_tfrag<0>({acc}, ht_wave + {vb * 2048}, lane16); store_tfrag<8>({acc}, ht_wave + {vb * 2048}, lane16);")
    pro += prologue_e
    pro += prog.panels[0]["pre"]
    final = list(prog.panels[-1]["post"])
    final += [
        "if (hi == 0 && s < M) {",
        "    const float noisy_density = dnoise ? raw_density + dnoise_scale * dnoise[s] : raw_density;   // mip_nerf.py:232-233",
        "    rgb_sigma[s] = make_float4(rgb_activation(raw_r, rgb_padding), rgb_activation(raw_g, rgb_padding),",
        "                               rgb_activation(raw_b, rgb_padding), density_activation(noisy_density, density_bias));",
        "    raw_out[s] = make_float4(raw_r, raw_g, raw_b, raw_density);",
        "}",
    ]
    emit_tile_body(e, prog, side, nchunks, pro, final, lda)
    e("    }")
    e("}")
    e("}  // namespace trainfwd" + sfx)
    e("")
    e(f"hipError_t launch_mlp_bf16_trainfwd{sfx}(const void* stream_w, const float* bias_tab, const void* pre_x, const void* pre_acc, const void* viewenc,")
    e("                                    float* rgb_sigma, float* raw_out, void* HT, void* masks, int64_t M, int num_samples,")
    e("                                    float density_bias, float rgb_padding, int grid_limit, const float* dnoise, float dnoise_scale,")
    e("                                    hipStream_t st) {")
    e(f"    using namespace trainfwd{sfx};")
    e("    const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);")
    e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
    e("    if (grid < 1) grid = 1;")
    e("    static int attr_done[64] = {};")
    e("    int dev = 0;")
    e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
    e("    if (!attr_done[dev]) {")
    e("        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_bf16_trainfwd_pre, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        attr_done[dev] = 1;")
    e("    }")
    e(f"    hipLaunchKernelGGL(k_mlp_bf16_trainfwd_pre, dim3(grid), dim3({WAVES * 64}), kLdsBytes, st, (const char*)stream_w, bias_tab,")
    e("                       (const char*)pre_x, (const char*)pre_acc, (const __bf16*)viewenc, (float4*)rgb_sigma, (float4*)raw_out, (char*)HT,")
    e("                       (char*)masks, M, num_samples, ntiles, density_bias, rgb_padding, dnoise, dnoise_scale);")
    e("    return hipGetLastError();")
    e("}")
    e("}  // namespace mip")
    return "\n".join(lines) + "\n"



# =================================================================================================================
#  dgrad
# =================================================================================================================
def build_dgrad_prog(tp: TrainPlan):
    prog = Prog()
    mk_toggle = 0
    mk_of_op = {}
    for oi, op in enumerate(tp.bops):
        if op.mask is not None:
            mk_of_op[oi] = f"MK{mk_toggle}"
            mk_toggle ^= 1
    for oi, op in enumerate(tp.bops):
        first_panel_of_op = True
        for t0 in range(0, op.ntiles, 2):
            t1 = t0 + 1 if t0 + 1 < op.ntiles else None
            pair = len(prog.panels) & 1
            first = len(prog.slots)
            spk = 1 if t1 is None else 2
            for ks in range(op.nk):
                seg, ksl = tp.bseg_of(op, ks)
                bexpr = "R" if seg.regset == "raw" else f"{seg.regset}[{seg.reg0 + ksl}]"
                for w in range(spk):
                    prog.slots.append(dict(acc=f"acc{pair}{w}", bexpr=bexpr, panel=len(prog.panels), ks=ks,
                                           first_of_ks=(w == 0), zero=(ks == 0), opname=op.name))
            post, post_t, post_late = [], [], []
            for which, t in ((0, t0), (1, t1)):
                if t is None:
                    continue
                acc = f"acc{pair}{which}"
                if op.mask is not None:
                    mk = f"{mk_of_op[oi]}[{t // 2}]"
                    sh = 8 * (t & 1)
                    post.append(f"depilogue_half<true, 0, {sh}>({acc}, {mk}, {op.out}[{2 * t}]);  /*op{oi}*/")
                    post.append(f"depilogue_half<true, 8, {sh}>({acc}, {mk}, {op.out}[{2 * t + 1}]);  /*op{oi}*/")
                else:
                    post.append(f"depilogue_half<false, 0, 0>({acc}, 0u, {op.out}[{2 * t}]);  /*op{oi}*/")
                    post.append(f"depilogue_half<false, 8, 0>({acc}, 0u, {op.out}[{2 * t + 1}]);  /*op{oi}*/")
                if op.gblock is not None:      # (the bottleneck delta is consumed in registers only)
                    if not ABLATE_TMFMA:
                        post_t.append(f"TMFMA0({acc}, {op.out}[{2 * t}], P1);  /*op{oi}*/")
                        post_t.append(f"MFMA({acc}, {op.out}[{2 * t + 1}], P2);  /*op{oi}*/")
                    post_late.append(f"store_tfrag<0>({acc}, gt_wave + {(op.gblock + t) * 2048}, lane16);")
                    post_late.append(f"store_tfrag<8>({acc}, gt_wave + {(op.gblock + t) * 2048}, lane16);")
            pre = []
            if first_panel_of_op and op.mask is not None:
                # the mask dwords of this op, read from the wave-private LDS copy while the previous op finishes
                pre.append(f"{mk_of_op[oi]} = LDM({op.mask * 1024});")
            first_panel_of_op = False
            prog.panels.append(dict(first=first, n=len(prog.slots) - first, pair=pair, spk=spk,
                                    post=post + post_t + post_late, pre=pre))
    return prog


def gen_dgrad(tp: TrainPlan, variant: int = 0) -> str:
    sfx = f"_v{variant}" if variant else ""
    nchunks = len(tp.bchunks)
    nreal = tp.n_bchunks_real
    assert nchunks % GROUP == 0 and (nchunks // GROUP) % SLOTS == 0
    ring_bytes = SLOTS * GROUP * CHUNK_BYTES
    priv_wave_bytes = (tp.NMASK + 1) * 1024          # mask rows + the d_raw row
    priv_off = ring_bytes
    lds_bytes = priv_off + WAVES * priv_wave_bytes
    assert lds_bytes <= 160 * 1024
    prog = build_dgrad_prog(tp)
    assert len(prog.slots) == nreal
    side = place_sides(prog, nreal, "dgrad")
    check_hazards(prog, side)
    lines = []
    e = lines.append
    file_header(e, "dgrad" + sfx, dict(kRingBytes=ring_bytes, kPrivOff=priv_off, kPrivWaveBytes=priv_wave_bytes,
                                 kLdsBytes=lds_bytes, kGroupBytes=GROUP * CHUNK_BYTES, kNumGroups=nchunks // GROUP,
                                 kTileSamples=WAVES * 32, kNG=tp.NG, kNMask=tp.NMASK))
    e(f"__global__ void __launch_bounds__({WAVES * 64})")
    e("k_mlp_bf16_dgrad(const char* __restrict__ stream, const float4* __restrict__ d_raw, const char* __restrict__ masks,")
    e("                 char* __restrict__ GT, int64_t M, int ntiles) {")
    e("    extern __shared__ __attribute__((aligned(16))) char smem[];")
    e("    const int tid = threadIdx.x;")
    e("    const int lane = tid & 63;")
    e("    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);")
    e("    const int hi = lane >> 5, n = lane & 31;")
    e("    const unsigned lane16 = (unsigned)lane * 16u;")
    e("    const char* ring_lane = smem + lane16;")
    e("    char* privw = smem + kPrivOff + wave * kPrivWaveBytes;          // wave-private (uniform base)")
    e("    const char* priv_lane = privw + lane16;")
    for ln in SELECTORS:
        e("    " + ln)
    if SETPRIO:
        e("    if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // as the inference kernel (gen_mlp_bf16.SETPRIO)")
    e("    if ((int)blockIdx.x < ntiles) issue_group<DMA>(stream, smem, 0, 0, wave, lane16);")
    e("    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {")
    e("        const bool has_next = tile + (int)gridDim.x < ntiles;")
    e("        const int64_t wt = (int64_t)tile * 8 + wave;")
    e("        const int64_t s = wt * 32 + n;")
    e("        const int64_t sc = s < M ? s : M - 1;")
    e("        char* gt_wave = GT + wt * (int64_t)(kNG * 2048);")
    e("        const char* mask_lane = masks + wt * (int64_t)(kNMask * 1024) + lane16;")
    e("        // stage this wave's ReLU masks (kNMask rows) and upstream gradients into its private LDS area")
    e("#pragma unroll")
    e("        for (int l = 0; l < kNMask; ++l) dma_1k(mask_lane + l * 1024, privw + l * 1024);")
    e("        dma_1k(d_raw + sc, privw + kNMask * 1024);")
    e("        bf16x8 X[16], Y[16], R, " + ", ".join(f"A{i}" for i in range(PREFETCH)) + ";")
    e("        f32x16 acc00, acc01, acc10, acc11;")
    e("        u32x4 MK0, MK1;")

    def lda(c):
        slot = (c // GROUP) % SLOTS
        off = slot * GROUP * CHUNK_BYTES + (c % GROUP) * CHUNK_BYTES
        return f"A{c % PREFETCH} = LDA({off});"
    rb = tp.g_blocks["raw"][0]
    pro = [
        "{",
        "    const float4 dr = *reinterpret_cast<const float4*>(priv_lane + kNMask * 1024);",
        "    const bool live = hi == 0 && s < M;          // B operand: k = hi*8 + j, only k < 4 carries data",
        "    R[0] = (__bf16)(live ? dr.x : 0.0f); R[1] = (__bf16)(live ? dr.y : 0.0f);",
        "    R[2] = (__bf16)(live ? dr.z : 0.0f); R[3] = (__bf16)(live ? dr.w : 0.0f);",
        "    R[4] = R[5] = R[6] = R[7] = (__bf16)0.0f;",
        "}",
        "TMFMA0(acc10, R, P1);",
    ]
    pro += prog.panels[0]["pre"]
    pro += [f"store_tfrag<0>(acc10, gt_wave + {rb * 2048}, lane16); store_tfrag<8>(acc10, gt_wave + {rb * 2048}, lane16);"]
    final = list(prog.panels[-1]["post"])
    emit_tile_body(e, prog, side, nchunks, pro, final, lda)
    e("    }")
    e("}")
    e("}  // namespace dgrad" + sfx)
    e("")
    e(f"int mlp_dgrad_lds_bytes{sfx}() {{ return dgrad{sfx}::kLdsBytes; }}")
    e(f"hipError_t launch_mlp_bf16_dgrad{sfx}(const void* stream_wT, const float* d_raw, const void* masks, void* GT, int64_t M,")
    e("                                 int grid_limit, hipStream_t st) {")
    e(f"    using namespace dgrad{sfx};")
    e("    const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);")
    e("    int grid = ntiles < grid_limit ? ntiles : grid_limit;")
    e("    if (grid < 1) grid = 1;")
    e("    static int attr_done[64] = {};")
    e("    int dev = 0;")
    e("    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;")
    e("    if (!attr_done[dev]) {")
    e("        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_bf16_dgrad, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);")
    e("        if (er != hipSuccess) return er;")
    e("        attr_done[dev] = 1;")
    e("    }")
    e(f"    hipLaunchKernelGGL(k_mlp_bf16_dgrad, dim3(grid), dim3({WAVES * 64}), kLdsBytes, st, (const char*)stream_wT,")
    e("                       (const float4*)d_raw, (const char*)masks, (char*)GT, M, ntiles);")
    e("    return hipGetLastError();")
    e("}")
    e("}  // namespace mip")
    return "\n".join(lines) + "\n"


def train_variants():
    """(variant index, TrainPlan, forward source, dgrad source) of every architecture of gen_mlp_bf16.VARIANTS the training plan
    covers AND whose generated schedule passes the hazard check (variant 0 = shipped).  A shape that fails either (e.g. a
    view layer so narrow that the colour head would read activations its epilogue has not written yet) simply has no bf16
    training kernels: it trains in fp32 mode, and the library says so."""
    from gen_mlp_bf16 import VARIANTS
    from mipnerf_pl_amd import mlp_pre_plan
    out = []
    for vi, arch in enumerate(VARIANTS):
        try:
            if mlp_pre_plan.supported(arch):
                # two-kernel bf16 form (wide encoding): k_pre_gemm + a trunk forward-with-save, the standard dgrad, weight-gradient jobs
                # over the encoding fragments (round 5)
                tp = TrainPlan.build(arch, pre_gemm=True)
                out.append((vi, tp, gen_trainfwd(tp, vi), gen_dgrad(tp, vi)))
                continue
            if not arch.bf16_kernels:
                raise NotImplementedError("fp32-only architecture variant")
            tp = TrainPlan.build(arch)
            out.append((vi, tp, gen_trainfwd(tp, vi), gen_dgrad(tp, vi)))
        except (NotImplementedError, AssertionError) as ex:
            if vi == 0:
                raise
            print(f"variant {vi}: no bf16 training kernels ({type(ex).__name__}: {str(ex)[:120]})")
    return out


def gen_train_variants_header(trainable, n, pre=()):
    """Declarations + dispatch tables of the per-variant training launchers and table blobs (nullptr: no bf16 training kernels).
    `pre`: the variants whose forward-with-save is the trunk of the two-kernel form (their row of kLaunchTrainFwd stays null)."""
    L = ["// AUTO-GENERATED by gen_mlp_train.py from gen_mlp_bf16.VARIANTS -- do not edit by hand.", "#pragma once", '#include "kernels.hpp"']
    for vi in trainable:
        sfx = f"_v{vi}" if vi else ""
        L.append(f'extern "C" const unsigned char mip_train_tables{sfx}[];')
    L += ["namespace mip {",
          "typedef hipError_t (*LaunchTrainFwdFn)(const void* stream_w, const float* bias_tab, const void* enc, const void* viewenc,",
          "                                       float* rgb_sigma, float* raw_out, void* HT, void* masks, int64_t M, int num_samples,",
          "                                       float density_bias, float rgb_padding, int grid_limit, const RayInputs* rays,",
          "                                       const float* dnoise, float dnoise_scale, hipStream_t st);",
          "typedef hipError_t (*LaunchTrainFwdPreFn)(const void* stream_w, const float* bias_tab, const void* pre_x, const void* pre_acc,",
          "                                          const void* viewenc, float* rgb_sigma, float* raw_out, void* HT, void* masks, int64_t M,",
          "                                          int num_samples, float density_bias, float rgb_padding, int grid_limit, const float* dnoise,",
          "                                          float dnoise_scale, hipStream_t st);",
          "typedef hipError_t (*LaunchDgradFn)(const void* stream_wT, const float* d_raw, const void* masks, void* GT, int64_t M,",
          "                                    int grid_limit, hipStream_t st);"]
    for vi in trainable:
        if vi == 0:
            continue
        if vi in pre:
            L.append(f"hipError_t launch_mlp_bf16_trainfwd_pre_v{vi}(const void*, const float*, const void*, const void*, const void*, float*, float*, void*, void*,")
            L.append("                                            int64_t, int, float, float, int, const float*, float, hipStream_t);")
        else:
            L.append(f"hipError_t launch_mlp_bf16_trainfwd_v{vi}(const void*, const float*, const void*, const void*, float*, float*, void*, void*,")
            L.append("                                        int64_t, int, float, float, int, const RayInputs*, const float*, float, hipStream_t);")
        L.append(f"hipError_t launch_mlp_bf16_dgrad_v{vi}(const void*, const float*, const void*, void*, int64_t, int, hipStream_t);")

    def tab(fmt0, fmtv, which):
        return ", ".join((fmt0 if vi == 0 else fmtv.format(vi)) if vi in which else "nullptr" for vi in range(n))
    std = [vi for vi in trainable if vi not in pre]
    L.append(f"static const LaunchTrainFwdFn kLaunchTrainFwd[{n}] = {{{tab('launch_mlp_bf16_trainfwd', 'launch_mlp_bf16_trainfwd_v{}', std)}}};")
    L.append(f"static const LaunchTrainFwdPreFn kLaunchTrainFwdPre[{n}] = {{{tab('nullptr', 'launch_mlp_bf16_trainfwd_pre_v{}', list(pre))}}};")
    L.append(f"static const LaunchDgradFn kLaunchDgrad[{n}] = {{{tab('launch_mlp_bf16_dgrad', 'launch_mlp_bf16_dgrad_v{}', trainable)}}};")
    L.append(f"static const unsigned char* const kTrainTableBlobs[{n}] = {{{tab('mip_train_tables', 'mip_train_tables_v{}', trainable)}}};")
    L.append("}  // namespace mip")
    return "\n".join(L) + "\n"


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else HERE
    from gen_mlp_bf16 import VARIANTS
    tvs = train_variants()
    with open(os.path.join(outdir, "mlp_train_variants_gen.hpp"), "w") as f:
        f.write(gen_train_variants_header([v[0] for v in tvs], len(VARIANTS), pre=[v[0] for v in tvs if v[1].pre_gemm]))
    for vi, tp, src_fwd, src_dgrad in tvs:
        sfx = f"_v{vi}" if vi else ""
        with open(os.path.join(outdir, f"mlp_bf16_trainfwd_pre_gen{sfx}.hip" if tp.pre_gemm else f"mlp_bf16_trainfwd_gen{sfx}.hip"), "w") as f:
            f.write(src_fwd)
        with open(os.path.join(outdir, f"mlp_bf16_dgrad_gen{sfx}.hip"), "w") as f:
            f.write(src_dgrad)
        with open(os.path.join(outdir, f"_gen_train_tables{sfx}.bin"), "wb") as f:
            f.write(tp.blob())
        print(f"generated training kernels (variant {vi}): fwd {tp.fwd.n_real_chunks} chunks, dgrad {tp.n_bchunks_real} "
              f"(+{len(tp.bchunks) - tp.n_bchunks_real} pad) chunks, {len(tp.jobs)} wgrad jobs -> {outdir}")


if __name__ == "__main__":
    main()
