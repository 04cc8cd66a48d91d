#!/usr/bin/env python3
"""Generator of the register-resident fp32 MLP kernels (mlp_f32r_gen_v<i>.hip) -- see mipnerf_pl_amd/mlp_f32r_plan.py for the dataflow.

    python gen_mlp_f32r.py [outdir]

For every architecture of gen_mlp_bf16.VARIANTS that fits (widths <= 256): one straight-line kernel body per tile of 128 samples (4 waves x
32 samples), one translation unit, one binary table blob (_gen_f32r_tables_v<i>.bin: weight-stream pack table + aux table, linked by
build.py), and a row in mlp_f32r_variants_gen.hpp.  Reference: MLP.forward, models/mip_nerf.py:75-111, activations 232-238.

Schedule of one k-step (8 or 4 MFMAs on different accumulators, 512 / 256 cycles of the matrix pipe at one wave per SIMD), pinned with
sched_barrier(0):
    A fragments of the NEXT k-step            two / one ds_read_b128 from the LDS ring
    B operand of the NEXT k-step              register of the previous op's D tile, ReLU applied on the way (one v_max), or the next
                                              16 bytes of a wave-private natural block every 4th k-step
    the MFMAs of THIS k-step, one LDS-DMA piece between them (the ring group after this one: 8 pieces per wave per group, one per
    k-step -- eight in a row cost ~480 issue cycles, measured 4.4 % of the kernel; natural blocks: 4 pieces each)
    VALU side work: a thin head's fma, the accumulator-init image of the tile that has just been read for the last time
Measured go / no-go of this structure (scripts/micro/f32r_probe.hip, profiles/r04b_f32r_probe.txt): 0.963 of the fp32 MFMA peak with the
ring, 0.943 with the layer structure.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from mipnerf_pl_amd.mlp_plan import DLAYOUT, NATURAL  # noqa: E402
from mipnerf_pl_amd.mlp_f32r_plan import GROUP_CHUNKS, NSLOT, RING_SLOTS, F32RPlan, supported  # noqa: E402

GROUP_BYTES = GROUP_CHUNKS * 1024
RING_BYTES = RING_SLOTS * GROUP_BYTES
WAVES = 4
PIECE_WINDOW = float(os.environ.get("MLP_F32R_PIECE_WINDOW", "0.5"))
GEN_ABLATE = int(os.environ.get("MLP_F32R_GEN_ABLATE", "0"))      # timing experiments (wrong results): 1 no LDS-DMA, 2 no accumulator images, 4 no thin heads, 8 no ReLU   # fraction of a ring group's k-steps that carry its LDS-DMA pieces
TILE_SAMPLES = 32 * WAVES


PREAMBLE = r"""
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDA(off) (*reinterpret_cast<const f32x4*>(ring_lane + (off)))
#define LDB(off) (*reinterpret_cast<const f32x4*>(nat_lane + (off)))
#define AUX4(off) (*reinterpret_cast<const f32x4*>(aux_lane + (off)))
#define BIAS(acc, off) acc = *reinterpret_cast<const f32x16*>(aux_lane + (off))
#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), acc, 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
// Ring-group barrier with a COUNTED wait: vmcnt retires in issue order, and this wave issued exactly K LDS-DMA pieces after the ones the
// next group needs (the pieces queued at the previous barrier: the ring is RING_SLOTS deep, group g + RING_SLOTS - 1 is in flight while
// g is read), so vmcnt(K) = "my share of the next group and of the natural blocks queued two barriers ago has landed"; lgkmcnt(0) = my
// reads of the slot that is about to be refilled have returned; the barrier makes both true for every wave.
#if defined(MLP_F32R_ABLATE) && MLP_F32R_ABLATE == 1          // timing experiments only (races): no barrier / no waits at all
#define GROUP_BEGIN(K) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K) : "memory")
#elif defined(MLP_F32R_ABLATE) && MLP_F32R_ABLATE == 2
#define GROUP_BEGIN(K) asm volatile("" ::: "memory")
#else
#define GROUP_BEGIN(K) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K) : "memory")
#endif

__device__ __forceinline__ float relu1(float x) { return __builtin_fmaxf(x, 0.0f); }

// One 1-KiB LDS-DMA piece of the weight stream: wave-uniform 64-bit base in SGPRs + 32-bit lane offset (saddr form) + an immediate that the
// hardware adds to BOTH the global address and the LDS destination (M0 + imm + lane * 16), so the four pieces of a batch share one base
// and one M0 value.  Inline asm on purpose (see gen_mlp_bf16.py: hipcc models the builtin as a flat access that degrades every later
// lgkmcnt wait).  The bases are made opaque per batch (OPAQUE_S): otherwise the compiler hoists ~600 loop-invariant 64-bit piece addresses
// out of the tile loop, spills them to VGPR lanes and pays two v_readlane per piece (measured: the pieces cost 4 % of the kernel that way).
#define OPAQUE_S(x) asm volatile("" : "+s"(x))
template <int IMM>
__device__ __forceinline__ void dma_piece(const char* gbase, unsigned lds_addr, unsigned lane16) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:%4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(gbase), "s"(lds_addr), "n"(IMM)
        : "memory");
}
// One piece of a natural block: every lane brings 16 bytes of ITS sample's row: per-lane 64-bit row address + immediate; M0 is the LDS
// target minus that immediate.  The rows are read once per tile (MLP_F32R_NAT_NT=1: non-temporal, so that hundreds of MB of encodings do
// not compete with the 2.33-MiB weight stream for the XCD's L2).
#if defined(MLP_F32R_NAT_NT) && MLP_F32R_NAT_NT
#define NAT_POLICY " nt"
#else
#define NAT_POLICY ""
#endif
template <int IMM>
__device__ __forceinline__ void dma_piece_v(const float* row, unsigned m0_val) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off offset:%3" NAT_POLICY "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(row), "s"(m0_val), "n"(IMM)
        : "memory");
}
"""


class Gen:
    def __init__(self, plan: F32RPlan, vi: int):
        self.p = plan
        self.vi = vi
        self.lay, self.H = plan.aux_layout()
        self.aux_bytes = (2 * self.H * 4 + 255) // 256 * 256
        self.aux_off = RING_BYTES
        self.nat_off = self.aux_off + self.aux_bytes
        self.lds_bytes = self.nat_off + WAVES * NSLOT * 4096
        assert self.lds_bytes <= 160 * 1024, self.lds_bytes
        assert plan.n_groups % RING_SLOTS == 0 and plan.n_groups >= 2 * RING_SLOTS
        self.sched = plan.natural_schedule()
        self.L = []

    def emit(self, s=""):
        self.L.append(s)

    # ---- helpers -----------------------------------------------------------------------------------
    def steps(self):
        """every k-step of the tile in program order: dict(oi, op, ks, blk, j, cpos (first chunk), nq)"""
        out = []
        for oi, op in enumerate(self.p.ops):
            for ks in range(op.nk):
                out.append(dict(oi=oi, op=op, ks=ks, blk=op.blocks[ks // 16], j=ks % 16, cpos=op.chunk0 + ks * op.quads, nq=op.quads))
        return out

    def use_of(self, oi, bi):
        for u in self.sched:
            if u["op"] == oi and u["block"] == bi:
                return u
        raise KeyError((oi, bi))

    def b_expr(self, st):
        """C++ expression of the B operand of step st (for DLAYOUT: the register, ReLU on the way); natural blocks are handled by quads"""
        op, blk, j = st["op"], st["blk"], st["j"]
        assert blk.kind == DLAYOUT
        reg = f"{blk.src}[{blk.index}][{j}]"
        return f"relu1({reg})" if op.in_relu and not (GEN_ABLATE & 8) else reg

    def nat_pieces(self, u, nxt, indent="        "):
        """the four LDS-DMA statements of one natural block (the first carries the opaque copy of the wave's LDS base)"""
        op = self.p.ops[u["op"]]
        blk = op.blocks[u["block"]]
        row = ("encp" if blk.src == "enc" else "viewp") + ("_nxt" if nxt else "_cur")
        out = []
        for q in range(4):
            imm = (32 * blk.index + 4 * q) * 4
            assert 0 <= imm < 4096
            pre = "nl = nat_lds; OPAQUE_S(nl); " if q == 0 else ""
            out.append(f"{pre}dma_piece_v<{imm}>({row}, nl + ({u['slot'] * 4096 + q * 1024 - imm}));")
        return out

    def ring_pieces(self, gn, groups):
        """the LDS-DMA statements of ring group gn: batches of four pieces share a base (immediate offsets 0 .. 3072)"""
        c0, k = groups[gn]
        npw = k // WAVES
        out = []
        for i in range(npw):
            pre = ""
            if i % 4 == 0:
                pre = (f"gp = sw{npw}; OPAQUE_S(gp); gp += {c0 * 1024 + i * 1024}; lp = lw{npw}; OPAQUE_S(lp); lp += {(gn % RING_SLOTS) * GROUP_BYTES + i * 1024}; ")
            out.append(f"{pre}dma_piece<{(i % 4) * 1024}>(gp, lp, lane16);")
        return out

    # ---- the tile body ------------------------------------------------------------------------------
    def body(self):
        p = self.p
        steps = self.steps()
        groups = p.groups()
        gstart = {c0: gi for gi, (c0, k) in enumerate(groups)}
        NG = len(groups)
        fetch_at = {}
        for u in self.sched:
            if u["fetch"]:
                fetch_at.setdefault(u["issue_group"], []).append(u)
        # accumulator-init (bias image) placement: the tiles of an MFMA op's output set are re-initialised right after their last read
        mfma_ops = [oi for oi, op in enumerate(p.ops) if op.tiles]
        bias_in, bias_head, bias_tile_start = {}, {}, []
        step_index = {(st["oi"], st["ks"]): i for i, st in enumerate(steps)}
        for n, oi in enumerate(mfma_ops):
            op = p.ops[oi]
            for t in range(len(op.tiles)):
                stmt = f"BIAS({op.out}[{t}], {self.lay[('bias', oi, t)] * 4});"
                if n == 0:
                    bias_tile_start.append(stmt)
                    continue
                last = None
                for pj in range(oi - 1, -1, -1):
                    pop = p.ops[pj]
                    for bi, blk in enumerate(pop.blocks):
                        if blk.kind == DLAYOUT and blk.src == op.out and blk.index == t:
                            cand = step_index[(pj, bi * 16 + 15)]
                            last = cand if last is None else max(last, cand)
                    if pop.tiles and pop.out == op.out:
                        break                              # older reads belong to an older value of the set
                first_of_op = step_index[(oi, 0)]
                prev_first = step_index[(mfma_ops[n - 1], 0)]
                at = prev_first + 1 + 2 * t if last is None else max(last + 1, prev_first + 1)
                while at < first_of_op and at in bias_in:          # one accumulator image (4 ds_read_b128) per k-step
                    at += 1
                if at >= first_of_op:
                    bias_head.setdefault(first_of_op, []).append(stmt)
                else:
                    bias_in.setdefault(at, []).append(stmt)
        E = self.emit
        E("        // ---- generated tile body ----")
        for s_ in bias_tile_start:
            E("        " + s_)
        pieces = []                # LDS-DMA statements waiting for a k-step to ride on
        have_bq = self.first_bq_key(steps)      # the prologue / the previous tile's last step has read the first natural quad
        b_ready = False            # `b` holds this step's B operand (computed by the previous step)
        thin_decl = set()
        nsteps = len(steps)
        mfma_idx = [i for i, st in enumerate(steps) if st["nq"]]
        nxt_mfma = {}              # step index -> the next MFMA step (cyclic over tiles)
        for k, i in enumerate(mfma_idx):
            nxt_mfma[i] = steps[mfma_idx[(k + 1) % len(mfma_idx)]]
        group_len = {gi: sum(1 for i in mfma_idx if c0 <= steps[i]["cpos"] < c0 + k) for gi, (c0, k) in enumerate(groups)}
        begun = 0                  # ring group whose BEGIN was emitted last (group 0's: by the prologue / the previous tile)
        since_begin = 1            # MFMA steps since that BEGIN (it sits one k-step before its group)
        for i, st in enumerate(steps):
            op, oi, blk, j, nq = st["op"], st["oi"], st["blk"], st["j"], st["nq"]
            nxt = steps[(i + 1) % nsteps]
            src_txt = f"reg {blk.src}[{blk.index}]" if blk.kind == DLAYOUT else f"{blk.src} block {blk.index}"
            E(f"        // {op.name} k-step {st['ks']} ({src_txt}, {j})")
            nm = nxt_mfma.get(i)
            if nq:
                g_cur = p.group_of(st["cpos"])
                assert g_cur == begun, (g_cur, begun)
                c0, k = groups[g_cur]
                assert c0 <= st["cpos"] and st["cpos"] + nq <= c0 + k, "a k-step straddles a ring group"
                if nm["cpos"] in gstart:
                    # ---- the NEXT k-step opens ring group g: its barrier sits HERE, one k-step early -- this step's A fragments are in
                    # registers already, so the slot of group g - 1 is free for group g + 1 once every wave is past the barrier, and the
                    # first reads of group g (issued during this step's MFMAs) never wait behind a barrier
                    g = gstart[nm["cpos"]]
                    assert g == (begun + 1) % NG
                    while pieces:                                # (normally empty: pieces are paced to finish in half a group)
                        E("        " + pieces.pop(0))
                    E(f"        GROUP_BEGIN({self.wait_count(g, groups, fetch_at)});")
                    self.group_pieces(g, groups, pieces, fetch_at)
                    begun, since_begin = g, 0
            for s_ in bias_head.get(i, []):
                E("        " + s_)
            # ---- B operand of THIS step
            if blk.kind == NATURAL:
                u = self.use_of(oi, st["ks"] // 16)
                key = (oi, st["ks"] // 16, j // 4)
                assert have_bq == key, ("natural quad not prefetched", key, have_bq)
                E(f"        b{i % 2} = bq[{j % 4}];")
            elif not b_ready:
                E(f"        b{i % 2} = {self.b_expr(st)};")
            # ---- reads for the NEXT step: A fragments (of the next MFMA step), B operand
            if nq:
                g_n = p.group_of(nm["cpos"])
                noff = (nm["cpos"] - groups[g_n][0]) * 1024 + (g_n % RING_SLOTS) * GROUP_BYTES
                E(f"        n0 = LDA({noff});" + (f" n1 = LDA({noff + 1024});" if nm["nq"] == 2 else ""))
            nxt_b = nxt["blk"].kind == DLAYOUT and nxt["op"] is op and i + 1 < nsteps      # (another op's registers are still accumulating)
            if nxt_b:
                E(f"        b{(i + 1) % 2} = {self.b_expr(nxt)};")
            nxt_q = None
            if nxt["blk"].kind == NATURAL:
                key = (nxt["oi"], nxt["ks"] // 16, nxt["j"] // 4)
                if have_bq != key:
                    un = self.use_of(nxt["oi"], nxt["ks"] // 16)
                    # (legal one step ahead even across a group boundary or the tile boundary: natural blocks are issued two groups before
                    # their first k-step, and one GROUP_BEGIN in between has waited for them)
                    E(f"        bqn = LDB({un['slot'] * 4096 + (nxt['j'] // 4) * 1024});")
                    nxt_q = key
            # ---- thin head riding on this step
            for th in (op.thin if not (GEN_ABLATE & 4) or not op.tiles else []):
                for row in range(th.nrows):
                    nm_ = f"thin{oi}_{row}"
                    if st["ks"] % 4 == 0:
                        E(f"        hw{row} = AUX4({(self.lay[('thin_w', oi, row)] + st['ks']) * 4});")
                    E(f"        {nm_} = fmaf(b{i % 2}, hw{row}[{st['ks'] % 4}], {nm_});")
                    thin_decl.add(nm_)
            # ---- the MFMAs, LDS-DMA pieces between them: all pieces of a group are issued in its first half (the last one lands well
            # before the next barrier asks for it)
            nt = len(op.tiles)
            npc = 0
            if pieces:
                if nq:
                    window = max(1, int(group_len[begun] * PIECE_WINDOW) - since_begin)
                    npc = -(-len(pieces) // window)
                else:
                    npc = 1
            at = {}
            for k in range(npc if nt else 0):
                pos = ((k + 1) * nt) // (npc + 1)
                at[pos] = at.get(pos, 0) + 1
            for t in range(nt):
                for _ in range(at.get(t, 0)):
                    pc = pieces.pop(0)
                    if not (GEN_ABLATE & 1) or pc.startswith("NEXT"):
                        E("        " + pc)
                E(f"        MFMA({op.out}[{t}], a{t // 4}[{t % 4}], b{i % 2});")
            if nt == 0:
                for _ in range(npc):
                    pc = pieces.pop(0)
                    if not (GEN_ABLATE & 1) or pc.startswith("NEXT"):
                        E("        " + pc)
            for s_ in bias_in.get(i, []):
                if not (GEN_ABLATE & 2):
                    E("        " + s_)
            if nt:
                E("        PIN();")
            if nq:
                E("        a0 = n0;" + (" a1 = n1;" if nm["nq"] == 2 else ""))
                since_begin += 1
            b_ready = bool(nxt_b)
            if nxt_q is not None:
                E("        bq = bqn;")
                have_bq = nxt_q
        assert begun == 0, begun
        for oi2, op2 in enumerate(p.ops):
            for th in op2.thin:
                for row in range(th.nrows):
                    thin_decl.add(f"thin{oi2}_{row}")
        while pieces:
            E("        " + pieces.pop(0))
        assert have_bq == self.first_bq_key(steps)
        return sorted(thin_decl)

    @staticmethod
    def first_bq_key(steps):
        st = steps[0]
        return (st["oi"], st["ks"] // 16, st["j"] // 4) if st["blk"].kind == NATURAL else None

    def queue_ops(self, g, groups, fetch_at):
        """number of LDS-DMA instructions queued at the barrier of group g"""
        gn = (g + RING_SLOTS - 1) % len(groups)
        return groups[gn][1] // WAVES + 4 * len(fetch_at.get(g, []))

    def wait_count(self, g, groups, fetch_at):
        """vmcnt operand of group g's barrier: the pieces queued at the barriers of groups g - RING_SLOTS + 2 .. g - 1 may stay in flight"""
        NG = len(groups)
        k = sum(self.queue_ops((g - d) % NG, groups, fetch_at) for d in range(1, RING_SLOTS - 1))
        assert 0 <= k <= 63
        return k

    def _fetch_at(self):
        out = {}
        for u in self.sched:
            if u["fetch"]:
                out.setdefault(u["issue_group"], []).append(u)
        return out

    def group_pieces(self, g, groups, pieces, fetch_at, prologue_cur=False):
        """queue the LDS-DMA pieces issued during ring group g: the weight-stream group after it (cyclically: the next tile's first),
        then the natural blocks scheduled here"""
        gn = (g + RING_SLOTS - 1) % len(groups)             # its slot, (g - 1) % RING_SLOTS, has just been read for the last time
        pieces += self.ring_pieces(gn, groups)
        for u in fetch_at.get(g, []):
            # group 0's barrier sits in the PREVIOUS tile's last k-step: what is issued behind it belongs to the next tile
            nxt = (u["prev_tile"] or g == 0) and not prologue_cur
            if nxt and not self._next_ptrs_emitted:
                pieces.append("NEXT_TILE_POINTERS();")
                self._next_ptrs_emitted = True
            pieces += self.nat_pieces(u, nxt)

    # ---- the whole translation unit ----------------------------------------------------------------------
    def source(self):
        p, a = self.p, self.p.arch
        ns = f"f32r_v{self.vi}"
        self._next_ptrs_emitted = False
        self.L = []
        thin = self.body()
        body = "\n".join(self.L)
        # prologue of the first tile = what the previous tile's tail does for every later one: the natural blocks issued "in the previous
        # tile" and ring group 0, group 0's barrier, what is queued behind it (ring group 1, natural blocks of issue group 0), and the
        # first k-step's operands
        groups = p.groups()
        prologue = []
        for u in [u for u in self.sched if u["fetch"] and u["prev_tile"]]:
            prologue += ["    " + x for x in self.nat_pieces(u, False)]
        for g0 in range(RING_SLOTS - 1):
            prologue += ["    " + x for x in self.ring_pieces(g0, groups)]
        prologue.append("    GROUP_BEGIN(0);")
        q0 = []
        keep = self._next_ptrs_emitted
        self._next_ptrs_emitted = True           # (no next-tile pointers in the prologue: the first tile's own)
        self.group_pieces(0, groups, q0, {g: [dict(u, prev_tile=False)] for g, us in self._fetch_at().items() for u in us if g == 0}, prologue_cur=True)
        self._next_ptrs_emitted = keep
        prologue += ["    " + x for x in q0]
        st0 = self.steps()[0]
        prologue.append(f"    a0 = LDA(0);" + (" a1 = LDA(1024);" if st0["nq"] == 2 else ""))      # group 0 sits in slot 0
        if st0["blk"].kind == NATURAL:
            prologue.append(f"    bq = LDB({self.use_of(0, 0)['slot'] * 4096});")
        head_ops = [(oi, op) for oi, op in enumerate(p.ops) if op.thin]
        dens = [(oi, op) for oi, op in head_ops if op.kind == 1]
        col = [(oi, op) for oi, op in head_ops if op.kind == 2]
        assert len(dens) == 1 and len(col) == 1
        doi, coi = dens[0][0], col[0][0]
        nrgb = col[0][1].thin[0].nrows
        has_view = any(b.src == "view" for op in p.ops for b in op.blocks)
        src = f"""// AUTO-GENERATED by gen_mlp_f32r.py from mlp_f32r_plan.py -- do not edit by hand.
// Register-resident fp32 MFMA MLP of Mip-NeRF (reference: models/mip_nerf.py:75-111 + activations 232-238), v_mfma_f32_32x32x2_f32.
// architecture variant {self.vi}: depth {a.net_depth} width {a.net_width} cond {a.net_depth_condition}x{a.net_width_condition} xyz {a.xyz_dim} use_viewdirs={int(a.use_viewdirs)}
// {p.n_real_chunks} chunks = {p.n_groups} ring groups per tile of {TILE_SAMPLES} samples; {sum(len(op.tiles) * op.nk for op in p.ops)} MFMAs per wave and tile
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include "raymath.hpp"
namespace mip {{
namespace {ns} {{
constexpr int kRingBytes = {RING_BYTES};
constexpr int kGroupBytes = {GROUP_BYTES};
constexpr int kAuxOff = {self.aux_off};
constexpr int kAuxHalfBytes = {self.H * 4};
constexpr int kAuxFloats = {2 * self.H};
constexpr int kNatOff = {self.nat_off};
constexpr int kLdsBytes = {self.lds_bytes};
constexpr int kTileSamples = {TILE_SAMPLES};
constexpr int kXyzDim = {a.xyz_dim};
{PREAMBLE}
__global__ void __launch_bounds__({WAVES * 64})
k_mlp_f32r(const char* __restrict__ stream_w, const float* __restrict__ aux, const float* __restrict__ enc, const float* __restrict__ viewenc,
           float4* __restrict__ rgb_sigma, float4* __restrict__ raw_out, int64_t M, int num_samples, int ntiles, float density_bias,
           float rgb_padding, const float* __restrict__ dnoise, float dnoise_scale) {{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, n = lane & 31;
    const unsigned lane16 = lane * 16;
    // aux table (accumulator-init images, thin-head weights): resident in LDS for the whole launch
    for (int i = threadIdx.x; i < kAuxFloats / 4; i += blockDim.x)
        reinterpret_cast<float4*>(smem + kAuxOff)[i] = reinterpret_cast<const float4*>(aux)[i];
    __syncthreads();
    const char* ring_lane = smem + lane16;
    const char* aux_lane = smem + kAuxOff + hi * kAuxHalfBytes;
    const char* nat_lane = smem + kNatOff + wave * {NSLOT * 4096} + lane16;
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem);
    const unsigned nat_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + kNatOff) + wave * {NSLOT * 4096};
{chr(10).join(f"    const char* sw{n} = stream_w + wave * {n * 1024}; const unsigned lw{n} = ring_base + wave * {n * 1024};" for n in sorted({k // WAVES for _, k in p.groups()}))}
    const char* gp = stream_w;
    unsigned lp = 0, nl = 0;
    // this lane's sample of a tile, its encoding row and its ray's view encoding (clamped past the end: loads only)
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int64_t s_cur = (int64_t)tile * kTileSamples + wave * 32 + n;
    const float* encp_cur = enc + (s_cur < M ? s_cur : M - 1) * kXyzDim + 16 * hi;
    const float* viewp_cur = viewenc + ((s_cur < M ? s_cur : M - 1) / num_samples) * 32 + 16 * hi;
    int64_t s_nxt = s_cur;
    const float* encp_nxt = encp_cur;
    const float* viewp_nxt = viewp_cur;
#define NEXT_TILE_POINTERS()                                                                     \\
    do {{                                                                                         \\
        const int tn_ = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;           \\
        s_nxt = (int64_t)tn_ * kTileSamples + wave * 32 + n;                                     \\
        const int64_t sc_ = s_nxt < M ? s_nxt : M - 1;                                           \\
        encp_nxt = enc + sc_ * kXyzDim + 16 * hi;                                                \\
        viewp_nxt = viewenc + (sc_ / num_samples) * 32 + 16 * hi;                                \\
    }} while (0)
    f32x16 X[8], Y[8];
    f32x4 a0, a1, n0, n1, bq, bqn, hw0, hw1, hw2;
    float b0, b1;
    // prologue: ring group 0 and the natural blocks the first groups read
{chr(10).join(prologue)}
    for (; tile < ntiles; tile += gridDim.x) {{
        float {", ".join(f"{t} = 0.0f" for t in thin)};
{body}
        // ---- thin heads: the two lane halves hold the two halves of every dot product
        {{
            float dn = thin{doi}_0 + __shfl_xor(thin{doi}_0, 32);
            dn += *reinterpret_cast<const float*>(aux_lane + {self.lay[('thin_b', doi)] * 4});
            float rr[3] = {{0.f, 0.f, 0.f}};
{chr(10).join(f"            rr[{c}] = thin{coi}_{c} + __shfl_xor(thin{coi}_{c}, 32) + *reinterpret_cast<const float*>(aux_lane + {(self.lay[('thin_b', coi)] + c) * 4});" for c in range(nrgb))}
            if (hi == 0 && s_cur < M) {{
                // mip_nerf.py:232-233: raw_density += density_noise * randn, before the activation
                const float nd = dnoise ? dn + dnoise_scale * dnoise[s_cur] : dn;
                rgb_sigma[s_cur] = make_float4(rgb_activation(rr[0], rgb_padding), rgb_activation(rr[1], rgb_padding),
                                               rgb_activation(rr[2], rgb_padding), density_activation(nd, density_bias));
                if (raw_out) raw_out[s_cur] = make_float4(rr[0], rr[1], rr[2], dn);
            }}
        }}
        s_cur = s_nxt;
        encp_cur = encp_nxt;
        viewp_cur = viewp_nxt;
    }}
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may land after the workgroup has released its LDS
#undef NEXT_TILE_POINTERS
}}
}}  // namespace {ns}

hipError_t launch_mlp_f32r_v{self.vi}(const void* stream_w, const float* aux, const float* enc, const float* viewenc, float* rgb_sigma, float* raw_out,
                               int64_t M, int num_samples, float density_bias, float rgb_padding, int grid_limit, const float* dnoise,
                               float dnoise_scale, hipStream_t st) {{
    using namespace {ns};
    static int attr_done[64] = {{}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {{
        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_f32r, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (er != hipSuccess) return er;
        attr_done[dev] = 1;
    }}
    const int64_t nt64 = (M + kTileSamples - 1) / kTileSamples;
    if (nt64 > 0x7fffffff) return hipErrorInvalidValue;
    const int ntiles = (int)nt64;
    int grid = ntiles < grid_limit ? ntiles : grid_limit;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_mlp_f32r, dim3(grid), dim3({WAVES * 64}), kLdsBytes, st, (const char*)stream_w, aux, enc, viewenc, (float4*)rgb_sigma,
                       (float4*)raw_out, M, num_samples, ntiles, density_bias, rgb_padding, dnoise, dnoise_scale);
    return hipGetLastError();
}}
}}  // namespace mip
"""
        return src


def variants_header(vis, n):
    L = ["// AUTO-GENERATED by gen_mlp_f32r.py -- do not edit by hand.", "#pragma once", '#include "kernels.hpp"', "namespace mip {",
         "typedef hipError_t (*LaunchF32RFn)(const void* stream_w, const float* aux, const float* enc, const float* viewenc, float* rgb_sigma,",
         "                                   float* raw_out, int64_t M, int num_samples, float density_bias, float rgb_padding, int grid_limit,",
         "                                   const float* dnoise, float dnoise_scale, hipStream_t st);"]
    for vi in vis:
        L.append(f"hipError_t launch_mlp_f32r_v{vi}(const void*, const float*, const float*, const float*, float*, float*, int64_t, int, float, float, int,")
        L.append("                               const float*, float, hipStream_t);")
        L.append(f'extern "C" const unsigned char mip_f32r_tables_v{vi}[];')
    names = ", ".join(f"launch_mlp_f32r_v{vi}" if vi in vis else "nullptr" for vi in range(n))
    blobs = ", ".join(f"mip_f32r_tables_v{vi}" if vi in vis else "nullptr" for vi in range(n))
    L.append(f"static const LaunchF32RFn kLaunchF32R[{n}] = {{{names}}};")
    L.append(f"static const unsigned char* const kF32RTableBlobs[{n}] = {{{blobs}}};")
    L.append("}  // namespace mip")
    return "\n".join(L) + "\n"


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else HERE
    sys.path.insert(0, HERE)
    from gen_mlp_bf16 import VARIANTS
    vis = []
    for vi, arch in enumerate(VARIANTS):
        if not supported(arch):
            print(f"f32r variant {vi}: not generated (wider than 256: the LDS-resident k_mlp_f32 serves it)")
            continue
        plan = F32RPlan.build(arch)
        g = Gen(plan, vi)
        with open(os.path.join(outdir, f"mlp_f32r_gen_v{vi}.hip"), "w") as f:
            f.write(g.source())
        with open(os.path.join(outdir, f"_gen_f32r_tables_v{vi}.bin"), "wb") as f:
            f.write(plan.blob())
        vis.append(vi)
        print(f"generated f32r variant {vi}: {plan.n_real_chunks} chunks, {plan.n_groups} groups, LDS {g.lds_bytes} B")
    with open(os.path.join(outdir, "mlp_f32r_variants_gen.hpp"), "w") as f:
        f.write(variants_header(vis, len(VARIANTS)))


if __name__ == "__main__":
    main()
