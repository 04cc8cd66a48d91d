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
// the ring group and the natural blocks issued two groups ago have landed (this wave's pieces: vmcnt; everybody's: the barrier), and every
// wave is done reading the slot the next group goes to
#define GROUP_BEGIN() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ float relu1(float x) { return __builtin_fmaxf(x, 0.0f); }

// One 1-KiB LDS-DMA piece of the weight stream: wave-uniform 64-bit base in SGPRs + 32-bit lane offset (saddr form), lands lane-linear
// at M0.  Inline asm on purpose (see gen_mlp_bf16.py: hipcc models the builtin as a flat access that degrades every later lgkmcnt wait).
__device__ __forceinline__ void dma_piece(const char* gbase, unsigned lds_addr, unsigned lane16) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(gbase), "s"(lds_addr)
        : "memory");
}
// One piece of a natural block: every lane brings 16 bytes of ITS sample's row (per-lane 64-bit address).
__device__ __forceinline__ void dma_piece_v(const float* src, unsigned lds_addr) {
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
        assert plan.n_groups % RING_SLOTS == 0
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
        return f"relu1({reg})" if op.in_relu else reg

    def nat_piece_src(self, u, q, nxt):
        op = self.p.ops[u["op"]]
        blk = op.blocks[u["block"]]
        base = ("encp" if blk.src == "enc" else "viewp") + ("_nxt" if nxt else "_cur")
        return f"{base} + {32 * blk.index + 4 * q}"

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
        cur_g = -1
        have_a = False             # a0 / a1 hold this step's A fragments (read by the previous step)
        have_bq = None             # (op, block, quad) held in `bq`
        b_ready = False            # `b` holds this step's B operand (computed by the previous step)
        thin_decl = set()
        nsteps = len(steps)
        for i, st in enumerate(steps):
            op, oi, blk, j, nq = st["op"], st["oi"], st["blk"], st["j"], st["nq"]
            nxt = steps[i + 1] if i + 1 < nsteps else None
            src_txt = f"reg {blk.src}[{blk.index}]" if blk.kind == DLAYOUT else f"{blk.src} block {blk.index}"
            E(f"        // {op.name} k-step {st['ks']} ({src_txt}, {j})")
            if nq and st["cpos"] in gstart:
                g = gstart[st["cpos"]]
                assert g == cur_g + 1, (g, cur_g)
                while pieces:                                # (should be empty: a group has at least as many k-steps as pieces)
                    E("        " + pieces.pop(0))
                E("        GROUP_BEGIN();")
                self.group_pieces(g, groups, pieces, fetch_at)
                cur_g = g
                have_a = False
            if nq:
                c0, k = groups[cur_g]
                assert c0 <= st["cpos"] and st["cpos"] + nq <= c0 + k, "a k-step straddles a ring group"
            for s_ in bias_head.get(i, []):
                E("        " + s_)
            slot = cur_g & 1
            if nq:
                coff = (st["cpos"] - groups[cur_g][0]) * 1024 + slot * GROUP_BYTES
                if not have_a:
                    E(f"        a0 = LDA({coff});" + (f" a1 = LDA({coff + 1024});" if nq == 2 else ""))
            # ---- B operand of THIS step
            if blk.kind == NATURAL:
                u = self.use_of(oi, st["ks"] // 16)
                key = (oi, st["ks"] // 16, j // 4)
                if have_bq != key:
                    E(f"        bq = LDB({u['slot'] * 4096 + (j // 4) * 1024});")
                    have_bq = key
                E(f"        b = bq[{j % 4}];")
            elif not b_ready:
                E(f"        b = {self.b_expr(st)};")
            # ---- reads for the NEXT step: A fragments, B operand
            nxt_a = nxt is not None and nq and nxt["nq"] and nxt["cpos"] not in gstart
            if nxt_a:
                noff = (nxt["cpos"] - groups[cur_g][0]) * 1024 + slot * GROUP_BYTES
                E(f"        n0 = LDA({noff});" + (f" n1 = LDA({noff + 1024});" if nxt["nq"] == 2 else ""))
            nxt_b = nxt is not None and nxt["blk"].kind == DLAYOUT and nxt["op"] is op      # (another op's registers are still accumulating)
            if nxt_b:
                E(f"        bn = {self.b_expr(nxt)};")
            nxt_q = None
            if nxt is not None and nxt["blk"].kind == NATURAL:
                key = (nxt["oi"], nxt["ks"] // 16, nxt["j"] // 4)
                if have_bq != key:
                    un = self.use_of(nxt["oi"], nxt["ks"] // 16)
                    # (legal one step ahead even on a group boundary: natural blocks are issued two groups before their first k-step)
                    E(f"        bqn = LDB({un['slot'] * 4096 + (nxt['j'] // 4) * 1024});")
                    nxt_q = key
            # ---- thin head riding on this step
            for th in op.thin:
                for row in range(th.nrows):
                    nm = f"thin{oi}_{row}"
                    if st["ks"] % 4 == 0:
                        E(f"        hw{row} = AUX4({(self.lay[('thin_w', oi, row)] + st['ks']) * 4});")
                    E(f"        {nm} = fmaf(b, hw{row}[{st['ks'] % 4}], {nm});")
                    thin_decl.add(nm)
            # ---- the MFMAs, one DMA piece between them
            nt = len(op.tiles)
            npc = 0
            if pieces:
                if nq:
                    c0g, kg = groups[cur_g]
                    left = sum(1 for s2 in steps[i:] if s2["nq"] and c0g <= s2["cpos"] < c0g + kg)       # k-steps left in this ring group
                    npc = -(-len(pieces) // left)
                else:
                    npc = 1
            at = {((k + 1) * nt) // (npc + 1): 0 for k in range(npc)} if nt else {}
            for k in range(npc if nt else 0):
                at[((k + 1) * nt) // (npc + 1)] += 1
            for t in range(nt):
                for _ in range(at.get(t, 0)):
                    E("        " + pieces.pop(0))
                E(f"        MFMA({op.out}[{t}], a{t // 4}[{t % 4}], b);")
            if nt == 0:
                for _ in range(npc):
                    E("        " + pieces.pop(0))
            for s_ in bias_in.get(i, []):
                E("        " + s_)
            if nt:
                E("        PIN();")
            if nxt_a:
                E("        a0 = n0;" + (" a1 = n1;" if nxt["nq"] == 2 else ""))
            have_a = bool(nxt_a)
            if nxt_b:
                E("        b = bn;")
            b_ready = bool(nxt_b)
            if nxt_q is not None:
                E("        bq = bqn;")
                have_bq = nxt_q
        assert cur_g == NG - 1, (cur_g, NG)
        while pieces:
            E("        " + pieces.pop(0))
        return sorted(thin_decl)

    def group_pieces(self, g, groups, pieces, fetch_at):
        """queue the LDS-DMA pieces issued during ring group g: the weight-stream group after it (cyclically: the next tile's first),
        then the natural blocks scheduled here"""
        gn = (g + 1) % len(groups)
        c0, k = groups[gn]
        npw = k // WAVES                                      # pieces per wave
        for i in range(npw):
            pieces.append(f"dma_piece(stream + {c0 * 1024} + wave * {npw * 1024} + {i * 1024}, ring_base + {(gn & 1) * GROUP_BYTES} + wave * {npw * 1024} + {i * 1024}, lane16);")
        for u in fetch_at.get(g, []):
            if u["prev_tile"] and not self._next_ptrs_emitted:
                pieces.append("NEXT_TILE_POINTERS();")
                self._next_ptrs_emitted = True
            for q in range(4):
                pieces.append(f"dma_piece_v({self.nat_piece_src(u, q, u['prev_tile'])}, nat_lds + {u['slot'] * 4096 + q * 1024});")

    # ---- the whole translation unit ----------------------------------------------------------------------
    def source(self):
        p, a = self.p, self.p.arch
        ns = f"f32r_v{self.vi}"
        self._next_ptrs_emitted = False
        self.L = []
        thin = self.body()
        body = "\n".join(self.L)
        first_fetch = [u for u in self.sched if u["fetch"] and u["prev_tile"]]
        prologue = []
        for u in first_fetch:
            for q in range(4):
                prologue.append(f"    dma_piece_v({self.nat_piece_src(u, q, False)}, nat_lds + {u['slot'] * 4096 + q * 1024});")
        g0c, g0k = p.groups()[0]
        for i in range(g0k // WAVES):
            prologue.append(f"    dma_piece(stream + wave * {g0k // WAVES * 1024} + {i * 1024}, ring_base + wave * {g0k // WAVES * 1024} + {i * 1024}, lane16);")
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
    const char* stream = stream_w;
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
    // prologue: ring group 0 and the natural blocks the first groups read
{chr(10).join(prologue)}
    f32x16 X[8], Y[8];
    f32x4 a0, a1, n0, n1, bq, bqn, hw0, hw1, hw2;
    float b, bn;
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
