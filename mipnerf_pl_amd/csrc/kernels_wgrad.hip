// Weight-gradient kernel of the bf16 Mip-NeRF MLP (torch autograd of models/mip_nerf.py:75-111:
// dW_j = sum_samples delta_j a_j^T, db_j = sum_samples delta_j) and the split reduction.
//
// Operands are the "T-blocks" written by the forward-with-save and dgrad kernels (mlp_train_plan.py): per wave
// tile (32 samples) and 32-feature block, two lane-linear 1-KiB fragments in which lane (hi, n) holds feature
// column n for 8 samples -- directly an A (delta) or B (activation) operand of v_mfma_f32_32x32x16_bf16 with the
// contraction running over samples.  A job = (<= 8 delta blocks) x (<= 8 activation blocks) of one weight matrix;
// a workgroup (8 waves, wave w owns delta block w and the 8+1 accumulator tiles of its row) streams its share of
// the wave tiles through a 4-deep LDS ring filled with global_load_lds (wave w moves delta block w and
// activation block w when they exist: a wave-uniform number of 1-KiB DMAs per stage, so the ring is synchronised
// with one counted s_waitcnt + one s_barrier per stage), accumulates in fp32 registers and writes one partial per split.
// Bound: HBM (1 KiB of operands per 32 KFLOP... 128 FLOP/B at 8x8 blocks); see DESIGN.md.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mip {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
#ifndef MIP_WGRAD_STAGES
#define MIP_WGRAD_STAGES 4
#endif
constexpr int kStages = MIP_WGRAD_STAGES;      // LDS ring depth (build knob MLP_WGRAD_STAGES: 3 / 5 measured no better, profiles/r03aa_wgrad_splits.txt)
constexpr int kStageBytes = 16 * 2048;       // [8 activation blocks | 8 delta blocks] x 2 KiB
constexpr int kWgradLds = kStages * kStageBytes;

// Every saved-activation / delta block is read exactly once per job: MLP_WGRAD_NT=1 (build knob) marks the stream non-temporal
#if defined(MIP_WGRAD_NT) && MIP_WGRAD_NT
#define MIP_WGRAD_LOAD_POLICY " nt"
#else
#define MIP_WGRAD_LOAD_POLICY ""
#endif
// two 1-KiB DMAs: global (uniform base + lane*16, +1024) -> LDS (uniform dst, +1024)
__device__ __forceinline__ void dma_block(const char* gbase, char* lbase, unsigned lane16) {
    const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbase;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2" MIP_WGRAD_LOAD_POLICY "\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024" MIP_WGRAD_LOAD_POLICY "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(gbase), "s"(lds_addr)
        : "memory");
}
// Timing experiment ("natural-layout hand-off"): MLP_WGRAD_TR=1 reads every operand fragment with two transposing
// ds_read_b64_tr_b16 instead of one ds_read_b128 -- the LDS traffic a weight-gradient kernel would have if the producers stored their
// k-step registers as they are (lane = sample) and the transposition happened here.  Same bytes, same MFMAs; the RESULTS ARE WRONG
// (the T-blocks in memory are still transposed): never the default.
#if defined(MIP_WGRAD_TR) && MIP_WGRAD_TR
typedef short v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_frag(const char* p) {
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(size_t)(p));
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(size_t)(p + 8));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}
#else
__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
#endif

// ---- b_src = 1 jobs (pre-GEMM plans, round 5): the B operand is a 32-feature column block of the ROW-MAJOR bf16 encoding [M, xyz_dim]
// (what k_pre_gemm reads in the training forward): no transposed copy of the 672-wide encoding is ever written.
// DMA: a block = 32 samples x 64 bytes; lane L of the first of its two DMAs moves the 16-byte piece (L & 3) of sample (L >> 2), the second
// the samples 16 .. 31 -- the LDS image is row-major [32 samples][32 features].  Rows past M are clamped (their deltas are zero).
// Operand read: fragment f of the weight-gradient MFMA's B operand wants, in lane (hi, n), feature column n of the block for the 8 samples
// drow(hi, 8 f + j) -- the sample order of the delta T-blocks (mlp_train_plan.frag_sample).  ds_read_b64_tr_b16 transposes 4 x 4 within
// 16-lane groups: lane i receives element (i & 3) of the 8-byte granules addressed by lanes (i >> 2) + 4 k, k = 0 .. 3 (probed:
// profiles/r02n_ds_read_tr_probe.txt).  Lane i of group (hi, g = n >> 4) therefore ADDRESSES the granule (feature quad 4 g + (i & 3),
// sample 16 f + 8 r + 4 hi + (i >> 2)) for read r in {0, 1} and RECEIVES feature 16 g + i of samples 16 f + 8 r + 4 hi + 0 .. 3.
typedef short v4s16e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_frag_enc(const char* blk, unsigned enc_lane_off, int f) {
    const char* p = blk + enc_lane_off + f * (16 * 64);
    const v4s16e lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16e*)(size_t)(p));
    const v4s16e hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16e*)(size_t)(p + 8 * 64));
    typedef short v8s16e __attribute__((ext_vector_type(8)));
    const v8s16e v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

// Timing experiment ("store every other layer, recompute the skipped one here"): MLP_WGRAD_RECOMPUTE_PROBE=<bit mask of
// job ids>.  The workgroups of those jobs run k_wgrad_recompute_body: a stage carries the instruction mix of the role-flipped recompute
// job -- wave w owns activation block w of the SKIPPED layer: it forms relu(W[block w, :] a_prev) for the stage's 32 samples (16 k-steps:
// A = a_prev read from the stage's activation T-blocks with transposing ds_read_b64_tr_b16, B = its 16-KiB slice of W held in 64
// registers), packs the tile to bf16 -- which IS its B operand of the weight-gradient MFMAs -- and contracts it against all eight delta
// blocks (16 ds_read_b128 as A operands); the bias gradient of delta block w is 16 VALU adds.  Same HBM bytes per stage as today's job
// (delta + a_prev); + 16 MFMAs, + 32 transposing LDS reads, + one epilogue per wave and stage.  The operands are whatever the T-blocks
// hold: the RESULTS ARE WRONG by construction (build.py demands an opt-in).
#if defined(MIP_WGRAD_RECOMPUTE_PROBE) && MIP_WGRAD_RECOMPUTE_PROBE
#ifndef MIP_WGRAD_RECOMPUTE_SCHED
#define MIP_WGRAD_RECOMPUTE_SCHED 0
#endif
typedef short v4s16p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_frag_tr(const char* p) {
    const v4s16p lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16p*)(size_t)(p));
    const v4s16p hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16p*)(size_t)(p + 8));
    typedef short v8s16p __attribute__((ext_vector_type(8)));
    const v8s16p v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void k_wgrad_recompute_body(char* smem, const char* __restrict__ HT, const char* __restrict__ GT, const WgradJob* jp,
                                                       const int4 wg, int64_t n_wt, int NH, int NG, float* __restrict__ partials, int lane, int wave) {
    const unsigned lane16 = (unsigned)lane * 16u;
    const int64_t lo = n_wt * wg.y / wg.z, hi = n_wt * (wg.y + 1) / wg.z;
    const int nst = (int)(hi - lo);
    // the job's own activation blocks are the ones the forward no longer stores (MLP_TRAIN_SKIP_STORES): read the NEXT layer's blocks
    // (8 further on, stored, ordinary relu activations) in their place -- same bytes, operands that toggle like the real ones would
    const int a_blk = jp->a_blk[wave], b_blk = jp->b_blk[wave] + 8;
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float bias_sum = 0.0f;
    bf16x8 wslice[16];                                    // this wave's 32 x 256 slice of the skipped layer's weights (any bits will do)
#pragma unroll
    for (int k = 0; k < 16; ++k) wslice[k] = *reinterpret_cast<const bf16x8*>(HT + ((int64_t)(wave * 16 + k) * 64 + lane) * 16);
    auto issue = [&](int64_t wt, int stage) {
        char* st = smem + stage * kStageBytes;
        dma_block(HT + (wt * NH + b_blk) * 2048, st + wave * 2048, lane16);
        dma_block(GT + (wt * NG + a_blk) * 2048, st + 16384 + wave * 2048, lane16);
    };
    if (nst > 0) {
#pragma unroll
        for (int s = 0; s < kStages - 1; ++s) issue(lo + (s < nst ? s : nst - 1), s);
        for (int i = 0; i < nst; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kStages - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            const int nxt = i + kStages - 1;
            issue(lo + (nxt < nst ? nxt : nst - 1), nxt % kStages);
            const char* st = smem + (i % kStages) * kStageBytes + lane16;
#if MIP_WGRAD_RECOMPUTE_SCHED == 0
            // first form (profiles/r05b_*): one accumulator chain, the next operand one k-step ahead
            f32x16 r0;
#pragma unroll
            for (int r = 0; r < 16; ++r) r0[r] = 0.0f;
            bf16x8 t = lds_frag_tr(st);
#pragma unroll
            for (int k = 0; k < 16; ++k) {                // 16 k-steps over a_prev's 8 blocks x 2 fragments, next operand one step ahead
                bf16x8 tn = t;
                if (k + 1 < 16) tn = lds_frag_tr(st + (k + 1) * 1024);
                r0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t, wslice[k], r0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                t = tn;
            }
            bf16x8 x0, x1;                                // relu + bf16: the recomputed activation block, as B fragments
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0[r] = (__bf16)fmaxf(r0[r], 0.0f);
                x1[r] = (__bf16)fmaxf(r0[r + 8], 0.0f);
            }
#else
            // second form: two accumulator chains (even / odd k-steps), operands four k-steps ahead (a ring of four fragments)
            f32x16 r0, r1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { r0[r] = 0.0f; r1[r] = 0.0f; }
            bf16x8 tq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) tq[k] = lds_frag_tr(st + k * 1024);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bf16x8 t = tq[k & 3];
                if (k & 1) r1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t, wslice[k], r1, 0, 0, 0);
                else r0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t, wslice[k], r0, 0, 0, 0);
                if (k + 4 < 16) tq[k & 3] = lds_frag_tr(st + (k + 4) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 x0, x1;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                x0[r] = (__bf16)fmaxf(r0[r] + r1[r], 0.0f);
                x1[r] = (__bf16)fmaxf(r0[r + 8] + r1[r + 8], 0.0f);
            }
#endif
            bf16x8 d0 = lds_frag(st + 16384), d1 = lds_frag(st + 16384 + 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) {                 // all eight delta blocks against the wave's activation block
                bf16x8 e0 = d0, e1 = d1;
                if (j + 1 < 8) { e0 = lds_frag(st + 16384 + (j + 1) * 2048); e1 = lds_frag(st + 16384 + (j + 1) * 2048 + 1024); }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, x0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, x1, acc[j], 0, 0, 0);
                if (j == wave) {                          // the bias gradient of delta block w: 16 VALU adds per stage
#pragma unroll
                    for (int r = 0; r < 8; ++r) bias_sum += (float)d0[r] + (float)d1[r];
                }
                __builtin_amdgcn_sched_barrier(0);
                d0 = e0; d1 = e1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float* out = partials + (int64_t)wg.w * kWgradJobFloats + (int64_t)wave * 9 * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(out + j * 1024 + q * 4) = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
    out[8 * 1024] = bias_sum;
}
#endif
}  // namespace

// One workgroup's share of one job.  ENC = false: the B blocks are T-blocks of the saved-activation buffer (every job of the standard
// shapes; exactly the code of rounds 1-4).  ENC = true (round 5, pre-GEMM plans): 32-feature column blocks of the row-major encoding.
// Two instantiations behind one workgroup-uniform branch, so the encoding path costs the standard jobs nothing (as one body with the
// choice inside the unrolled operand loop, the standard training step ran 12 % slower: 4.73 instead of 4.20 ms).
// ENC = 2 (the one-call training step of the unbounded model): the encoding arrives as the B-operand FRAGMENTS k_pre_gemm reads fastest
// ([wave tile][k-step][lane (hi, n)][8 features]); a 32-feature block = two consecutive k-steps = 2 KiB, gathered into the same row-major LDS
// image as ENC = 1 (see the DMA below), so both encoding forms share the operand reads.
template <int ENC>
__device__ __forceinline__ void wgrad_body(char* smem, const char* __restrict__ HT, const char* __restrict__ GT, const WgradJob* jp, const int4 wg,
                                           int64_t n_wt, int NH, int NG, float* __restrict__ partials, const WgradEnc* __restrict__ Ep, int lane,
                                           int wave) {
    const unsigned lane16 = (unsigned)lane * 16u;
    const int nA = jp->nA, nB = jp->nB;
    const bool with_bias = jp->bias != 0;
    const int64_t lo = n_wt * wg.y / wg.z, hi = n_wt * (wg.y + 1) / wg.z;
    const int nst = (int)(hi - lo);
    const int a_blk = jp->a_blk[wave], b_blk = jp->b_blk[wave];
    const bool active = wave < nA;
    WgradEnc E = {nullptr, 1, 0, 0};
    if (ENC) E = *Ep;                                       // (uniform scalar loads) the record the training forward left behind the T-blocks
    // encoding jobs: DMA source offset of this lane (sample lane >> 2 of a 16-sample half, 16-byte piece lane & 3) and its operand-read offset
    const unsigned enc_dma_piece = (unsigned)(lane & 3) * 16u;
    const unsigned enc_lane_off = (unsigned)((4 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (4 * ((lane >> 4) & 1) + (lane & 3)) * 8);

    f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // Each wave moves only blocks that exist: its delta block (wave < nA) and one activation block (wave < nB),
    // 2 DMAs each.  The number of DMAs per stage is wave-uniform, so each wave waits with its own counted vmcnt.
    const bool has_a = wave < nA, has_b = wave < nB;
    const int ndma = (has_a ? 2 : 0) + (has_b ? 2 : 0);
    auto issue = [&](int64_t wt, int stage) {
        char* st = smem + stage * kStageBytes;
        if (has_b) {
            if (ENC == 2) {
                // fragment source: the block's two k-steps = 2 KiB, piece (k-step ks, lane half hi, sample n) at ks * 1024 + hi * 512 + n * 16.
                // Landed lane-linearly they would make the transposing reads 4-way bank-conflicted (k-step and lane half are multiples of 256
                // bytes apart); every lane therefore FETCHES the piece that belongs at its place of the row-major [32 samples][64 B] image
                // the ENC = 1 path reads conflict-free: lane L of DMA d brings feature octet L & 3 of sample 16 d + (L >> 2).
                const char* base = (const char*)E.enc + (wt * E.frag_ksteps + 2 * b_blk) * 1024;
                const char* p0 = base + ((lane >> 1) & 1) * 1024 + (lane & 1) * 512 + (lane >> 2) * 16;
                const char* p1 = p0 + 256;
                char* dst = st + wave * 2048;
                const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dst;
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off" MIP_WGRAD_LOAD_POLICY "\n\t"
                    "s_add_u32 m0, m0, 1024\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %2, off" MIP_WGRAD_LOAD_POLICY "\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(p0), "v"(p1), "s"(lds_addr)
                    : "memory");
            } else if (ENC == 1) {
                // rows wt * 32 + (lane >> 2) [+ 16], clamped to the last sample (their deltas are zero); the clamp makes the row per-lane,
                // so both DMAs take a full 64-bit lane address
                const int64_t s0 = wt * 32 + (lane >> 2), s1 = s0 + 16;
                const int64_t r0 = s0 < E.M ? s0 : E.M - 1, r1 = s1 < E.M ? s1 : E.M - 1;
                const char* base = (const char*)E.enc + (int64_t)b_blk * 64;
                const char* p0 = base + r0 * E.row_bytes + enc_dma_piece;
                const char* p1 = base + r1 * E.row_bytes + enc_dma_piece;
                char* dst = st + wave * 2048;
                const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dst;
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off" MIP_WGRAD_LOAD_POLICY "\n\t"
                    "s_add_u32 m0, m0, 1024\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %2, off" MIP_WGRAD_LOAD_POLICY "\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(p0), "v"(p1), "s"(lds_addr)
                    : "memory");
            } else {
                dma_block(HT + (wt * NH + b_blk) * 2048, st + wave * 2048, lane16);
            }
        }
        if (has_a) dma_block(GT + (wt * NG + a_blk) * 2048, st + 16384 + wave * 2048, lane16);
    };

    if (nst > 0) {
#pragma unroll
        for (int s = 0; s < kStages - 1; ++s) issue(lo + (s < nst ? s : nst - 1), s);
        for (int i = 0; i < nst; ++i) {
            // own DMAs of stage i have landed (stages i+1 .. i+kStages-2 may still be in flight) ...
            if (ndma == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kStages - 2)) : "memory");
            else if (ndma == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kStages - 2)) : "memory");
            // ... and everybody's: stage i is readable, and the slot of stage i-1 is free for refilling
            __builtin_amdgcn_s_barrier();
            const int nxt = i + kStages - 1;
            issue(lo + (nxt < nst ? nxt : nst - 1), nxt % kStages);
            if (active) {
                const char* sb = smem + (i % kStages) * kStageBytes;      // (stage base without the lane-linear offset)
                const char* st = sb + lane16;
                const bf16x8 a0 = lds_frag(st + 16384 + wave * 2048);
                const bf16x8 a1 = lds_frag(st + 16384 + wave * 2048 + 1024);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < nB) {
                        const bf16x8 b0 = ENC ? lds_frag_enc(sb + j * 2048, enc_lane_off, 0) : lds_frag(st + j * 2048);
                        const bf16x8 b1 = ENC ? lds_frag_enc(sb + j * 2048, enc_lane_off, 1) : lds_frag(st + j * 2048 + 1024);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
                    }
                }
                if (with_bias) {
                    acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ones, acc[8], 0, 0, 0);
                    acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ones, acc[8], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail DMAs before the LDS is released
    }
    if (active) {
        float* out = partials + (int64_t)wg.w * kWgradJobFloats + (int64_t)wave * 9 * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j < nB || (j == 8 && with_bias)) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(out + j * 1024 + q * 4) =
                        make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
            }
        }
    }
}

__global__ void __launch_bounds__(512)
k_mlp_wgrad(const char* __restrict__ HT, const char* __restrict__ GT, const WgradJob* __restrict__ jobs,
            const int4* __restrict__ wg_tab, int64_t n_wt, int NH, int NG, float* __restrict__ partials, const WgradEnc* __restrict__ Ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int4 wg = wg_tab[blockIdx.x];                     // job, split, nsplits, partial slot
    const WgradJob* jp = jobs + wg.x;                       // uniform: scalar loads
#if defined(MIP_WGRAD_RECOMPUTE_PROBE) && MIP_WGRAD_RECOMPUTE_PROBE
    if ((MIP_WGRAD_RECOMPUTE_PROBE >> wg.x) & 1) {          // timing experiment, workgroup-uniform
        k_wgrad_recompute_body(smem, HT, GT, jp, wg, n_wt, NH, NG, partials, lane, wave);
        return;
    }
#endif
    if (jp->b_src != 0) {                                   // workgroup-uniform
        if (Ep->row_bytes == 0) wgrad_body<2>(smem, HT, GT, jp, wg, n_wt, NH, NG, partials, Ep, lane, wave);
        else wgrad_body<1>(smem, HT, GT, jp, wg, n_wt, NH, NG, partials, Ep, lane, wave);
    } else {
        wgrad_body<0>(smem, HT, GT, jp, wg, n_wt, NH, NG, partials, Ep, lane, wave);
    }
}

// grad[idx] = sum over the job's splits of the partial at `pos`, for every position that feeds a parameter
// (idx < nparams: assign or accumulate) or the fp32 scratch region behind them (idx >= nparams: always assigned)
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float* __restrict__ partials, const int32_t* __restrict__ otab, const int2* __restrict__ job_slots,
               float* __restrict__ grad_flat, float* __restrict__ scratch, int nparams, int accumulate) {
    // 64 positions x 4 split groups per workgroup: group g sums splits g, g+4, ... with several loads in flight, the four group
    // sums meet in LDS in a fixed order (deterministic; one thread per position walked the ~26 splits one load at a time)
    __shared__ float red[4][64];
    const int job = blockIdx.y;
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int pos = blockIdx.x * 64 + o;
    const bool live = pos < kWgradJobFloats;
    const int32_t idx = live ? otab[(int64_t)job * kWgradJobFloats + pos] : -1;
    const int2 js = job_slots[job];                          // first partial slot, number of splits
    float s = 0.0f;
    if (idx >= 0) {
        const float* p = partials + (int64_t)js.x * kWgradJobFloats + pos;
#pragma unroll 8
        for (int k = g; k < js.y; k += 4) s += p[(int64_t)k * kWgradJobFloats];
    }
    red[g][o] = s;
    __syncthreads();
    if (g != 0 || idx < 0) return;
    s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    if (idx >= nparams) scratch[idx - nparams] = s;
    else grad_flat[idx] = accumulate ? grad_flat[idx] + s : s;     // every parameter has exactly one source position
}

// The bottleneck (extra_layer, mip_nerf.py:102) has no activation, so its T-blocks are never stored; with
// M = sum_s delta_view x8^T [Wc, W] and dbv = sum_s delta_view [Wc] (scratch, from the "M" job):
//   dW_view[:, :W] = M W_extra^T + dbv b_extra^T ;  db_view = dbv ;  dW_extra = W_view[:, :W]^T M ;  db_extra = W_view[:, :W]^T dbv
// fp32 on the master parameters, 25 MFLOP.
__global__ void __launch_bounds__(256)
k_wgrad_post(WgradPost P, const float* __restrict__ scratch, float* __restrict__ grad_flat, int accumulate) {
    const int W = P.W, Wc = P.Wc, ldv = P.ldv;
    const float* M = scratch;
    const float* dbv = scratch + Wc * W;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float val;
    float* dst;
    if (i < W * W) {                                   // dW_extra[o][c] = sum_r W_view[r][o] M[r][c]
        const int o = i / W, c = i - o * W;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;      // 4 independent chains: 16 loads in flight per lane
#pragma unroll 2
        for (int r = 0; r < Wc; r += 4) {
            s0 += P.view_w[(r + 0) * ldv + o] * M[(r + 0) * W + c];
            s1 += P.view_w[(r + 1) * ldv + o] * M[(r + 1) * W + c];
            s2 += P.view_w[(r + 2) * ldv + o] * M[(r + 2) * W + c];
            s3 += P.view_w[(r + 3) * ldv + o] * M[(r + 3) * W + c];
        }
        val = (s0 + s1) + (s2 + s3);
        dst = grad_flat + P.off_extra_w + i;
    } else if ((i -= W * W) < Wc * W) {                // dW_view[r][c] = sum_k M[r][k] W_extra[c][k] + dbv[r] b_extra[c]
        const int r = i / W, c = i - r * W;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll 2
        for (int k = 0; k < W; k += 4) {                   // transposed copy of W_extra: coalesced over c
            s0 += M[r * W + k + 0] * P.extra_wT[(k + 0) * W + c];
            s1 += M[r * W + k + 1] * P.extra_wT[(k + 1) * W + c];
            s2 += M[r * W + k + 2] * P.extra_wT[(k + 2) * W + c];
            s3 += M[r * W + k + 3] * P.extra_wT[(k + 3) * W + c];
        }
        val = (s0 + s1) + (s2 + s3) + dbv[r] * P.extra_b[c];
        dst = grad_flat + P.off_view_w + r * ldv + c;
    } else if ((i -= Wc * W) < W) {                    // db_extra[o] = sum_r W_view[r][o] dbv[r]
        float s = 0.0f;
        for (int r = 0; r < Wc; ++r) s += P.view_w[r * ldv + i] * dbv[r];
        val = s;
        dst = grad_flat + P.off_extra_b + i;
    } else if ((i -= W) < Wc) {                        // db_view
        val = dbv[i];
        dst = grad_flat + P.off_view_b + i;
    } else {
        return;
    }
    *dst = accumulate ? *dst + val : val;
}

__global__ void __launch_bounds__(256) k_transpose_sq(int n, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float t[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
    if (x < n && y < n) t[ty][tx] = in[(size_t)y * n + x];
    __syncthreads();
    const int ox = blockIdx.y * 16 + tx, oy = blockIdx.x * 16 + ty;
    if (ox < n && oy < n) out[(size_t)oy * n + ox] = t[tx][ty];
}

hipError_t launch_transpose_sq(int n, const float* in, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_transpose_sq, dim3((n + 15) / 16, (n + 15) / 16), dim3(256), 0, st, n, in, out);
    return hipGetLastError();
}

int mlp_wgrad_lds_bytes() { return kWgradLds; }

__global__ void k_wgrad_record_enc(WgradEnc* rec, const void* enc, int64_t M, int row_bytes, int frag_ksteps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { rec->enc = enc; rec->M = M; rec->row_bytes = row_bytes; rec->frag_ksteps = frag_ksteps; }
}
hipError_t launch_wgrad_record_enc(void* record, const void* enc, int64_t M, int row_bytes, hipStream_t st, int frag_ksteps) {
    hipLaunchKernelGGL(k_wgrad_record_enc, dim3(1), dim3(64), 0, st, (WgradEnc*)record, enc, M, row_bytes, frag_ksteps);
    return hipGetLastError();
}

hipError_t launch_mlp_wgrad(const void* HT, const void* GT, const WgradJob* jobs, const void* wg_tab, int num_wgs,
                            int64_t n_wt, int NH, int NG, float* partials, hipStream_t st, const WgradEnc* enc) {
    static int attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, kWgradLds);
        if (er != hipSuccess) return er;
        attr_done[dev] = 1;
    }
    hipLaunchKernelGGL(k_mlp_wgrad, dim3(num_wgs), dim3(512), kWgradLds, st, (const char*)HT, (const char*)GT, jobs,
                       (const int4*)wg_tab, n_wt, NH, NG, partials, enc);
    return hipGetLastError();
}

hipError_t launch_wgrad_reduce(const float* partials, const int32_t* otab, const void* job_slots, int njobs,
                               float* grad_flat, float* scratch, int nparams, const WgradPost& post, bool accumulate,
                               hipStream_t st) {
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((kWgradJobFloats + 63) / 64, njobs), dim3(256), 0, st, partials, otab,
                       (const int2*)job_slots, grad_flat, scratch, nparams, accumulate ? 1 : 0);
    if (post.W == 0) return hipGetLastError();        // architecture without a bottleneck (use_viewdirs=False): nothing to post-process
    const int n = post.W * post.W + post.Wc * post.W + post.W + post.Wc;
    hipLaunchKernelGGL(k_wgrad_post, dim3((n + 255) / 256), dim3(256), 0, st, post, scratch, grad_flat, accumulate ? 1 : 0);
    return hipGetLastError();
}

}  // namespace mip
