// fp32 GEMM on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32) for the PARITY-MODE backward of the
// Mip-NeRF MLP (what torch autograd derives from models/mip_nerf.py:75-111): dgrad  dX = dY W  and wgrad
// dW = dY^T X, db = dY^T 1, with fp32 products and fp32 accumulation like the reference's CPU GEMMs (only the
// summation order differs).  Correctness-first: 64x64 output tile per workgroup (4 waves, one 32x32 MFMA tile each),
// K staged through LDS in blocks of 32, split-K with a deterministic second pass for the sample-contracted products.
// The performance path is the bf16 one (kernels_wgrad.hip, generated dgrad); this file exists so that fp32 training
// needs no library GEMM either.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mip {

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
constexpr int BM = 64, BN = 64, BK = 32;

// C[M,N] (row-major, ldc) = sum_k A(m,k) B(k,n), optionally restricted to k in this split's range.
//   TA = false: A(m,k) = A[m*lda + k]        (activations / deltas as [rows, features])
//   TA = true : A(m,k) = A[k*lda + m]        (contraction over rows = samples)
//   B(k,n) = B[(k / b_row_div) * ldb + n]    (b_row_div > 1: one B row per ray, e.g. the view encoding);
//   b_ones: B(k,n) = 1 (column sums).
// out: C if gridDim.z == 1, else partial[z][M][N] (summed by k_gemm_reduce).
template <bool TA>
__global__ void __launch_bounds__(256)
k_gemm_f32(int M, int N, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
           int b_row_div, int b_ones, float* __restrict__ C, int64_t ldc, int accumulate, float* __restrict__ partial) {
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int64_t kchunk = (K + gridDim.z - 1) / gridDim.z;
    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    const int64_t kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        // ---- stage A and B tiles (zero-filled outside the matrix / the split's k range)
        for (int i = tid; i < BM * BK; i += 256) {
            int mm, kk;
            if (TA) { mm = i % BM; kk = i / BM; } else { kk = i % BK; mm = i / BK; }     // fastest index = contiguous in memory
            const int64_t m = m0 + mm, k = k0 + kk;
            float v = 0.0f;
            if (m < M && k < kend) v = TA ? A[k * lda + m] : A[m * lda + k];
            As[kk][mm] = v;
        }
        for (int i = tid; i < BK * BN; i += 256) {
            const int nn = i % BN, kk = i / BN;
            const int64_t k = k0 + kk;
            const int n = n0 + nn;
            float v = 0.0f;
            if (n < N && k < kend) v = b_ones ? 1.0f : B[(k / b_row_div) * ldb + n];
            Bs[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D tile: lane (hi, n) register r <-> row (r&3) + 8(r>>2) + 4 hi, column n
    const int hi = lane >> 5, n = n0 + wn * 32 + (lane & 31);
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < M) {
                if (gridDim.z > 1) partial[((int64_t)blockIdx.z * M + m) * N + n] = acc[r];
                else C[m * ldc + n] = accumulate ? C[m * ldc + n] + acc[r] : acc[r];
            }
        }
    }
}

// ---- the big GEMMs of the backward: 256 x 128 output tile per workgroup (8 waves, each 64 x 64 = four 32x32 MFMA tiles),
// K in blocks of KB through a double-buffered LDS stage, global loads as float4 one block ahead (registers), so the MFMAs of
// a block (KB = 16: 32 per wave = 2048 cycles) hide the next block's loads.
// Tile shape: at fp32 MFMA rate a CU retires 292 FLOP/cycle; a 128 x 128 tile moves (128 + 128) x 4 B per 2 x 128 x 128 FLOP
// = 9.1 B/cycle/CU, which IS the ~10 B/cycle/CU a CU can pull through global_load_dwordx4 (measured: 85 TF/s, 0.54);
// 256 x 128 needs 6.8 B/cycle (wgrad: 0.52 ms = 132 TF/s = 0.84).
//   TA = false (dgrad, dX = dY W):  A(m,k) = A[m*lda + k], rows = samples.  K is only the layer width (256), so a tile's
//                                   output (C write + ReLU-mask read, 256 KB) weighs as much as its input (A 256 KB + B 128 KB):
//                                   640 KB per 65k MFMA cycles = 9.8 B/cycle/CU, again the CU's memory pipeline: 0.76 ms =
//                                   90 TF/s.  Tried and worse: KB = 32 for whole 128-byte lines per row (100 KB of LDS, one
//                                   workgroup per CU: 1.32 ms); a 128 x 256 tile that fetches every dY row once (0.93 ms).  Epilogue:
//                                   optional rank-1 term v += r1_col[m * r1_ld] * r1_row[n] (the density head's dgrad) and the
//                                   ReLU mask of the layer input (relu_x[m*ldc + n] <= 0 -> 0), fusing two separate passes
//   TA = true  (wgrad, dW = dY^T X): A(m,k) = A[k*lda + m], tile 256 x 128, contraction over samples, split-K over gridDim.z; the
//                                   workgroups of column-tile 0 also produce the bias gradient db[m] = sum_k A(m,k)
//   B(k,n) = B[k*ldb + n].
constexpr int GT = 512;      // GT threads = 8 waves; tile = TM x TN with TM * TN = 256 * 128
struct GemmEpi {
    const float* relu_x;     // [M, ldc] or null
    const float* r1_col;     // [M] with stride r1_ld, or null
    const float* r1_row;     // [N]
    int64_t r1_ld;
    const unsigned long long* relu_bits;   // sign bits of relu_x written by k_mlp_f32 (f32_bits_slot_words layout, vpr = ldc / 4) or null
};
template <bool TA, int KB, int GM, int GN>
__global__ void __launch_bounds__(GT)
k_gemm_f32_big(int M, int N, int64_t K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
               float* __restrict__ C, int64_t ldc, int accumulate, float* __restrict__ partial, GemmEpi epi,
               float* __restrict__ bias_partial, int b_vec) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    constexpr int LDA_S = GM + 4, LDB_S = GN + 4;
    float* const As = gsm;                                   // [2][KB][LDA_S]
    float* const Bs = gsm + 2 * KB * LDA_S;                  // [2][KB][LDB_S]
    constexpr int NA = GM * KB / 4 / GT, NB = KB * GN / 4 / GT;     // float4 loads per thread per block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WN = GN / 64;
    const int wm = wave / WN, wn = wave % WN;     // wave tile: rows [64 wm, +64), columns [64 wn, +64)
    const int64_t m0 = (int64_t)blockIdx.x * GM;
    const int n0 = blockIdx.y * GN;
    const int64_t kchunk = ((K + gridDim.z - 1) / gridDim.z + KB - 1) / KB * KB;
    const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
    const int64_t kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float4 ra[NA], rb[NB];
    auto load_tiles = [&](int64_t k0) {
#pragma unroll
        for (int h = 0; h < NA; ++h) {
            const int f = tid + h * GT;               // float4 index in the A tile
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TA) {                                 // A(m,k) = A[k*lda + m]: contiguous along m
                const int kk = f / (GM / 4), mm = (f % (GM / 4)) * 4;
                const int64_t k = k0 + kk, m = m0 + mm;
                if (k < kend) {
                    const float* src = A + k * lda + m;
                    if (m + 3 < M) v = *reinterpret_cast<const float4*>(src);
                    else { if (m < M) v.x = src[0]; if (m + 1 < M) v.y = src[1]; if (m + 2 < M) v.z = src[2]; }
                }
            } else {                                  // A(m,k) = A[m*lda + k]: contiguous along k, KB / 4 lanes per row
                const int mm = f / (KB / 4), kk = (f % (KB / 4)) * 4;
                const int64_t k = k0 + kk, m = m0 + mm;
                if (m < M) {
                    const float* src = A + m * lda + k;
                    if (k + 3 < kend) v = *reinterpret_cast<const float4*>(src);
                    else { if (k < kend) v.x = src[0]; if (k + 1 < kend) v.y = src[1]; if (k + 2 < kend) v.z = src[2]; }
                }
            }
            ra[h] = v;
        }
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int f = tid + h * GT;
            const int kk = f / (GN / 4), nn = (f % (GN / 4)) * 4;
            const int64_t k = k0 + kk;
            const int n = n0 + nn;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) {
                const float* src = B + k * ldb + n;
                if (b_vec && n + 3 < N) v = *reinterpret_cast<const float4*>(src);
                else { if (n < N) v.x = src[0]; if (n + 1 < N) v.y = src[1]; if (n + 2 < N) v.z = src[2]; if (n + 3 < N) v.w = src[3]; }
            }
            rb[h] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        float* as = As + buf * KB * LDA_S;
        float* bs = Bs + buf * KB * LDB_S;
#pragma unroll
        for (int h = 0; h < NA; ++h) {
            const int f = tid + h * GT;
            if (TA) {
                const int kk = f / (GM / 4), mm = (f % (GM / 4)) * 4;
                *reinterpret_cast<float4*>(as + kk * LDA_S + mm) = ra[h];
            } else {
                const int mm = f / (KB / 4), kk = (f % (KB / 4)) * 4;
                as[kk * LDA_S + mm] = ra[h].x; as[(kk + 1) * LDA_S + mm] = ra[h].y;
                as[(kk + 2) * LDA_S + mm] = ra[h].z; as[(kk + 3) * LDA_S + mm] = ra[h].w;
            }
        }
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int f = tid + h * GT;
            const int kk = f / (GN / 4), nn = (f % (GN / 4)) * 4;
            *reinterpret_cast<float4*>(bs + kk * LDB_S + nn) = rb[h];
        }
    };
    float bsum = 0.0f;                               // bias gradient: thread t < GM sums row m0 + t of A over this split's k range
    const bool want_bias = TA && bias_partial != nullptr && blockIdx.y == 0;
    int buf = 0;
    if (kbeg < kend) {
        load_tiles(kbeg);
        store_tiles(0);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += KB) {
        const bool more = k0 + KB < kend;
        if (more) load_tiles(k0 + KB);               // next block's global loads fly during this block's MFMAs
        const float* as = As + buf * KB * LDA_S;
        const float* bs = Bs + buf * KB * LDB_S;
        if (want_bias && tid < GM) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) bsum += as[kk * LDA_S + tid];
        }
#pragma unroll
        for (int kk = 0; kk < KB; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a0 = as[kr * LDA_S + wm * 64 + (lane & 31)], a1 = as[kr * LDA_S + wm * 64 + 32 + (lane & 31)];
            const float b0 = bs[kr * LDB_S + wn * 64 + (lane & 31)], b1 = bs[kr * LDB_S + wn * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_tiles(buf ^ 1);              // the other buffer: its last readers passed the previous barrier
        __syncthreads();
        buf ^= 1;
    }
    // D tile: lane (hi, n) register r <-> row (r&3) + 8(r>>2) + 4 hi, column n
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= N) continue;
            const float r1w = epi.r1_col ? epi.r1_row[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (gridDim.z > 1) { partial[((int64_t)blockIdx.z * M + m) * N + n] = v; continue; }
                if (epi.r1_col) v += epi.r1_col[m * epi.r1_ld] * r1w;
                if (epi.relu_bits) {          // 1 bit instead of 4 bytes per element: float4 number F of the [M, ldc] slot, component n & 3
                    const int64_t F = m * (ldc >> 2) + (n >> 2);
                    if (!((epi.relu_bits[(F >> 6) * 4 + (n & 3)] >> (F & 63)) & 1ull)) v = 0.0f;
                } else if (epi.relu_x && !(epi.relu_x[m * ldc + n] > 0.0f)) v = 0.0f;
                C[m * ldc + n] = accumulate ? C[m * ldc + n] + v : v;
            }
        }
    if (want_bias && tid < GM && m0 + tid < M) bias_partial[(int64_t)blockIdx.z * M + m0 + tid] = bsum;
}

// ---- thin weight gradients (colour / density heads: 1-3 output rows; the 27 view features): HBM-bound reductions, not GEMMs.
//   out[c * ldo_c + r * ldo_r] (+)= sum_s Xc[s * ldxc + c] * Yr[(s / rowdiv) * ldyr + r],   c < C ("wide", coalesced), r < R <= 32
// plus, when `bias`, one extra column c == C with Xc == 1 (column sums of Yr -> out_bias[r]).  A workgroup = 64 columns x
// 4 sample lanes over a slice of the samples; per-slice partials are summed by k_thin_reduce in slice order (deterministic).
template <int RMAX>
__global__ void __launch_bounds__(256)
k_thin_wgrad(int64_t S, int C, int R, const float* __restrict__ Xc, int64_t ldxc, const float* __restrict__ Yr, int64_t ldyr,
             int rowdiv, int bias, int64_t slice, float* __restrict__ partial) {
    __shared__ float red[4][64][RMAX + 1];
    const int col_l = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col_l;
    const int CC = C + (bias ? 1 : 0);
    const int64_t s0 = (int64_t)blockIdx.y * slice;
    const int64_t s1 = s0 + slice < S ? s0 + slice : S;
    float acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = 0.0f;
    if (c < CC && rowdiv > 1) {
        // Yr has one row per GROUP of `rowdiv` consecutive samples (a ray's view features): sum the group's Xc first, then
        // one multiply-add per row and group -- rowdiv times fewer FMAs and broadcast loads (slices are whole groups)
        for (int64_t g0 = s0; g0 < s1; g0 += rowdiv) {
            const int64_t g1 = g0 + rowdiv < s1 ? g0 + rowdiv : s1;
            float xs = 0.0f;
#pragma unroll 8
            for (int64_t s = g0 + sl; s < g1; s += 4) xs += c < C ? Xc[s * ldxc + c] : 1.0f;
            const float* y = Yr + (g0 / rowdiv) * ldyr;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) acc[r] = fmaf(xs, y[r], acc[r]);
        }
    } else if (c < CC) {
        for (int64_t s = s0 + sl; s < s1; s += 4) {
            const float x = c < C ? Xc[s * ldxc + c] : 1.0f;
            const float* y = Yr + s * ldyr;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) acc[r] = fmaf(x, y[r], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) red[sl][col_l][r] = acc[r];
    __syncthreads();
    if (sl == 0 && c < CC)
        for (int r = 0; r < R; ++r)
            partial[((int64_t)blockIdx.y * R + r) * CC + c] = ((red[0][col_l][r] + red[1][col_l][r]) + red[2][col_l][r]) + red[3][col_l][r];
}

// Same contract for the heads proper (R <= 4 output rows, rowdiv == 1, C % 4 == 0, 16-byte aligned rows): a thread owns FOUR
// consecutive columns, so a wavefront reads whole 1-KiB row segments (float4 per lane) with several rows in flight, instead of
// 256-byte pieces one row at a time (0.31 -> 0.1x ms per head at 524,288 x 256).  The bias column is summed by lane 0 of column
// block 0 from the same (wave-uniform) Yr values.
__global__ void __launch_bounds__(256)
k_thin_wgrad_v4(int64_t S, int C, int R, const float* __restrict__ Xc, int64_t ldxc, const float* __restrict__ Yr, int64_t ldyr,
                int bias, int64_t slice, float* __restrict__ partial) {
    __shared__ float red[4][64][20];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 256 + lane * 4;
    const int CC = C + (bias ? 1 : 0);
    const int64_t s0 = (int64_t)blockIdx.y * slice;
    const int64_t s1 = s0 + slice < S ? s0 + slice : S;
    float acc[4][4], ysum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ysum[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r][k] = 0.0f;
    }
    const bool colok = c0 < C;
#pragma unroll 8
    for (int64_t s = s0 + sl; s < s1; s += 4) {
        const float4 x = colok ? *reinterpret_cast<const float4*>(Xc + s * ldxc + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* y = Yr + s * ldyr;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < R) {
                const float yv = y[r];
                acc[r][0] = fmaf(x.x, yv, acc[r][0]); acc[r][1] = fmaf(x.y, yv, acc[r][1]);
                acc[r][2] = fmaf(x.z, yv, acc[r][2]); acc[r][3] = fmaf(x.w, yv, acc[r][3]);
                ysum[r] += yv;
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[sl][lane][r * 4 + k] = acc[r][k];
        red[sl][lane][16 + r] = ysum[r];
    }
    __syncthreads();
    if (sl != 0) return;
    for (int r = 0; r < R; ++r) {
        float* dst = partial + ((int64_t)blockIdx.y * R + r) * CC;
        if (colok)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                dst[c0 + k] = ((red[0][lane][r * 4 + k] + red[1][lane][r * 4 + k]) + red[2][lane][r * 4 + k]) + red[3][lane][r * 4 + k];
        if (bias && blockIdx.x == 0 && lane == 0)
            dst[C] = ((red[0][0][16 + r] + red[1][0][16 + r]) + red[2][0][16 + r]) + red[3][0][16 + r];
    }
}

__global__ void __launch_bounds__(256)
k_thin_reduce(int nslices, int C, int R, int bias, const float* __restrict__ partial, float* __restrict__ out, int64_t ldo_c,
              int64_t ldo_r, float* __restrict__ out_bias, int accumulate) {
    // 64 outputs x 4 slice groups per workgroup (fixed combination order), as k_gemm_reduce
    __shared__ float red[4][64];
    const int CC = C + (bias ? 1 : 0);
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    const bool live = i < R * CC;
    const int r = live ? i / CC : 0, c = live ? i - r * CC : 0;
    float s = 0.0f;
    if (live) {
#pragma unroll 8
        for (int z = g; z < nslices; z += 4) s += partial[((int64_t)z * R + r) * CC + c];
    }
    red[g][o] = s;
    __syncthreads();
    if (g != 0 || !live) return;
    s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    float* dst = c < C ? out + c * ldo_c + r * ldo_r : out_bias + r;
    *dst = accumulate ? *dst + s : s;
}

__global__ void __launch_bounds__(256)
k_gemm_reduce(int M, int N, int splits, const float* __restrict__ partial, float* __restrict__ C, int64_t ldc, int accumulate) {
    // 64 outputs x 4 split groups per workgroup: group g sums splits g, g+4, ... (eight loads in flight), the four group sums
    // are combined in a fixed order -> deterministic, and 4x the memory-level parallelism of one thread per output
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t mn = (int64_t)M * N;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    float s = 0.0f;
    if (i < mn) {
        const float* p = partial + i;
#pragma unroll 8
        for (int z = g; z < splits; z += 4) s += p[(int64_t)z * mn];
    }
    red[g][o] = s;
    __syncthreads();
    if (g != 0 || i >= mn) return;
    s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    const int64_t m = i / N, n = i - m * N;
    C[m * ldc + n] = accumulate ? C[m * ldc + n] + s : s;
}

// g[s, c] *= (x[s, c] > 0)   (ReLU backward with the saved post-activation)
__global__ void __launch_bounds__(256)
k_relu_mask(int64_t n, const float* __restrict__ x, float* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(x[i] > 0.0f)) g[i] = 0.0f;
}
}  // namespace

hipError_t launch_gemm_f32(bool trans_a, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           int b_row_div, bool b_ones, float* C, int64_t ldc, bool accumulate, int splits, float* partial,
                           hipStream_t st) {
    if (splits < 1) splits = 1;
    const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN), (unsigned)splits);
    if (trans_a)
        hipLaunchKernelGGL(k_gemm_f32<true>, grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, b_row_div, b_ones ? 1 : 0, C, ldc,
                           accumulate ? 1 : 0, partial);
    else
        hipLaunchKernelGGL(k_gemm_f32<false>, grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, b_row_div, b_ones ? 1 : 0, C, ldc,
                           accumulate ? 1 : 0, partial);
    if (splits > 1) {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, M, N, splits, partial, C, ldc,
                           accumulate ? 1 : 0);
    }
    return hipGetLastError();
}

bool gemm_f32_big_ok(int M, int N, int64_t K, const float* A, int64_t lda) {
    return M >= 64 && N >= 64 && K >= 64 && lda % 4 == 0 && (uintptr_t)A % 16 == 0;
}

// Big-GEMM entry: same contract as launch_gemm_f32 plus (a) dgrad epilogue (splits == 1): optional rank-1 term
// r1_col[m * r1_ld] * r1_row[n] and ReLU mask relu_x[m*ldc + n] <= 0 -> 0; (b) `bias_out` (wgrad only): db[m] (+)= sum_k A(m,k),
// reduced over the splits like C.  `partial` must hold splits * M * (N + 1) floats.
hipError_t launch_gemm_f32_big(bool trans_a, int M, int N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                               float* C, int64_t ldc, bool accumulate, int splits, float* partial, const float* relu_x,
                               const float* r1_col, int64_t r1_ld, const float* r1_row, float* bias_out, hipStream_t st,
                               const unsigned long long* relu_bits) {
    if (splits < 1) splits = 1;
    if (!gemm_f32_big_ok(M, N, K, A, lda) || ((relu_x || r1_col) && (splits != 1 || trans_a)) || (bias_out && !trans_a))
        return hipErrorInvalidValue;
    const int b_vec = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    float* bias_partial = bias_out ? partial + (int64_t)splits * M * N : nullptr;
    const GemmEpi epi = {relu_x, r1_col, r1_row, r1_ld, relu_x && ldc % 4 == 0 ? relu_bits : nullptr};
    constexpr int kLds = 2 * 16 * (256 + 4 + 128 + 4) * 4;
    static int attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t er = hipFuncSetAttribute((const void*)k_gemm_f32_big<true, 16, 256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (er != hipSuccess) return er;
        er = hipFuncSetAttribute((const void*)k_gemm_f32_big<false, 16, 256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (er != hipSuccess) return er;
        attr_done[dev] = 1;
    }
    if (trans_a) {
        const dim3 grid((unsigned)((M + 255) / 256), (unsigned)((N + 127) / 128), (unsigned)splits);
        hipLaunchKernelGGL((k_gemm_f32_big<true, 16, 256, 128>), grid, dim3(GT), kLds, st, M, N, K, A, lda, B, ldb, C, ldc,
                           accumulate ? 1 : 0, partial, epi, bias_partial, b_vec);
    } else {
        const dim3 grid((unsigned)((M + 255) / 256), (unsigned)((N + 127) / 128), (unsigned)splits);
        hipLaunchKernelGGL((k_gemm_f32_big<false, 16, 256, 128>), grid, dim3(GT), kLds, st, M, N, K, A, lda, B, ldb, C, ldc,
                           accumulate ? 1 : 0, partial, epi, bias_partial, b_vec);
    }
    if (splits > 1) {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, M, N, splits, partial, C, ldc,
                           accumulate ? 1 : 0);
    }
    if (bias_out)
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, st, M, 1, splits, bias_partial, bias_out,
                           (int64_t)1, accumulate ? 1 : 0);
    return hipGetLastError();
}

// thin weight gradient (see k_thin_wgrad).  `partial` needs ceil(S / 2048) * max(R, 16) * (C + 1) floats (the float4 path for
// R <= 4 cuts 512-sample slices).
hipError_t launch_thin_wgrad(int64_t S, int C, int R, const float* Xc, int64_t ldxc, const float* Yr, int64_t ldyr, int rowdiv,
                             float* out, int64_t ldo_c, int64_t ldo_r, float* out_bias, bool accumulate, float* partial,
                             hipStream_t st) {
    if (R < 1 || R > 32 || C < 1 || S < 1) return hipErrorInvalidValue;
    const int64_t slice = rowdiv > 1 ? (int64_t)rowdiv * ((2048 + rowdiv - 1) / rowdiv) : 2048;     // whole groups per slice
    const int nslices = (int)((S + slice - 1) / slice);
    const int bias = out_bias ? 1 : 0;
    const dim3 grid((unsigned)((C + bias + 63) / 64), (unsigned)nslices);
    if (R <= 4 && rowdiv == 1 && C % 4 == 0 && ldxc % 4 == 0 && (uintptr_t)Xc % 16 == 0) {
        // 512-sample slices: ~4 workgroups per CU and 8 rows in flight per wave cover the HBM latency (2048-sample slices with
        // 4 in flight streamed at 1.7 TB/s); 4 x the slices of the generic path at <= 1/8 of its rows: same partial buffer
        const int64_t slice4 = 512;
        const int nsl4 = (int)((S + slice4 - 1) / slice4);
        hipLaunchKernelGGL(k_thin_wgrad_v4, dim3((unsigned)((C + 255) / 256), (unsigned)nsl4), dim3(256), 0, st, S, C, R, Xc, ldxc, Yr,
                           ldyr, bias, slice4, partial);
        const int n4 = R * (C + bias);
        hipLaunchKernelGGL(k_thin_reduce, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, nsl4, C, R, bias, partial, out, ldo_c, ldo_r,
                           out_bias, accumulate ? 1 : 0);
        return hipGetLastError();
    }
    if (R <= 4)
        hipLaunchKernelGGL(k_thin_wgrad<4>, grid, dim3(256), 0, st, S, C, R, Xc, ldxc, Yr, ldyr, rowdiv, bias, slice, partial);
    else
        hipLaunchKernelGGL(k_thin_wgrad<32>, grid, dim3(256), 0, st, S, C, R, Xc, ldxc, Yr, ldyr, rowdiv, bias, slice, partial);
    const int n = R * (C + bias);
    hipLaunchKernelGGL(k_thin_reduce, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, nslices, C, R, bias, partial, out, ldo_c, ldo_r,
                       out_bias, accumulate ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_relu_mask(int64_t n, const float* x, float* g, hipStream_t st) {
    hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x, g);
    return hipGetLastError();
}

}  // namespace mip
