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

__global__ void __launch_bounds__(256)
k_gemm_reduce(int M, int N, int splits, const float* __restrict__ partial, float* __restrict__ C, int64_t ldc, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    float s = 0.0f;
    for (int z = 0; z < splits; ++z) s += partial[(int64_t)z * M * N + i];
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
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, M, N, splits, partial, C, ldc,
                           accumulate ? 1 : 0);
    }
    return hipGetLastError();
}

hipError_t launch_relu_mask(int64_t n, const float* x, float* g, hipStream_t st) {
    hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, x, g);
    return hipGetLastError();
}

}  // namespace mip
