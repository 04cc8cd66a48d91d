// Evaluation metrics on the device: eval_errors of utils/metrics.py:191-197 =
// PSNR (metrics.py:175-188) + mean SSIM with an 11x11 Gaussian window, sigma 1.5, zero padding
// (metrics.py:44-126), fused in one pass over the rendered frame: each 16x16 pixel block stages its 26x26 halo of both
// images in LDS, every thread evaluates the five windowed moments of its pixel for the 3 channels, block sums go to a
// partial buffer and a second tiny kernel folds them in fp64 (deterministic).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mip {
namespace {
constexpr int kT = 16, kR = 5, kHalo = kT + 2 * kR;   // 26

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
}  // namespace

__global__ void __launch_bounds__(256)
k_eval_errors(int H, int W, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial) {
    __shared__ float sa[3][kHalo][kHalo + 1], sb[3][kHalo][kHalo + 1];
    __shared__ float g[11];
    __shared__ float red[4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
    if (threadIdx.x < 11) {     // metrics.py:10-17: exp(-(x - 5)^2 / (2 sigma^2)) normalised
        float s = 0.0f, mine = 0.0f;
        for (int i = 0; i < 11; ++i) {
            const float e = expf(-(float)((i - 5) * (i - 5)) / 4.5f);
            s += e;
            if (i == (int)threadIdx.x) mine = e;
        }
        g[threadIdx.x] = mine / s;
    }
    for (int i = threadIdx.x; i < kHalo * kHalo; i += 256) {
        const int hy = i / kHalo, hx = i - hy * kHalo;
        const int y = y0 + hy - kR, x = x0 + hx - kR;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;       // zero padding (metrics.py:64-68)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sa[c][hy][hx] = in ? a[((size_t)y * W + x) * 3 + c] : 0.0f;
            sb[c][hy][hx] = in ? b[((size_t)y * W + x) * 3 + c] : 0.0f;
        }
    }
    __syncthreads();
    const bool live = (y0 + ty) < H && (x0 + tx) < W;
    float ssim_sum = 0.0f, se_sum = 0.0f;
    if (live) {
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
            for (int dy = 0; dy < 11; ++dy) {
                const float gy = g[dy];
#pragma unroll
                for (int dx = 0; dx < 11; ++dx) {
                    const float w = gy * g[dx];
                    const float p = sa[c][ty + dy][tx + dx], q = sb[c][ty + dy][tx + dx];
                    m1 += w * p; m2 += w * q; s11 += w * p * p; s22 += w * q * q; s12 += w * p * q;
                }
            }
            const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
            const float sig1 = s11 - mu1_sq, sig2 = s22 - mu2_sq, sig12 = s12 - mu12;
            ssim_sum += ((2.0f * mu12 + C1) * (2.0f * sig12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sig1 + sig2 + C2));
            const float d = sa[c][ty + kR][tx + kR] - sb[c][ty + kR][tx + kR];
            se_sum += d * d;
        }
    }
    const float bs = block_sum(ssim_sum, red);
    const float be = block_sum(se_sum, red);
    if (threadIdx.x == 0) {
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * blk] = bs;
        partial[2 * blk + 1] = be;
    }
}

__global__ void k_eval_finish(int nblk, double count, const float* __restrict__ partial, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0, e = 0;
    for (int i = 0; i < nblk; ++i) { s += partial[2 * i]; e += partial[2 * i + 1]; }
    out[0] = (float)(-10.0 * log10(e / count));     // calc_psnr, metrics.py:182-188
    out[1] = (float)(s / count);                    // ssim(..., reduction='mean')
}

int64_t eval_errors_partial_floats(int H, int W) { return 2LL * ((H + kT - 1) / kT) * ((W + kT - 1) / kT); }

hipError_t launch_eval_errors(int H, int W, const float* pred, const float* gt, float* partial, float* out, hipStream_t st) {
    const dim3 grid((W + kT - 1) / kT, (H + kT - 1) / kT);
    hipLaunchKernelGGL(k_eval_errors, grid, dim3(256), 0, st, H, W, pred, gt, partial);
    hipLaunchKernelGGL(k_eval_finish, dim3(1), dim3(64), 0, st, (int)(grid.x * grid.y), (double)H * W * 3.0, partial, out);
    return hipGetLastError();
}
}  // namespace mip
