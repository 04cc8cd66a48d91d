// fp32 MLP of Mip-NeRF (reference: models/mip_nerf.py:75-111 + activations 236-238) on the
// exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (products and accumulation are plain
// fp32 fma chains -- this is the parity mode, and the compute mode of BASELINE configs[3]).
//
// One workgroup (8 wavefronts = two per SIMD, so one wave's MFMAs cover the other's operand loads)
// owns a tile of 64 samples whose activations live in LDS as X[64][ldx] fp32 in the reference's
// natural feature order, DOUBLE-BUFFERED:
//     cols [0, W)        buffer B  \  a layer reads one and writes the other: no barrier between
//     cols [W, 2W)       buffer A  /  its MFMAs and its stores, ONE barrier per layer
//     cols [2W, 2W+E)    the integrated positional encoding (layer 0 and the skip concat of
//                        mip_nerf.py:96-97); after the head layer the 32-padded view encoding
//                        (cat([bottleneck, view_direction]), mip_nerf.py:106-107)
// A layer's K dimension is one or two SEGMENTS of 16-wide k blocks (activation buffer, then
// encoding / view features), each with its own LDS column base -- the weight chunks are packed in
// that k order (capi.hip build_tables), so the per-output summation order is the reference's.
// Layers are computed swapped, D[out, sample] = W[out, k] * X^T[k, sample]: the A operand is
// a pre-packed 2-KiB chunk of the weight (32 out-rows x 16 k, 8 fp32 per lane, read straight
// from L2 with two dwordx4 loads, the next chunk in flight during the 16 MFMAs of the current one),
// the B operand two ds_read_b128 of X per 32 samples.  Wave w owns out-tile w (and tile w + 8, the
// density row of the head) and both 32-sample halves.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "raymath.hpp"

namespace mip {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kF32TileSamples = 64;
constexpr int kF32Waves = 8;
constexpr int kF32Rounds = 2;   // ceil(9 tiles / 8 waves); up to 16 tiles

__global__ void __launch_bounds__(kF32Waves * 64)
k_mlp_f32(const F32Net net, const float* __restrict__ wstream, const float* __restrict__ bias_tab,
          const float* __restrict__ enc, const float* __restrict__ viewenc, float4* __restrict__ rgb_sigma,
          float4* __restrict__ raw_out, int64_t M, int num_samples, int ntiles_total, float density_bias,
          float rgb_padding, float* __restrict__ save, const float* __restrict__ dnoise, float dnoise_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* X = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, n = lane & 31;
    const int ldx = net.ldx;
    const int W = net.width;
    const int ecol = net.enc_col;

    for (int tile = blockIdx.x; tile < ntiles_total; tile += gridDim.x) {
        const int64_t s0 = (int64_t)tile * kF32TileSamples;
        __syncthreads();     // the previous tile's last reads of X are done
        // ---- stage the encoding: X[s][ecol + c] = enc[s0+s][c]
        {
            const int vec_per_row = net.xyz_dim / 4;
            for (int i = tid; i < kF32TileSamples * vec_per_row; i += blockDim.x) {
                const int r = i / vec_per_row, c4 = i - r * vec_per_row;
                int64_t s = s0 + r;
                if (s >= M) s = M - 1;
                const float4 v = *reinterpret_cast<const float4*>(enc + s * net.xyz_dim + c4 * 4);
                *reinterpret_cast<float4*>(X + r * ldx + ecol + c4 * 4) = v;
            }
        }
        __syncthreads();

        float dens[2] = {0.0f, 0.0f};
        for (int L = 0; L < net.nlayers; ++L) {
            const F32Layer ly = net.layers[L];
            const int kbt = ly.kb0 + ly.kb1;
            f32x16 acc[kF32Rounds][2];
#pragma unroll
            for (int rd = 0; rd < kF32Rounds; ++rd) {
                const int t = rd * kF32Waves + wave;             // wave-uniform
                if (t < ly.ntiles) {
                    // accumulators start at the bias: lane (hi, .) register r <-> row (r&3)+8(r>>2)+4hi
                    const float* bp = bias_tab + ((size_t)(ly.first_tile + t) * 2 + hi) * 16;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bv = bp[r];
                        acc[rd][0][r] = bv;
                        acc[rd][1][r] = bv;
                    }
                    const float* wp = wstream + ((size_t)ly.chunk0 + (size_t)t * kbt) * 512 + lane * 8;
                    const float* xa = X + n * ldx + ly.x_in0 + hi * 8;                       // segment 0
                    const float* xb = X + n * ldx + ly.x_in1 + hi * 8 - ly.kb0 * 16;         // segment 1 (indexed by the global kb)
                    float4 a0 = *reinterpret_cast<const float4*>(wp);
                    float4 a1 = *reinterpret_cast<const float4*>(wp + 4);
                    for (int kb = 0; kb < kbt; ++kb) {
                        const int kn = kb + 1 < kbt ? kb + 1 : kb;
                        const float4 n0 = *reinterpret_cast<const float4*>(wp + (size_t)kn * 512);
                        const float4 n1 = *reinterpret_cast<const float4*>(wp + (size_t)kn * 512 + 4);
                        const float* x0 = (kb < ly.kb0 ? xa : xb) + kb * 16;
                        const float* x1 = x0 + 32 * ldx;
                        const float4 b00 = *reinterpret_cast<const float4*>(x0);
                        const float4 b01 = *reinterpret_cast<const float4*>(x0 + 4);
                        const float4 b10 = *reinterpret_cast<const float4*>(x1);
                        const float4 b11 = *reinterpret_cast<const float4*>(x1 + 4);
                        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                        const float b0[8] = {b00.x, b00.y, b00.z, b00.w, b01.x, b01.y, b01.z, b01.w};
                        const float b1[8] = {b10.x, b10.y, b10.z, b10.w, b11.x, b11.y, b11.z, b11.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            acc[rd][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc[rd][0], 0, 0, 0);
                            acc[rd][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc[rd][1], 0, 0, 0);
                        }
                        a0 = n0;
                        a1 = n1;
                    }
                    // ---- this tile's results: the OTHER activation buffer (nobody reads it during this layer)
                    const bool is_density = (ly.kind == 1) && (t == ly.ntiles - 1);
                    if (ly.kind == 2) {
                        // colour head: rows 0..2 of tile 0 = lanes hi==0, registers 0..2 (wave 0 only)
                        if (hi == 0) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                const int64_t s = s0 + nt * 32 + n;
                                if (s < M) {
                                    const float r0 = acc[rd][nt][0], r1 = acc[rd][nt][1], r2 = acc[rd][nt][2];
                                    const float dn = X[(nt * 32 + n) * ldx + net.dens_col];
                                    // mip_nerf.py:232-233: raw_density += density_noise * randn, before the activation
                                    const float nd = dnoise ? dn + dnoise_scale * dnoise[s] : dn;
                                    rgb_sigma[s] = make_float4(rgb_activation(r0, rgb_padding),
                                                               rgb_activation(r1, rgb_padding),
                                                               rgb_activation(r2, rgb_padding),
                                                               density_activation(nd, density_bias));
                                    if (raw_out) raw_out[s] = make_float4(r0, r1, r2, dn);
                                }
                            }
                        }
                    } else if (is_density) {
                        // row 0 of the density tile (lanes hi == 0, register 0): parked in a spare LDS column until the
                        // colour head (a different wave) needs it
                        if (hi == 0) {
                            X[n * ldx + net.dens_col] = acc[rd][0][0];
                            X[(32 + n) * ldx + net.dens_col] = acc[rd][1][0];
                        }
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            float* xo = X + (nt * 32 + n) * ldx + ly.x_out + t * 32 + hi * 4;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float4 v = make_float4(acc[rd][nt][4 * g], acc[rd][nt][4 * g + 1],
                                                       acc[rd][nt][4 * g + 2], acc[rd][nt][4 * g + 3]);
                                if (ly.relu) {
                                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                                    v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                                }
                                *reinterpret_cast<float4*>(xo + 8 * g) = v;
                            }
                        }
                    }
                }
            }
            if (ly.stage_view) {
                // the encoding has had its last reader (the skip layer is behind a barrier): put the (32-padded) view
                // encoding of each sample's ray at cols [ecol, ecol+32)  (mip_nerf.py:106-107 cat([bottleneck, view_direction]))
                for (int i = tid; i < kF32TileSamples * 8; i += blockDim.x) {
                    const int r = i >> 3, c4 = i & 7;
                    int64_t s = s0 + r;
                    if (s >= M) s = M - 1;
                    const int64_t ray = s / num_samples;
                    *reinterpret_cast<float4*>(X + r * ldx + ecol + c4 * 4) =
                        *reinterpret_cast<const float4*>(viewenc + ray * 32 + c4 * 4);
                }
            }
            __syncthreads();     // this layer's outputs are visible; its input buffer is free
            if (save && ly.kind != 2) {
                // training (parity mode): keep this layer's output -- slot L of `save` is [M, width] fp32, width =
                // 32 x (hidden tiles); the density tile of the head is not part of the bottleneck.  The next layer
                // writes the OTHER buffer, so no barrier is needed after this copy.
                const int width = 32 * (ly.ntiles - (ly.kind == 1 ? 1 : 0));
                float* dst = save + (int64_t)L * M * W;
                const int vpr = width / 4;
                for (int i = tid; i < kF32TileSamples * vpr; i += blockDim.x) {
                    const int r = i / vpr, c4 = i - r * vpr;
                    const int64_t s = s0 + r;
                    if (s < M)
                        *reinterpret_cast<float4*>(dst + s * width + c4 * 4) =
                            *reinterpret_cast<const float4*>(X + r * ldx + ly.x_out + c4 * 4);
                }
            }
        }
    }
}

hipError_t launch_mlp_f32(const F32Net& net, const float* stream_w, const float* bias_tab, const float* enc,
                          const float* viewenc, float* rgb_sigma, float* raw_out, int64_t M, int num_samples,
                          float density_bias, float rgb_padding, float* save, const float* dnoise, float dnoise_scale,
                          hipStream_t st) {
    const int ntiles = (int)((M + kF32TileSamples - 1) / kF32TileSamples);
    const int lds = kF32TileSamples * net.ldx * (int)sizeof(float);
    static int attr_lds = 0;
    if (attr_lds < lds) {
        hipError_t er = hipFuncSetAttribute((const void*)k_mlp_f32, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (er != hipSuccess) return er;
        attr_lds = lds;
    }
    int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_mlp_f32, dim3(grid), dim3(kF32Waves * 64), lds, st, net, stream_w, bias_tab, enc, viewenc,
                       (float4*)rgb_sigma, (float4*)raw_out, M, num_samples, ntiles, density_bias, rgb_padding, save, dnoise, dnoise_scale);
    return hipGetLastError();
}

}  // namespace mip
