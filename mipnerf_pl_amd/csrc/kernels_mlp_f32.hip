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
// the B operand two ds_read_b128 of X per 32 samples.  Wave w owns out-tile w and both 32-sample halves.
// The two THIN heads -- density (1 output row) and colour (3 rows) -- would each occupy a whole 32-row MFMA tile (97 %
// zero rows) and serialise the workgroup behind one wave; they run on the VALU instead: wave w accumulates the k-slice
// [K w / 8, K (w+1) / 8) of every sample's dot products as plain fp32 fma chains (weights by scalar loads from the fp32
// master parameters), the partials meet in spare LDS columns and are summed in wave order.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "raymath.hpp"

namespace mip {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kF32Waves = 8;
constexpr int kF32Rounds = 2;   // hidden tiles per wave: widths up to 512 (16 tiles; a 256-wide layer skips round 1); thin heads: VALU

// TS = samples per workgroup tile: 64 (two 32-sample MFMA halves per weight chunk; every shape whose LDS rows fit: 64 x ldx x 4 B
// <= 160 KiB) or 32 (one half; wide encodings such as the 672 off-axis features of the unbounded-scene model: half the reuse of
// every weight chunk, same arithmetic and summation order per sample)
// STREAM = the sample encoding is not staged in LDS: the layers that read it take that B operand from global memory (wide encodings;
// a separate instantiation, so that the resident-encoding kernel keeps its inner loop)
template <int TS, bool STREAM>
__global__ void __launch_bounds__(kF32Waves * 64)
k_mlp_f32(const F32Net net, const float* __restrict__ wstream, const float* __restrict__ bias_tab,
          const float* __restrict__ enc, const float* __restrict__ viewenc, float4* __restrict__ rgb_sigma,
          float4* __restrict__ raw_out, int64_t M, int num_samples, int ntiles_total, float density_bias,
          float rgb_padding, float* __restrict__ save, unsigned long long* __restrict__ save_bits,
          const float* __restrict__ dnoise, float dnoise_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* X = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, n = lane & 31;
    const int ldx = net.ldx;
    const int W = net.width;
    const int ecol = net.enc_col;

    // deferred copy of a layer's output into `save` (slot pend_slot = [M, pend_width] fp32) + its ReLU sign bits for the dgrad
    // epilogue (k_gemm_f32_big): float4 number F = s * vpr + c4 of the slot <-> bit F & 63 of the four words [(F >> 6) * 4 + k],
    // k = component; one wavefront covers one 64-float4 group (item strides are multiples of 64)
    int pend_slot = 0, pend_xout = 0, pend_width = 0, pend_done = 0, pend_items = 0;
    auto save_item = [&](int64_t s0, int it) {
        const int vpr = pend_width / 4;
        const int i = tid + it * (int)blockDim.x;
        const int r = i / vpr, c4 = i - r * vpr;
        const int64_t s = s0 + r;
        const float4 v = *reinterpret_cast<const float4*>(X + r * ldx + pend_xout + c4 * 4);
        const bool live = s < M;
        if (live) *reinterpret_cast<float4*>(save + (int64_t)pend_slot * M * W + s * pend_width + c4 * 4) = v;
        if (save_bits) {
            const unsigned long long b0 = __ballot(live && v.x > 0.0f), b1 = __ballot(live && v.y > 0.0f),
                                     b2 = __ballot(live && v.z > 0.0f), b3 = __ballot(live && v.w > 0.0f);
            const int64_t g = (s0 * vpr + i) >> 6;
            if (lane < 4 && (g << 6) < M * vpr)
                save_bits[(int64_t)pend_slot * f32_bits_slot_words(M, W) + g * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
        }
    };

    constexpr int kF32TileSamples = TS;
    constexpr int NT = TS / 32;                // 32-sample MFMA halves per tile
    for (int tile = blockIdx.x; tile < ntiles_total; tile += gridDim.x) {
        const int64_t s0 = (int64_t)tile * kF32TileSamples;
        __syncthreads();     // the previous tile's last reads of X are done
        // ---- stage the encoding: X[s][ecol + c] = enc[s0+s][c]  (not for wide encodings: their two readers stream it, see the k loop)
        if (!STREAM) {
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

        for (int L = 0; L < net.nlayers; ++L) {
            const F32Layer ly = net.layers[L];
            const int kbt = ly.kb0 + ly.kb1;
            const int mfma_tiles = ly.kind == 2 ? 0 : (ly.kind == 1 ? ly.ntiles - 1 : ly.ntiles);   // thin heads: VALU below
            f32x16 acc[kF32Rounds][2];
#pragma unroll
            for (int rd = 0; rd < kF32Rounds; ++rd) {
                const int t = rd * kF32Waves + wave;             // wave-uniform
                if (t < mfma_tiles) {
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
                    // streamed encoding segments (wide encodings): lane (hi, n) reads its 8 k values of k block kb straight from the
                    // global encoding rows of samples n and 32 + n, one k block ahead (the same values the LDS copy would hold)
                    const int gseg = STREAM ? ly.pad : 0;                      // wave-uniform; 0 at compile time in the resident kernel
                    const float* ge0 = nullptr;
                    const float* ge1 = nullptr;
                    if (STREAM && gseg) {
                        int64_t r0 = s0 + n, r1 = s0 + 32 + n;
                        if (r0 >= M) r0 = M - 1;
                        if (r1 >= M) r1 = M - 1;
                        ge0 = enc + r0 * net.xyz_dim + hi * 8;
                        ge1 = enc + r1 * net.xyz_dim + hi * 8;
                    }
                    // (plain code, no lambdas: by-reference captures would push the four registers to scratch)
#define MIP_F32_IS_GLOB(KB) (((KB) < ly.kb0 ? (gseg & 1) : (gseg & 2)) != 0)
#define MIP_F32_GLOAD(KB)                                                          \
    do {                                                                           \
        const int c_ = ((KB) < ly.kb0 ? (KB) : (KB) - ly.kb0) * 16;                \
        g00 = *reinterpret_cast<const float4*>(ge0 + c_);                          \
        g01 = *reinterpret_cast<const float4*>(ge0 + c_ + 4);                      \
        if (NT > 1) {                                                              \
            g10 = *reinterpret_cast<const float4*>(ge1 + c_);                      \
            g11 = *reinterpret_cast<const float4*>(ge1 + c_ + 4);                  \
        }                                                                          \
    } while (0)
                    float4 g00 = make_float4(0.f, 0.f, 0.f, 0.f), g01 = g00, g10 = g00, g11 = g00;
                    if (STREAM && gseg && kbt > 0 && MIP_F32_IS_GLOB(0)) MIP_F32_GLOAD(0);
                    for (int kb = 0; kb < kbt; ++kb) {
                        const int kn = kb + 1 < kbt ? kb + 1 : kb;
                        const float4 n0 = *reinterpret_cast<const float4*>(wp + (size_t)kn * 512);
                        const float4 n1 = *reinterpret_cast<const float4*>(wp + (size_t)kn * 512 + 4);
                        float4 b00, b01, b10, b11;
                        if (STREAM && gseg && MIP_F32_IS_GLOB(kb)) {
                            b00 = g00; b01 = g01;
                            if (NT > 1) { b10 = g10; b11 = g11; } else { b10 = g00; b11 = g01; }
                        } else {
                            const float* x0 = (kb < ly.kb0 ? xa : xb) + kb * 16;
                            const float* x1 = NT > 1 ? x0 + 32 * ldx : x0;
                            b00 = *reinterpret_cast<const float4*>(x0);
                            b01 = *reinterpret_cast<const float4*>(x0 + 4);
                            b10 = *reinterpret_cast<const float4*>(x1);
                            b11 = *reinterpret_cast<const float4*>(x1 + 4);
                        }
                        if (STREAM && gseg && kb + 1 < kbt && MIP_F32_IS_GLOB(kb + 1)) MIP_F32_GLOAD(kb + 1);      // next k block's B operand, under this block's 16 MFMAs
                        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                        const float b0[8] = {b00.x, b00.y, b00.z, b00.w, b01.x, b01.y, b01.z, b01.w};
                        const float b1[8] = {b10.x, b10.y, b10.z, b10.w, b11.x, b11.y, b11.z, b11.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            acc[rd][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[j], acc[rd][0], 0, 0, 0);
                            if (NT > 1) acc[rd][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[j], acc[rd][1], 0, 0, 0);
                        }
                        a0 = n0;
                        a1 = n1;
                        if (pend_done < pend_items) save_item(s0, pend_done++);     // previous layer's output -> save (see below)
                    }
                    // ---- this tile's results: the OTHER activation buffer (nobody reads it during this layer)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
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
            if (ly.kind != 0) {
                // ---- thin head on the VALU: lane = sample, wave = k-slice; nout = 1 (density, kind 1) or num_rgb (colour)
                const int nout = ly.kind == 1 ? 1 : net.num_rgb;
                const int K = ly.kb0 * 16;                          // in_features of the head (net_width / net_width_cond)
                const int ks = K / kF32Waves;                       // k-slice of this wave (K is a multiple of 32)
                const float* wrow = (ly.kind == 1 ? net.dens_w : net.col_w) + wave * ks;
                const int hl = lane < TS ? lane : TS - 1;             // TS = 32: the upper lane half shadows sample 31 (no stores)
                const float* xr = X + hl * ldx + ly.x_in0 + wave * ks;
                float part[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < ks; k += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < nout) {
                            const float* wc = wrow + c * K + k;     // wave-uniform address: scalar loads
                            part[c] = fmaf(xv.x, wc[0], part[c]);
                            part[c] = fmaf(xv.y, wc[1], part[c]);
                            part[c] = fmaf(xv.z, wc[2], part[c]);
                            part[c] = fmaf(xv.w, wc[3], part[c]);
                        }
                }
                // partials: spare encoding columns [ecol + 32, ecol + 32 + 8 * 4) of the sample's row (the view encoding uses
                // [ecol, ecol + 32); the integrated encoding's last reader is behind a barrier); density keeps slot 4*w + 3
                float* pr = X + hl * ldx + ecol + 32 + wave * 4;
                if (lane >= TS) {} else if (ly.kind == 1) pr[3] = part[0];
                else { pr[0] = part[0]; pr[1] = part[1]; pr[2] = part[2]; }
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
            while (pend_done < pend_items) save_item(s0, pend_done++);     // what the k loop did not cover (idle waves, thin heads)
            __syncthreads();     // this layer's outputs are visible; its input buffer is free
            if (ly.kind == 2 && wave == 0) {
                // finalise: sum the 8 k-slice partials in wave order, add the biases, activations (mip_nerf.py:232-238)
                const int64_t s = s0 + lane;
                if (s < M && lane < TS) {
                    const float* pr = X + lane * ldx + ecol + 32;
                    float r[3] = {0.f, 0.f, 0.f}, dn = 0.f;
#pragma unroll
                    for (int w = 0; w < kF32Waves; ++w) {
                        r[0] += pr[4 * w]; r[1] += pr[4 * w + 1]; r[2] += pr[4 * w + 2]; dn += pr[4 * w + 3];
                    }
                    const float r0 = r[0] + net.col_b[0], r1 = r[1] + net.col_b[1], r2 = r[2] + net.col_b[2];
                    dn += net.dens_b[0];
                    // mip_nerf.py:232-233: raw_density += density_noise * randn, before the activation
                    const float nd = dnoise ? dn + dnoise_scale * dnoise[s] : dn;
                    rgb_sigma[s] = make_float4(rgb_activation(r0, rgb_padding), rgb_activation(r1, rgb_padding),
                                               rgb_activation(r2, rgb_padding), density_activation(nd, density_bias));
                    if (raw_out) raw_out[s] = make_float4(r0, r1, r2, dn);
                }
            }
            if (save && ly.kind != 2) {
                // training (parity mode): keep this layer's output.  The copy is deferred into the NEXT layer's k loop (one
                // float4 per thread and k block, in the shadow of that layer's MFMAs): the next layer reads this buffer and
                // writes the other one, so the values stay put until the next layer's closing barrier
                pend_slot = L;
                pend_xout = ly.x_out;
                pend_width = 32 * (ly.ntiles - (ly.kind == 1 ? 1 : 0));     // the density tile of the head is not part of the bottleneck
                pend_done = 0;
                pend_items = kF32TileSamples * (pend_width / 4) / (int)blockDim.x;
            }
        }
        while (pend_done < pend_items) save_item(s0, pend_done++);     // the last layer's output (no next layer to hide behind)
    }
}

// samples per workgroup tile for an LDS row of ldx floats: 64 when it fits the CU's 160 KiB, else 32, else 0
int mlp_f32_tile_samples(int ldx) {
    const int cap = 160 * 1024;
    if (64 * ldx * (int)sizeof(float) <= cap) return 64;
    if (32 * ldx * (int)sizeof(float) <= cap) return 32;
    return 0;
}

hipError_t launch_mlp_f32(const F32Net& net_in, const float* stream_w, const float* bias_tab, const float* enc,
                          const float* viewenc, float* rgb_sigma, float* raw_out, int64_t M, int num_samples,
                          float density_bias, float rgb_padding, float* save, unsigned long long* save_bits, const float* dnoise,
                          float dnoise_scale, hipStream_t st) {
    const F32Net& net = net_in;
    if (!net.dens_w || !net.dens_b || !net.col_w || !net.col_b || net.num_rgb > 3 || net.width > 32 * kF32Waves * kF32Rounds ||
        net.ldx - net.enc_col - 4 < 64)      // VALU-head partials live in encoding columns [32, 64)
        return hipErrorInvalidValue;
    const int ts = mlp_f32_tile_samples(net.ldx);
    if (ts == 0) return hipErrorInvalidValue;      // not even a 32-sample tile fits the CU's LDS
    const int ntiles = (int)((M + ts - 1) / ts);
    const int lds = ts * net.ldx * (int)sizeof(float);
    const bool stream = net.pad != 0;
    if (stream && ts != 64) return hipErrorInvalidValue;       // the streamed-encoding kernel exists for 64-sample tiles
    static int attr_lds_dev[64][3] = {};      // the attribute is per device: a process may drive several GPUs
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int* attr_lds = attr_lds_dev[dev];
    const int which = stream ? 2 : (ts == 64 ? 1 : 0);
    if (attr_lds[which] < lds) {
        hipError_t er = which == 2 ? hipFuncSetAttribute((const void*)k_mlp_f32<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                      : which == 1 ? hipFuncSetAttribute((const void*)k_mlp_f32<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                                   : hipFuncSetAttribute((const void*)k_mlp_f32<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (er != hipSuccess) return er;
        attr_lds[which] = lds;
    }
    int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
    if (grid < 1) grid = 1;
#define MIP_F32_LAUNCH(TSV, STR)                                                                                                       \
    hipLaunchKernelGGL((k_mlp_f32<TSV, STR>), dim3(grid), dim3(kF32Waves * 64), lds, st, net, stream_w, bias_tab, enc, viewenc,         \
                       (float4*)rgb_sigma, (float4*)raw_out, M, num_samples, ntiles, density_bias, rgb_padding, save, save_bits, dnoise, \
                       dnoise_scale)
    if (which == 2) MIP_F32_LAUNCH(64, true);
    else if (which == 1) MIP_F32_LAUNCH(64, false);
    else MIP_F32_LAUNCH(32, false);
#undef MIP_F32_LAUNCH
    return hipGetLastError();
}

}  // namespace mip
