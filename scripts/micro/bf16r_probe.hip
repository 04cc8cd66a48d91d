// Go / no-go probe of a k-step-major bf16 MLP kernel at ONE wave per SIMD (the form the unbounded-scene model's 672-wide first layer
// needs: its 42 encoding k-steps cannot be re-streamed once per two-tile panel, so all 8 output tiles of a layer must be live per
// k-step -- 128 accumulator registers, which leaves room for one wave per SIMD only).  VERDICT r03 #6 asks for >= 0.45 of the bf16 peak.
//   mode 0  register-fed: 8 accumulators, A fragments in registers
//   mode 1  one ds_read_b128 A fragment per MFMA from a resident LDS ring (1 KiB per MFMA and wave: 128 B/clk/CU at full rate)
//   mode 2  + ring refilled by LDS-DMA (32 KiB per 4 k-steps), barrier per group, one piece per MFMA pair
//   mode 3  + layer structure: the B operand of k-step k + 1 is built from the previous layer's accumulators (8 x relu, 4 x cvt_pk)
//           while k-step k's MFMAs run; accumulator images from LDS
// One "layer" = 16 k-steps (K = 16) x 8 output tiles = 128 MFMAs per wave.
//   hipcc -O3 --offload-arch=gfx950 -fno-honor-nans scripts/micro/bf16r_probe.hip -o scripts/micro/bf16r_probe.out
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int kGroupBytes = 32768;      // 32 chunks of 1 KiB = 4 k-steps x 8 tiles
constexpr int kRingBytes = 2 * kGroupBytes;
constexpr int kBiasBytes = 8 * 128;
constexpr int kStreamGroups = 38;       // 1.19 MiB, L2-resident

#define PIN() __builtin_amdgcn_sched_barrier(0)
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

// B operand of k-step (ti, half u): registers 8u .. 8u+7 of the previous layer's D tile ti, ReLU + bf16 (RNE)
template <int MODE, int U>
__device__ __forceinline__ bf16x8 make_b(const f32x16& t) {
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float v = MODE >= 3 ? __builtin_fmaxf(t[8 * U + r], 0.0f) : t[8 * U + r];
        o[r] = (__bf16)v;
    }
    return o;
}

template <int MODE>
__device__ __forceinline__ void layer(f32x16 (&IN)[8], f32x16 (&OUT)[8], const bf16x8 (&AR)[8], const char* ring_lane, const char* bias_lane,
                                      const char* stream_w, unsigned ring_w, int& group, unsigned lane16) {
    bf16x8 b = make_b<MODE, 0>(IN[0]);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int slot = (ks >> 2) & 1;
        const char* gp = stream_w;
        unsigned lp = ring_w;
        if (MODE >= 2 && (ks & 3) == 0) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            int g = group + 1;
            if (g >= kStreamGroups) g = 0;
            group = g;
        }
        if (MODE >= 2) {
            OPAQUE_S(gp);
            OPAQUE_S(lp);
            gp += (size_t)group * kGroupBytes + ((ks & 1) ? 4096 : 0);
            lp += (slot ^ 1) * kGroupBytes + ((ks & 1) ? 4096 : 0);
        }
        bf16x8 bn = b;
        if (ks < 15) bn = (ks & 1) ? make_b<MODE, 0>(IN[(ks + 1) >> 1]) : make_b<MODE, 1>(IN[ks >> 1]);
        if (MODE >= 3 && ks >= 2 && (ks & 1) == 0) IN[(ks >> 1) - 1] = *reinterpret_cast<const f32x16*>(bias_lane + ((ks >> 1) - 1) * 128);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            bf16x8 a;
            if (MODE >= 1) a = *reinterpret_cast<const bf16x8*>(ring_lane + slot * kGroupBytes + ((ks & 3) * 8 + t) * 1024);
            else a = AR[t];
            OUT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, OUT[t], 0, 0, 0);
            // 8 pieces per wave and group, over the first two k-steps' MFMA pairs
            if (MODE >= 2 && (ks & 3) < 2 && (t & 1)) {
                if ((t >> 1) == 0) dma_piece<0>(gp, lp, lane16);
                else if ((t >> 1) == 1) dma_piece<1024>(gp, lp, lane16);
                else if ((t >> 1) == 2) dma_piece<2048>(gp, lp, lane16);
                else dma_piece<3072>(gp, lp, lane16);
            }
        }
        PIN();
        b = bn;
    }
    if (MODE >= 3) IN[7] = *reinterpret_cast<const f32x16*>(bias_lane + 7 * 128);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(const char* __restrict__ stream, const float* __restrict__ xin, float* __restrict__ out, int pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane16 = lane * 16;
    f32x16 X[8], Y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            X[t][r] = xin[(t * 16 + r) * 64 + lane];
            Y[t][r] = 0.0f;
        }
    bf16x8 AR[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) AR[i] = reinterpret_cast<const bf16x8*>(stream)[i * 64 + lane];
    for (int i = threadIdx.x; i < kRingBytes / 16; i += blockDim.x)
        reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(stream)[i];
    for (int i = threadIdx.x; i < kBiasBytes / 4; i += blockDim.x)
        reinterpret_cast<float*>(smem + kRingBytes)[i] = 0.01f * (float)((i * 7) % 13 - 6);
    __syncthreads();
    const char* ring_lane = smem + lane16;
    const char* bias_lane = smem + kRingBytes + (lane >> 5) * 64;
    const char* stream_w = stream + wave * 8192;
    const unsigned ring_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
    int group = 0;
    for (int it = 0; it < pairs; ++it) {
        layer<MODE>(X, Y, AR, ring_lane, bias_lane, stream_w, ring_w, group, lane16);
        layer<MODE>(Y, X, AR, ring_lane, bias_lane, stream_w, ring_w, group, lane16);
        if (MODE < 3) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    X[t][r] = __builtin_fminf(__builtin_fmaxf(X[t][r], -1.0f), 1.0f);
                    Y[t][r] = 0.0f;
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += X[t][r] + Y[t][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
void run(const char* d_stream, const float* d_x, float* d_out, double secs, int cus) {
    const int lds = kRingBytes + kBiasBytes;
    CK(hipFuncSetAttribute((const void*)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int pairs = 256;     // 512 layers = 65,536 MFMAs per wave per launch
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double sum_ms = 0;
    int n = 0;
    for (int phase = 0; phase < 2; ++phase) {
        double elapsed = 0;
        while (elapsed < secs * 0.5) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_probe<MODE>, dim3(cus), dim3(256), lds, 0, d_stream, d_x, d_out, pairs);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            elapsed += ms * 1e-3;
            if (phase == 1) { sum_ms += ms; n += 4; }
        }
    }
    const double ms = sum_ms / n;
    const double flop = (double)cus * 4 * pairs * 2 * 128 * 32768.0;     // 32x32x16 MFMA = 32768 FLOP
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("{\"mode\": %d, \"ms_per_launch\": %.4f, \"tflops\": %.1f, \"frac_of_2500\": %.4f}\n", MODE, ms, tf, tf / 2500.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t nstream = (size_t)kStreamGroups * kGroupBytes / 2;
    std::vector<uint16_t> hs(nstream);
    std::vector<float> hx(128 * 64);
    srand(1);
    for (auto& v : hs) {       // bf16 weights ~ U(-0.03, 0.03)
        const float f = 0.06f * ((float)rand() / (float)RAND_MAX - 0.5f);
        uint32_t u;
        memcpy(&u, &f, 4);
        v = (uint16_t)(u >> 16);
    }
    for (auto& v : hx) { v = (float)rand() / (float)RAND_MAX - 0.3f; if (v < 0) v = 0; }
    char* d_stream;
    float *d_x, *d_out;
    CK(hipMalloc(&d_stream, nstream * 2));
    CK(hipMalloc(&d_x, hx.size() * 4));
    CK(hipMalloc(&d_out, (size_t)cus * 256 * 4));
    CK(hipMemcpy(d_stream, hs.data(), nstream * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    run<0>(d_stream, d_x, d_out, secs, cus);
    run<1>(d_stream, d_x, d_out, secs, cus);
    run<2>(d_stream, d_x, d_out, secs, cus);
    run<3>(d_stream, d_x, d_out, secs, cus);
    return 0;
}
