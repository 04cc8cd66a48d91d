// Micro-benchmark: does the ORDER in which a wave presents operands to v_mfma_f32_32x32x16_bf16 change what the power-limited
// MI355X sustains?  Register-only loop on random operands (as mfma_peak.hip); variants:
//   hold_b = H: the B fragment (activations) stays the same for H consecutive MFMAs (k_mlp_bf16 today: 2 -- a panel of two
//               output tiles per k-step), the A fragment changes every MFMA
//   hold_a = H: the A fragment (weights) stays for H consecutive MFMAs (what 64 samples per wave would give: 2)
//   which  = both | a_only | b_only: which operand carries random data (the other is zero)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_toggle.hip -o /tmp/mfma_toggle.out && /tmp/mfma_toggle.out [secs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kIters = 256, kUnroll = 64;

template <int HA, int HB>
__global__ void __launch_bounds__(512) k_mfma(const bf16x8* __restrict__ a_src, const bf16x8* __restrict__ b_src,
                                             float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 A[8], B[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = a_src[(size_t)i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) B[i] = b_src[(size_t)i * 64 + lane];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kUnroll; ++j)
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(j / HA) & 7], B[((j / HB) * 5) & 15], acc[j & 3], 0, 0, 0);
        if ((it & 15) == 15) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.5;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t nA = 8 * 64 * 8, nB = 16 * 64 * 8;
    std::vector<uint16_t> ha(nA), hb(nB);
    bf16x8 *dA, *dB;
    float* dOut;
    CK(hipMalloc(&dA, nA * 2));
    CK(hipMalloc(&dB, nB * 2));
    CK(hipMalloc(&dOut, (size_t)cus * 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V { const char* name; int which, ha, hb; };
    const V vs[] = {{"both random, A and B change every MFMA", 0, 1, 1}, {"both random, B held for 2 MFMAs (k_mlp_bf16 today)", 0, 1, 2},
                    {"both random, B held for 4", 0, 1, 4}, {"both random, B held for 8", 0, 1, 8},
                    {"both random, A held for 2, B every MFMA", 0, 2, 1}, {"both random, A held for 4", 0, 4, 1},
                    {"A random, B zero", 1, 1, 1}, {"A zero, B random", 2, 1, 1}, {"both zero", 3, 1, 1}};
    for (const V& v : vs) {
        srand(1);
        for (size_t i = 0; i < nA; ++i) ha[i] = (v.which == 0 || v.which == 1) ? f2bf(((float)rand() / RAND_MAX - 0.5f) * 0.2f) : 0;
        for (size_t i = 0; i < nB; ++i) {
            float g = 0;
            for (int k = 0; k < 12; ++k) g += (float)rand() / RAND_MAX;
            g -= 6.0f;
            hb[i] = (v.which == 0 || v.which == 2) ? f2bf(g > 0 ? g : 0.0f) : 0;
        }
        CK(hipMemcpy(dA, ha.data(), nA * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hb.data(), nB * 2, hipMemcpyHostToDevice));
        auto launch = [&]() {
            if (v.ha == 1 && v.hb == 1) hipLaunchKernelGGL((k_mfma<1, 1>), dim3(cus), dim3(512), 0, 0, dA, dB, dOut, kIters);
            else if (v.hb == 2) hipLaunchKernelGGL((k_mfma<1, 2>), dim3(cus), dim3(512), 0, 0, dA, dB, dOut, kIters);
            else if (v.hb == 4) hipLaunchKernelGGL((k_mfma<1, 4>), dim3(cus), dim3(512), 0, 0, dA, dB, dOut, kIters);
            else if (v.hb == 8) hipLaunchKernelGGL((k_mfma<1, 8>), dim3(cus), dim3(512), 0, 0, dA, dB, dOut, kIters);
            else if (v.ha == 2) hipLaunchKernelGGL((k_mfma<2, 1>), dim3(cus), dim3(512), 0, 0, dA, dB, dOut, kIters);
            else hipLaunchKernelGGL((k_mfma<4, 1>), dim3(cus), dim3(512), 0, 0, dA, dB, dOut, kIters);
        };
        const double flop = 2.0 * 32 * 32 * 16 * (double)kUnroll * kIters * 8 * cus;
        launch();
        CK(hipDeviceSynchronize());
        float ms = 0;
        double el = 0;
        while (el < secs * 0.5) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            el += ms * 1e-3;
        }
        double tot = 0;
        long n = 0;
        while (tot < secs * 500.0) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
            n += 20;
        }
        const double per = tot / n;
        printf("%-58s %7.1f TFLOP/s  %.3f of 2.5 PF  (2 waves/SIMD)\n", v.name, flop / (per * 1e-3) / 1e12, flop / (per * 1e-3) / 1e12 / 2500.0);
        fflush(stdout);
    }
    return 0;
}
