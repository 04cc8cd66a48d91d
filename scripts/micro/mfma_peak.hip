// Micro-benchmark: what does v_mfma_f32_32x32x16_bf16 SUSTAIN on this MI355X, and what do operand values and the
// LDS operand traffic of k_mlp_bf16 do to it?  
//
// Register-only loop: every wave holds 8 A fragments + 16 B fragments + 4 accumulators in VGPRs and issues
// back-to-back MFMAs on rotating accumulators (no memory access in the loop).  Variants:
//   data  = zero | random (A ~ U(-0.1,0.1) "weights", B = relu(N(0,1)) "activations", like the MLP's operands)
//   waves = 1 or 2 per SIMD (256- or 512-thread workgroups, one per CU)
//   lds   = 0: no LDS;  1: one ds_read_b128 A fragment per MFMA (what k_mlp_bf16 does);  2: one per TWO MFMAs
//           (64 samples per wave);  the LDS ring holds 64 random chunks.
// Each variant runs back-to-back launches for >= `secs` seconds (DVFS settles within milliseconds; the package warms
// over seconds) and reports TFLOP/s over the LAST half of that time and the effective shader clock
// (s_memtime ticks / wall time of one launch).
//
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_peak.hip -o scripts/micro/mfma_peak.out && scripts/micro/mfma_peak.out [secs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kIters = 256;       // outer iterations per launch
constexpr int kUnroll = 64;       // MFMAs per iteration

template <int LDS>
__global__ void __launch_bounds__(512) k_mfma(const bf16x8* __restrict__ a_src, const bf16x8* __restrict__ b_src,
                                             float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 A[8], B[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = a_src[(size_t)i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) B[i] = b_src[(size_t)i * 64 + lane];
    if (LDS) {
        for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x)
            reinterpret_cast<bf16x8*>(smem)[i] = a_src[i % (8 * 64)];
        __syncthreads();
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const char* lane_base = smem + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const char* base = lane_base + (it & 1) * 32768;
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            bf16x8 a;
            if (LDS == 1) a = *reinterpret_cast<const bf16x8*>(base + (j & 31) * 1024);
            else if (LDS == 2) a = *reinterpret_cast<const bf16x8*>(base + ((j >> 1) & 31) * 1024);
            else a = A[j & 7];
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B[(j * 5) & 15], acc[j & 3], 0, 0, 0);
        }
        if ((it & 15) == 15) {      // keep the accumulators bounded (the MLP re-initialises them every 16-22 MFMAs)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t nA = 64 * 64 * 8, nB = 16 * 64 * 8;
    std::vector<uint16_t> ha(nA), hb(nB);
    bf16x8 *dA, *dB;
    float* dOut;
    unsigned long long* dT;
    CK(hipMalloc(&dA, nA * 2));
    CK(hipMalloc(&dB, nB * 2));
    CK(hipMalloc(&dOut, (size_t)cus * 512 * 4));
    CK(hipMalloc(&dT, (size_t)cus * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"device_cus\": %d, \"seconds_per_variant\": %.1f, \"variants\": [\n", cus, secs);
    bool first = true;
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (size_t i = 0; i < nA; ++i) ha[i] = data ? f2bf(((float)rand() / RAND_MAX - 0.5f) * 0.2f) : 0;
        for (size_t i = 0; i < nB; ++i) {
            // relu(N(0,1)) via a cheap Irwin-Hall normal
            float g = 0;
            for (int k = 0; k < 12; ++k) g += (float)rand() / RAND_MAX;
            g -= 6.0f;
            hb[i] = data ? f2bf(g > 0 ? g : 0.0f) : 0;
        }
        CK(hipMemcpy(dA, ha.data(), nA * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hb.data(), nB * 2, hipMemcpyHostToDevice));
        for (int waves = 1; waves <= 2; ++waves)
            for (int lds = 0; lds < 3; ++lds) {
                const int threads = 256 * waves;
                auto launch = [&]() {
                    if (lds == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(cus), dim3(threads), 65536, 0, dA, dB, dOut, dT, kIters);
                    else if (lds == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(cus), dim3(threads), 65536, 0, dA, dB, dOut, dT, kIters);
                    else hipLaunchKernelGGL(k_mfma<2>, dim3(cus), dim3(threads), 65536, 0, dA, dB, dOut, dT, kIters);
                };
                if (lds) {
                    CK(hipFuncSetAttribute((const void*)k_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
                    CK(hipFuncSetAttribute((const void*)k_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
                }
                const double flop = 2.0 * 32 * 32 * 16 * (double)kUnroll * kIters * (threads / 64) * cus;
                launch();
                CK(hipDeviceSynchronize());
                // heat-up half
                double elapsed = 0;
                float ms = 0;
                int batch = 20;
                while (elapsed < secs * 0.5) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < batch; ++i) launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    elapsed += ms * 1e-3;
                }
                // measured half
                double tot_ms = 0;
                long launches = 0;
                double first_ms = -1, last_ms = 0;
                while (tot_ms < secs * 500.0) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < batch; ++i) launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (first_ms < 0) first_ms = ms / batch;
                    last_ms = ms / batch;
                    tot_ms += ms;
                    launches += batch;
                }
                std::vector<unsigned long long> ht(cus);
                CK(hipMemcpy(ht.data(), dT, (size_t)cus * 8, hipMemcpyDeviceToHost));
                double tick = 0;
                for (int i = 0; i < cus; ++i) tick += (double)ht[i];
                tick /= cus;
                const double per_launch_ms = tot_ms / launches;
                // s_memtime on gfx950 counts at a constant 100 MHz; the SHADER clock follows from the MFMA issue rate:
                // kUnroll*kIters MFMAs per wave x 32 cycles (x waves per SIMD) is the minimum cycle count of the loop
                const double min_cycles = (double)kUnroll * kIters * 32.0 * waves;
                printf("%s  {\"data\": \"%s\", \"waves_per_simd\": %d, \"lds_a_reads_per_mfma\": %s, \"tflops\": %.1f, \"frac_of_2500\": %.3f, "
                       "\"ms_per_launch\": %.4f, \"first_batch_ms\": %.4f, \"last_batch_ms\": %.4f, \"memtime_ticks\": %.0f, "
                       "\"mfma_bound_clock_ghz_if_pipe_full\": %.3f}",
                       first ? "" : ",\n", data ? "random" : "zero", waves, lds == 0 ? "0" : (lds == 1 ? "1" : "0.5"),
                       flop / (per_launch_ms * 1e-3) / 1e12, flop / (per_launch_ms * 1e-3) / 1e12 / 2500.0, per_launch_ms, first_ms, last_ms,
                       tick, min_cycles / (per_launch_ms * 1e-3) / 1e9);
                first = false;
                fflush(stdout);
            }
    }
    printf("\n]}\n");
    return 0;
}
