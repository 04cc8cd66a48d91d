// Hardware self-test of the three gfx950 behaviours every MFMA kernel in this library is
// built on: (1) operand / result lane layout of v_mfma_f32_32x32x16_bf16, (2) of
// v_mfma_f32_32x32x2_f32, (3) global_load_lds_dwordx4 landing lane-linear at a wave-uniform
// LDS base.  Run once on the GPU box (tests + smoke) so a layout assumption that is wrong
// fails loudly instead of silently transposing a layer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "kernels.hpp"

namespace mip {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// A [32][16] row-major, B [16][32] row-major (asymmetric), D [32][32]
__global__ void k_selftest_bf16(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[m * 16 + hi * 8 + j];     // A operand: row = lane&31, k = hi*8+j
        b[j] = (__bf16)B[(hi * 8 + j) * 32 + m];   // B operand: col = lane&31, k = hi*8+j
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + m] = acc[r];
}

// A [32][2], B [2][32]
__global__ void k_selftest_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * 2 + hi], B[hi * 32 + m], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + m] = acc[r];
}

// 4 waves; wave w DMAs 4 x 1 KiB (one base, immediate offsets 0/1024/2048/3072 applied to both the
// global and the LDS side -- exactly the form issue_group() uses) to LDS base 2048 + w*4096, then
// every chunk is read back by ANOTHER wave.
__global__ void k_selftest_dma(const float* __restrict__ src, float* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned lane16 = lane * 16u;
    const char* gbase = reinterpret_cast<const char*>(src) + (size_t)wave * 4096;
    asm volatile("" : "+v"(lane16));
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(gbase + lane16);
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(smem + 2048 + wave * 4096);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int c = 0; c < 16; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(smem + 2048 + c * 1024 + lane * 16);
        if (wave == (((c >> 2) + 1) & 3)) reinterpret_cast<float4*>(dst)[c * 64 + lane] = v;
    }
}

static float bf16r(float x) {   // host RNE to bf16
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&x, &u, 4);
    return x;
}

int run_selftest(hipStream_t st, char* msg, int cap) {
    int bad = 0;
    float *dA = nullptr, *dB = nullptr, *dD = nullptr;
    hipError_t em = hipMalloc(&dA, 16384);
    if (em == hipSuccess) em = hipMalloc(&dB, 4096);
    if (em == hipSuccess) em = hipMalloc(&dD, 16384);
    if (em != hipSuccess) {
        snprintf(msg, cap, "selftest: hipMalloc failed: %s", hipGetErrorString(em));
        return -1;
    }
    std::vector<float> A(4096), B(1024), D(4096);
    auto rnd = [](int i) { return (float)((i * 7919 + 13) % 23 - 11) * 0.125f; };   // exact in bf16
    // ---- bf16 32x32x16
    for (int i = 0; i < 512; ++i) { A[i] = rnd(i); B[i] = rnd(3 * i + 5); }
    hipMemcpyAsync(dA, A.data(), 2048, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(dB, B.data(), 2048, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_selftest_bf16, dim3(1), dim3(64), 0, st, dA, dB, dD);
    hipMemcpyAsync(D.data(), dD, 4096, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    int nb = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float r = 0;
            for (int k = 0; k < 16; ++k) r += bf16r(A[i * 16 + k]) * bf16r(B[k * 32 + j]);
            if (fabsf(r - D[i * 32 + j]) > 1e-4f) ++nb;
        }
    if (nb) bad |= 1;
    // ---- f32 32x32x2
    for (int i = 0; i < 64; ++i) { A[i] = rnd(i) + 0.001f * i; B[i] = rnd(2 * i + 1) - 0.003f * i; }
    hipMemcpyAsync(dA, A.data(), 256, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(dB, B.data(), 256, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_selftest_f32, dim3(1), dim3(64), 0, st, dA, dB, dD);
    hipMemcpyAsync(D.data(), dD, 4096, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    int nf = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            const float r = fmaf(A[i * 2 + 1], B[32 + j], A[i * 2] * B[j]);
            if (fabsf(r - D[i * 32 + j]) > 1e-5f) ++nf;
        }
    if (nf) bad |= 2;
    // ---- LDS DMA
    for (int i = 0; i < 4096; ++i) A[i] = (float)i + 0.5f;
    hipMemcpyAsync(dA, A.data(), 16384, hipMemcpyHostToDevice, st);
    hipMemsetAsync(dD, 0, 16384, st);
    hipLaunchKernelGGL(k_selftest_dma, dim3(1), dim3(256), 2048 + 16384, st, dA, dD);
    hipMemcpyAsync(D.data(), dD, 16384, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    int nd = 0;
    for (int i = 0; i < 4096; ++i)
        if (D[i] != A[i]) ++nd;
    if (nd) bad |= 4;
    const hipError_t er = hipGetLastError();
    snprintf(msg, cap, "selftest: bf16_mfma_mismatch=%d f32_mfma_mismatch=%d lds_dma_mismatch=%d hip=%s", nb, nf, nd,
             hipGetErrorString(er));
    hipFree(dA); hipFree(dB); hipFree(dD);
    if (er != hipSuccess) return -1;
    return bad;
}

}  // namespace mip
