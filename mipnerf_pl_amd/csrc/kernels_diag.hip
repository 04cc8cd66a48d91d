// Diagnostics that ship inside the product library so that bench.py can report them from the SAME process as the
// headline (VERDICT r02 #2a): what a register-resident / LDS-fed v_mfma_f32_32x32x16_bf16 stream sustains on THIS
// chip, on operands shaped like the MLP's (weights ~ U(-0.1, 0.1), activations = relu(N(0,1))).
// The loop is the one of scripts/micro/mfma_peak.hip (round 2); no memory access inside it except, for lds = 1, one
// conflict-free ds_read_b128 A fragment per MFMA -- exactly the operand traffic of k_mlp_bf16.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kernels.hpp"

namespace mip {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kIters = 256;    // outer iterations per launch
constexpr int kUnroll = 64;    // MFMAs per iteration
constexpr unsigned kStreamChunks = 1216;   // one-KiB chunks of the shipped network's weight stream (mlp_plan.py): 1.19 MiB

template <int LDS>
__global__ void __launch_bounds__(512) k_mfma_ceiling(const bf16x8* __restrict__ a_src, const bf16x8* __restrict__ b_src,
                                                     float* __restrict__ out, int iters, const bf16x8* __restrict__ dma_src,
                                                     bf16x8* __restrict__ store_dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 A[8], B[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = a_src[(size_t)i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) B[i] = b_src[(size_t)i * 64 + lane];
    if (LDS) {
        for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) reinterpret_cast<bf16x8*>(smem)[i] = a_src[i];
        __syncthreads();
    }
    if (LDS >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const char* lane_base = smem + lane * 16;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: feeds SGPR asm operands
    for (int it = 0; it < iters; ++it) {
        const char* base = lane_base + (it & 1) * 32768;
        if (LDS >= 2) {
            // the weight stream of k_mlp_bf16: every wave DMAs 4 of every 32 one-KiB chunks its workgroup consumes (global_load_lds, L2 ->
            // LDS), i.e. 8 chunks per 64 MFMAs of its own; source = a 1.19-MiB stream that stays in the XCD's L2, destination = the half
            // of the ring the MFMAs are NOT reading in this iteration.  The previous iteration's DMAs are drained first (the kernel waits
            // for them before its ring barrier).
            if (LDS >= 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");      // the DMAs are done; last iteration's 7 stores may still fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned chunk0 = ((unsigned)(it * 8 + wave) * 8u) % (kStreamChunks - 8);
            const char* g = reinterpret_cast<const char*>(dma_src) + (size_t)chunk0 * 1024;
            char* l = smem + ((it + 1) & 1) * 32768 + (wave & 3) * 8192;
            const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)l;
            const unsigned lane16 = (unsigned)lane * 16u;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2\n\t"
                "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                "s_mov_b32 m0, %4\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %5\n\t"
                "global_load_lds_dwordx4 %1, %5 offset:1024\n\t"
                "global_load_lds_dwordx4 %1, %5 offset:2048\n\t"
                "global_load_lds_dwordx4 %1, %5 offset:3072\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(lane16), "s"(g), "s"(lds_addr), "s"(lds_addr + 4096u), "s"(g + 4096)
                : "memory");
        }
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            bf16x8 a;
            if (LDS >= 1) a = *reinterpret_cast<const bf16x8*>(base + (j & 31) * 1024);
            else a = A[j & 7];
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B[(j * 5) & 15], acc[j & 3], 0, 0, 0);
            if (LDS >= 3 && (j % 9) == 4) {
                // the saved-activation stream of k_mlp_bf16_trainfwd: 4,608 B per sample = one 1-KiB non-temporal store per wave every
                // 9.4 MFMAs (7 per 64), each to a fresh address (write-once, 3.7 GB per launch)
                // LDS == 3: every wave fills its own contiguous region (tile-major T-blocks, what the training kernels do);
                // LDS == 4: at every step the 2048 waves of the chip write ADJACENT 1-KiB chunks (block-major: [store index][wave]) --
                // a separate instantiation (MIPNERF_CEILING_STORE_PATTERN=1), so that neither pays for choosing at run time
                bf16x8* dst = LDS == 3
                    ? store_dst + ((((size_t)blockIdx.x * 8 + wave) * (size_t)iters + it) * 7 + j / 9) * 64 + lane
                    : store_dst + (((size_t)it * 7 + j / 9) * ((size_t)gridDim.x * 8) + (size_t)blockIdx.x * 8 + wave) * 64 + lane;
                __builtin_nontemporal_store(B[j & 15], dst);
            }
        }
        if ((it & 15) == 15) {      // keep the accumulators bounded (the MLP re-initialises them every 16-22 MFMAs)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
        }
    }
    if (LDS >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the fp32 matrix instruction of the parity mode (v_mfma_f32_32x32x2_f32: 4,096 flop, 64 cycles per SIMD), register-fed, back to back
__global__ void __launch_bounds__(512) k_mfma_ceiling_f32(const float* __restrict__ a_src, const float* __restrict__ b_src,
                                                         float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    float A[8], B[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = a_src[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) B[i] = b_src[i * 64 + lane];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kUnroll; ++j)
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j & 7], B[(j * 5) & 15], acc[j & 3], 0, 0, 0);
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

uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

// xorshift: the operand values must be the same on every box
struct Rng {
    uint64_t s;
    float uni() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (float)((s >> 11) & 0xFFFFFF) / 16777216.0f;
    }
};

// device buffers / events of a diagnostic run: released on every exit path (the DG macro returns early on a HIP error)
struct DevBufs {
    std::vector<void*> ptrs;
    std::vector<hipEvent_t> events;
    template <typename T>
    hipError_t alloc(T** p, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes);
        if (e == hipSuccess) ptrs.push_back(q);
        *p = static_cast<T*>(q);
        return e;
    }
    hipError_t event(hipEvent_t* ev) {
        const hipError_t e = hipEventCreate(ev);
        if (e == hipSuccess) events.push_back(*ev);
        return e;
    }
    ~DevBufs() {
        for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
        for (void* q : ptrs) (void)hipFree(q);
    }
};

}  // namespace

// Returns 0 on success.  tflops: algorithmic 2*32*32*16 flop per MFMA over the measured half of `seconds`;
// clock_ghz: the shader clock implied by the MFMA issue rate if the pipe never idles (32 cycles per MFMA per SIMD).
int run_mfma_ceiling(int lds_reads_per_mfma, int waves_per_simd, int random_operands, double seconds, double* tflops,
                     double* ms_per_launch, double* clock_ghz, hipStream_t st, char* msg, int msg_cap) {
#define DG(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { snprintf(msg, msg_cap, "%s: %s", #x, hipGetErrorString(e_)); return -1; } \
    } while (0)
    int dev = 0, cus = 0;
    DevBufs res;
    DG(hipGetDevice(&dev));
    DG(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t nA = 64 * 64 * 8, nB = 16 * 64 * 8;
    std::vector<uint16_t> ha(nA, 0), hb(nB, 0);
    if (random_operands) {
        Rng r{0x9E3779B97F4A7C15ull};
        for (size_t i = 0; i < nA; ++i) ha[i] = f2bf((r.uni() - 0.5f) * 0.2f);
        for (size_t i = 0; i < nB; ++i) {
            float g = 0;     // Irwin-Hall normal, then ReLU
            for (int k = 0; k < 12; ++k) g += r.uni();
            g -= 6.0f;
            hb[i] = f2bf(g > 0 ? g : 0.0f);
        }
    }
    bf16x8 *dA = nullptr, *dB = nullptr, *dS = nullptr;
    float* dOut = nullptr;
    std::vector<uint16_t> hs((size_t)kStreamChunks * 512, 0);
    if (random_operands) {
        Rng r2{0xD1B54A32D192ED03ull};
        for (auto& x : hs) x = f2bf((r2.uni() - 0.5f) * 0.2f);
    }
    DG(res.alloc(&dS, hs.size() * 2));
    DG(hipMemcpyAsync(dS, hs.data(), hs.size() * 2, hipMemcpyHostToDevice, st));
    DG(res.alloc(&dA, nA * 2));
    DG(res.alloc(&dB, nB * 2));
    DG(res.alloc(&dOut, (size_t)cus * 512 * 4));
    DG(hipMemcpyAsync(dA, ha.data(), nA * 2, hipMemcpyHostToDevice, st));
    DG(hipMemcpyAsync(dB, hb.data(), nB * 2, hipMemcpyHostToDevice, st));
    DG(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    DG(res.event(&e0));
    DG(res.event(&e1));
    const int threads = 256 * waves_per_simd;
    const bool fp32_mode = lds_reads_per_mfma == 10;
    float *fA = nullptr, *fB = nullptr;
    if (fp32_mode) {
        std::vector<float> ha32(8 * 64, 0.0f), hb32(16 * 64, 0.0f);
        if (random_operands) {
            Rng r3{0xA0761D6478BD642Full};
            for (auto& x : ha32) x = (r3.uni() - 0.5f) * 0.2f;
            for (auto& x : hb32) {
                float g = 0;
                for (int k = 0; k < 12; ++k) g += r3.uni();
                g -= 6.0f;
                x = g > 0 ? g : 0.0f;
            }
        }
        DG(res.alloc(&fA, ha32.size() * 4));
        DG(res.alloc(&fB, hb32.size() * 4));
        DG(hipMemcpyAsync(fA, ha32.data(), ha32.size() * 4, hipMemcpyHostToDevice, st));
        DG(hipMemcpyAsync(fB, hb32.data(), hb32.size() * 4, hipMemcpyHostToDevice, st));
        DG(hipStreamSynchronize(st));
    }
    DG(hipFuncSetAttribute((const void*)k_mfma_ceiling<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    DG(hipFuncSetAttribute((const void*)k_mfma_ceiling<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    DG(hipFuncSetAttribute((const void*)k_mfma_ceiling<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    DG(hipFuncSetAttribute((const void*)k_mfma_ceiling<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    bf16x8* dStore = nullptr;
    const char* pat_env = getenv("MIPNERF_CEILING_STORE_PATTERN");      // experiment knob of feeding mode 3 (see the kernel)
    const int store_pattern = pat_env ? atoi(pat_env) : 0;
    if (lds_reads_per_mfma == 3) DG(res.alloc(&dStore, (size_t)cus * 8 * kIters * 7 * 1024));       // 3.7 GB: every store address is written once per launch
    auto launch = [&]() {
        if (fp32_mode) { hipLaunchKernelGGL(k_mfma_ceiling_f32, dim3(cus), dim3(threads), 0, st, fA, fB, dOut, kIters); return; }
        if (lds_reads_per_mfma == 3 && store_pattern == 1) hipLaunchKernelGGL(k_mfma_ceiling<4>, dim3(cus), dim3(threads), 65536, st, dA, dB, dOut, kIters, dS, dStore);
        else if (lds_reads_per_mfma == 3) hipLaunchKernelGGL(k_mfma_ceiling<3>, dim3(cus), dim3(threads), 65536, st, dA, dB, dOut, kIters, dS, dStore);
        else if (lds_reads_per_mfma == 2) hipLaunchKernelGGL(k_mfma_ceiling<2>, dim3(cus), dim3(threads), 65536, st, dA, dB, dOut, kIters, dS, dStore);
        else if (lds_reads_per_mfma) hipLaunchKernelGGL(k_mfma_ceiling<1>, dim3(cus), dim3(threads), 65536, st, dA, dB, dOut, kIters, dS, dStore);
        else hipLaunchKernelGGL(k_mfma_ceiling<0>, dim3(cus), dim3(threads), 0, st, dA, dB, dOut, kIters, dS, dStore);
    };
    const double flop = 2.0 * 32 * 32 * (fp32_mode ? 2 : 16) * (double)kUnroll * kIters * (threads / 64) * cus;
    launch();
    DG(hipStreamSynchronize(st));
    const int batch = 20;
    float ms = 0;
    double elapsed = 0;
    while (elapsed < seconds * 0.5) {         // heat-up half: the package settles on its power budget
        DG(hipEventRecord(e0, st));
        for (int i = 0; i < batch; ++i) launch();
        DG(hipEventRecord(e1, st));
        DG(hipEventSynchronize(e1));
        DG(hipEventElapsedTime(&ms, e0, e1));
        elapsed += ms * 1e-3;
    }
    double tot_ms = 0;
    long launches = 0;
    while (tot_ms < seconds * 500.0) {        // measured half
        DG(hipEventRecord(e0, st));
        for (int i = 0; i < batch; ++i) launch();
        DG(hipEventRecord(e1, st));
        DG(hipEventSynchronize(e1));
        DG(hipEventElapsedTime(&ms, e0, e1));
        tot_ms += ms;
        launches += batch;
    }
    const double per = tot_ms / launches;
    *ms_per_launch = per;
    *tflops = flop / (per * 1e-3) / 1e12;
    *clock_ghz = (double)kUnroll * kIters * (fp32_mode ? 64.0 : 32.0) * waves_per_simd / (per * 1e-3) / 1e9;
    snprintf(msg, msg_cap, "ok");
    return 0;
#undef DG
}


// ======================================================================================================================
// CU -> CU hand-off probe (milestone 1): can sample tiles stream from one workgroup to another through
// L2 / Infinity Cache fast enough for a layer-pipelined training backward (each CU group owns ONE layer; delta tiles of
// 512 B/sample go group -> group instead of through HBM)?  128 producer workgroups each hand `tiles` tiles of
// `tile_bytes` to one consumer workgroup through a ring of `ring` slots in global memory:
//   producer: wait slot free (consumed counter) -> 16-B stores of a checkable pattern -> publish (ready counter)
//   consumer: poll ready (one lane, relaxed sc1 load + s_sleep) -> agent acquire -> barrier -> 16-B loads, every word
//             verified -> consumed counter
// Placement: block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md); same_xcd = 1 pairs b with b + 8 (same XCD),
// 0 pairs 2p with 2p + 1 (neighbouring XCDs).  store flavour 0: plain stores + __syncthreads + lane-0 agent release fence
// + vmcnt(0) + relaxed agent flag; 1: write-through (sc1) 16-B stores, every wave drains vmcnt, barrier, sc1 flag.
// `mfma_per_wave` register-only MFMAs per wave and tile on BOTH sides stand for the layer's arithmetic (a 256-sample tile of
// one 256x256 layer's dgrad + wgrad = 256 MFMAs per wave).  Every poll loop is bounded: on a timeout the launch sets an
// abort word, every loop leaves, and the probe reports it -- a missing co-resident partner cannot hang the GPU.
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kProbeThreads = 512;
constexpr unsigned kMaxSpins = 1u << 21;

__device__ __forceinline__ uint4 probe_pattern(unsigned pair, unsigned tile, unsigned idx) {
    unsigned h = pair * 0x9E3779B1u ^ tile * 0x85EBCA77u ^ idx * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return make_uint4(h, h ^ 0xA5A5A5A5u, h + idx, h - tile);
}

__device__ __forceinline__ unsigned load_flag(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load_dword sc1: L2-served
}

__device__ __forceinline__ void store_sc1_x4(uint4* p, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    // s_nop 1: a VMEM store of more than 8 bytes followed by a VALU write of its data VGPRs needs TWO wait states on gfx940+ (one on
    // older gfx9); the compiler's hazard recogniser does not look inside asm and the next pattern computation reuses the registers at
    // once.  Measured: with no / one wait state 1-3 % of the words arrive corrupted in the unrolled per-wave kernels (r03w), with two: 0.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}

// 16-byte load that bypasses this CU's L1 (served by the XCD's L2): flavour 2 reads a same-XCD producer's plain stores with it
__device__ __forceinline__ void load_sc1_x4_issue(const uint4* p, u32x4& dst) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}

// lane 0 waits until *flag >= need (or abort); result broadcast through LDS.  Returns false on abort.
__device__ __forceinline__ bool wait_counter(const unsigned* flag, unsigned need, unsigned* abort_word, int* lds_ok,
                                             unsigned long long& stall_ticks) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned spins = 0;
        int ok = 1;
        while (load_flag(flag) < need) {
            if (++spins > kMaxSpins || load_flag(abort_word) != 0) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        stall_ticks += __builtin_amdgcn_s_memtime() - t0;
        *lds_ok = ok;
    }
    __syncthreads();
    const bool ok = *lds_ok != 0;
    __syncthreads();
    return ok;
}

template <int FLAVOUR>
__global__ void __launch_bounds__(kProbeThreads) k_handoff_probe(uint4* __restrict__ ring_mem, unsigned* __restrict__ ready,
                                                                unsigned* __restrict__ consumed, unsigned* __restrict__ abort_word,
                                                                unsigned long long* __restrict__ stall, unsigned* __restrict__ errors,
                                                                int same_xcd, int tiles, int ring, int tile_vec, int mfma_per_wave,
                                                                const bf16x8* __restrict__ ab_src, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // only to force one workgroup per CU
    __shared__ int lds_ok;
    const int b = blockIdx.x;
    int pair, is_consumer;
    if (same_xcd) {
        const int row = b >> 3, x = b & 7;          // row 0..31 within the XCD
        pair = (row >> 1) * 8 + x;
        is_consumer = row & 1;
    } else {
        pair = b >> 1;
        is_consumer = b & 1;
    }
    uint4* slots = ring_mem + (size_t)pair * ring * tile_vec;
    unsigned long long stall_ticks = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    unsigned bad = 0;
    const int lane = threadIdx.x & 63;
    bf16x8 A0 = ab_src[lane], B0 = ab_src[64 + lane];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    const int per_thread = tile_vec / kProbeThreads;     // 16-B vectors per thread and tile (multiple of 8)
    for (int t = 0; t < tiles; ++t) {
        uint4* slot = slots + (size_t)(t % ring) * tile_vec;
        if (!is_consumer) {
            if (t >= ring && !wait_counter(&consumed[pair], (unsigned)(t - ring + 1), abort_word, &lds_ok, stall_ticks)) break;
            for (int i = 0; i < per_thread; ++i) {
                const unsigned idx = (unsigned)(i * kProbeThreads + threadIdx.x);
                const uint4 v = probe_pattern((unsigned)pair, (unsigned)t, idx);
                if (FLAVOUR == 1) store_sc1_x4(slot + idx, v);
                else slot[idx] = v;
            }
            for (int m = 0; m < mfma_per_wave; m += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc1, 0, 0, 0);
            }
            if (FLAVOUR == 1 || FLAVOUR == 2) {
                // 2: plain stores are write-through to the XCD's L2; a same-XCD consumer that bypasses ITS L1 needs no fence at all
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(&ready[pair], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __syncthreads();
                if (threadIdx.x == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&ready[pair], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            if (!wait_counter(&ready[pair], (unsigned)(t + 1), abort_word, &lds_ok, stall_ticks)) break;
            if (FLAVOUR != 2 && threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            for (int i0 = 0; i0 < per_thread; i0 += 8) {
                uint4 v[8];
                u32x4 r[8];
                if (FLAVOUR == 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) load_sc1_x4_issue(slot + (size_t)(i0 + i) * kProbeThreads + threadIdx.x, r[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = slot[(size_t)(i0 + i) * kProbeThreads + threadIdx.x];
                }
                if (i0 == 0)
                    for (int m = 0; m < mfma_per_wave; m += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc1, 0, 0, 0);
                    }
                if (FLAVOUR == 2) {       // the asm loads are invisible to the compiler's waitcnt insertion: wait by hand, after the MFMAs
                    // the registers are operands of the wait, so the compiler cannot read them before it
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                                 :: "memory");
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = make_uint4(r[i].x, r[i].y, r[i].z, r[i].w);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint4 w = probe_pattern((unsigned)pair, (unsigned)t, (unsigned)((i0 + i) * kProbeThreads + threadIdx.x));
                    bad += (v[i].x != w.x) | (v[i].y != w.y) | (v[i].z != w.z) | (v[i].w != w.w);
                }
            }
            __syncthreads();      // every lane's loads have returned (they were compared)
            if (threadIdx.x == 0) __hip_atomic_store(&consumed[pair], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) sink[b] = s;      // keeps the filler MFMAs alive
    if (bad) atomicAdd(errors, bad);
    if (threadIdx.x == 0) {          // both in s_memtime ticks of the SAME run: their ratio needs no clock assumption
        stall[b] = stall_ticks;
        stall[256 + b] = __builtin_amdgcn_s_memtime() - t_begin;
    }
    (void)smem;
}

// ---- probe v2: per-WAVE streams ---------------------------------------------------------------------------------------------
// The workgroup-wide protocol above serialises flag -> loads -> compare per tile on the consumer (one round of loads in flight,
// two barriers): its rate is partly a latency figure.  Here wave w of the producer workgroup streams sub-tiles (tile_bytes / 8) to
// wave w of the consumer workgroup through its own ring and flags: no workgroup barrier anywhere, a whole sub-tile (PL 16-byte
// vectors per lane) in flight per wave, and the eight waves of a CU drift apart so that one wave's flag / load latency overlaps the
// others' transfers -- the most a CU pair can hand over with ordinary loads and stores.
//   FLAVOUR 3: write-through (sc1) stores, vmcnt(0), flag; consumer: flag, agent-scope acquire (per wave), plain loads.  Any placement.
//   FLAVOUR 4: plain stores (land in the XCD's L2), vmcnt(0), flag; consumer: flag, sc1 loads (bypass its L1), no fence.  Same XCD only.
template <int FLAVOUR, int PL>
__global__ void __launch_bounds__(kProbeThreads) k_handoff_waves(uint4* __restrict__ ring_mem, unsigned* __restrict__ ready,
                                                                unsigned* __restrict__ consumed, unsigned* __restrict__ abort_word,
                                                                unsigned* __restrict__ errors, int same_xcd, int tiles, int ring,
                                                                int mfma_per_wave, const bf16x8* __restrict__ ab_src,
                                                                float* __restrict__ sink, int active_pairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // only to force one workgroup per CU
    const int b = blockIdx.x;
    int pair, is_consumer;
    if (same_xcd) {
        const int row = b >> 3, x = b & 7;
        pair = (row >> 1) * 8 + x;
        is_consumer = row & 1;
    } else {
        pair = b >> 1;
        is_consumer = b & 1;
    }
    if (pair >= active_pairs) return;             // MIPNERF_PROBE_PAIRS: is the rate a per-pair or a chip-wide limit?
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stream = pair * 8 + wave;
    constexpr int kSub = PL * 64;                                    // 16-byte vectors per sub-tile
    uint4* slots = ring_mem + (size_t)stream * ring * kSub;
    bf16x8 A0 = ab_src[lane], B0 = ab_src[64 + lane];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    unsigned bad = 0;
    bool alive = true;
    auto wait_for = [&](const unsigned* flag, unsigned need) {
        unsigned spins = 0;
        while (load_flag(flag) < need) {
            if (++spins > kMaxSpins || load_flag(abort_word) != 0) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        return true;
    };
    for (int t = 0; t < tiles && alive; ++t) {
        uint4* slot = slots + (size_t)(t % ring) * kSub;
        if (!is_consumer) {
            if (t >= ring) alive = wait_for(&consumed[stream], (unsigned)(t - ring + 1));
            if (!alive) break;
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                const unsigned idx = (unsigned)(i * 64 + lane);
                const uint4 v = probe_pattern((unsigned)stream, (unsigned)t, idx);
                if (FLAVOUR == 3) store_sc1_x4(slot + idx, v);
                else slot[idx] = v;
            }
            for (int m = 0; m < mfma_per_wave; m += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc1, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores are acknowledged (L2 / memory side)
            if (lane == 0) __hip_atomic_store(&ready[stream], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            alive = wait_for(&ready[stream], (unsigned)(t + 1));
            if (!alive) break;
            uint4 v[PL];
            if (FLAVOUR == 3) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
                for (int i = 0; i < PL; ++i) v[i] = slot[i * 64 + lane];
                for (int m = 0; m < mfma_per_wave; m += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc1, 0, 0, 0);
                }
            } else {
                u32x4 r[PL];
#pragma unroll
                for (int i = 0; i < PL; ++i) load_sc1_x4_issue(slot + i * 64 + lane, r[i]);
                for (int m = 0; m < mfma_per_wave; m += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc1, 0, 0, 0);
                }
                // the asm loads are invisible to the compiler's waitcnt insertion; the registers are operands of the waits (in issue
                // order, 8 per wait), so nothing reads them early
#pragma unroll
                for (int g = 0; g < PL / 8; ++g) {
                    u32x4* q = r + g * 8;
                    if (PL / 8 - 1 - g == 3) asm volatile("s_waitcnt vmcnt(24)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) :: "memory");
                    else if (PL / 8 - 1 - g == 2) asm volatile("s_waitcnt vmcnt(16)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) :: "memory");
                    else if (PL / 8 - 1 - g == 1) asm volatile("s_waitcnt vmcnt(8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) :: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) :: "memory");
                }
#pragma unroll
                for (int i = 0; i < PL; ++i) v[i] = make_uint4(r[i].x, r[i].y, r[i].z, r[i].w);
            }
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                const uint4 w = probe_pattern((unsigned)stream, (unsigned)t, (unsigned)(i * 64 + lane));
                bad += (v[i].x != w.x) | (v[i].y != w.y) | (v[i].z != w.z) | (v[i].w != w.w);
            }
            // every lane's data has arrived (it was compared): the slot may be overwritten
            if (__builtin_amdgcn_readfirstlane((int)bad) >= 0 && lane == 0)
                __hip_atomic_store(&consumed[stream], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) sink[b] = s;
    if (bad) atomicAdd(errors, bad);
    (void)smem;
}

}  // namespace

// out[0] = aggregate GB/s handed off (payload bytes / kernel time), out[1] = ms, out[2] = mean producer stall fraction,
// out[3] = mean consumer stall fraction, out[4] = mismatching 16-B words, out[5] = 1 if a poll timed out.
int run_handoff_probe(int same_xcd, int flavour, int tiles, int ring, int tile_bytes, int mfma_per_wave, int reps, double* out,
                      hipStream_t st, char* msg, int msg_cap) {
#define DG(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { snprintf(msg, msg_cap, "%s: %s", #x, hipGetErrorString(e_)); return -1; } \
    } while (0)
    int dev = 0, cus = 0;
    DevBufs res;
    DG(hipGetDevice(&dev));
    DG(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus != 256) { snprintf(msg, msg_cap, "probe assumes 256 CUs in 8 XCDs, device has %d", cus); return -2; }
    if (tile_bytes % (16 * kProbeThreads * 8) || ring < 1 || tiles < 1 || reps < 1) { snprintf(msg, msg_cap, "bad probe arguments"); return -2; }
    if (flavour < 0 || flavour > 4 || (flavour >= 3 && tile_bytes != 65536 && tile_bytes != 131072) ||
        ((flavour == 2 || flavour == 4) && !same_xcd)) {
        snprintf(msg, msg_cap, "bad probe flavour (0-4; 3 / 4 take 64 / 128 KiB tiles; 2 / 4 are same-XCD protocols)");
        return -2;
    }
    const int pairs = 128, tile_vec = tile_bytes / 16;
    int active = pairs;                           // per-wave protocols only: pairs 0 .. active-1 run (same_xcd: 8 per XCD row)
    if (const char* e = getenv("MIPNERF_PROBE_PAIRS")) {
        active = atoi(e);
        if (active < 1 || active > pairs || flavour < 3) { snprintf(msg, msg_cap, "MIPNERF_PROBE_PAIRS must be 1..128 and needs flavour 3 / 4"); return -2; }
    }
    uint4* ring_mem = nullptr;
    unsigned *ready = nullptr, *consumed = nullptr, *abort_word = nullptr, *errors = nullptr;
    unsigned long long* stall = nullptr;
    bf16x8* ab = nullptr;
    float* sink = nullptr;
    DG(res.alloc(&ring_mem, (size_t)pairs * ring * tile_bytes));
    DG(res.alloc(&ready, pairs * 8 * 4));          // probe v2: one counter per wave stream
    DG(res.alloc(&consumed, pairs * 8 * 4));
    DG(res.alloc(&abort_word, 4));
    DG(res.alloc(&errors, 4));
    DG(res.alloc(&stall, 512 * 8));
    DG(res.alloc(&ab, 128 * 16));
    DG(res.alloc(&sink, 256 * 4));
    std::vector<uint16_t> hab(128 * 8);
    Rng r{0x1234567ull};
    for (auto& x : hab) x = f2bf(r.uni() - 0.3f);
    DG(hipMemcpyAsync(ab, hab.data(), 128 * 16, hipMemcpyHostToDevice, st));
    DG(hipMemsetAsync(errors, 0, 4, st));
    DG(hipMemsetAsync(abort_word, 0, 4, st));
    const size_t lds = 96 * 1024;                 // > half of the CU's 160 KiB: one workgroup per CU
    DG(hipFuncSetAttribute((const void*)k_handoff_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DG(hipFuncSetAttribute((const void*)k_handoff_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DG(hipFuncSetAttribute((const void*)k_handoff_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DG(hipFuncSetAttribute((const void*)k_handoff_waves<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DG(hipFuncSetAttribute((const void*)k_handoff_waves<3, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DG(hipFuncSetAttribute((const void*)k_handoff_waves<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DG(hipFuncSetAttribute((const void*)k_handoff_waves<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    DG(res.event(&e0));
    DG(res.event(&e1));
    double best_ms = 1e30;
    for (int rep = 0; rep < reps + 1; ++rep) {    // first repetition = warm-up
        DG(hipMemsetAsync(ready, 0, pairs * 8 * 4, st));
        DG(hipMemsetAsync(consumed, 0, pairs * 8 * 4, st));
        DG(hipEventRecord(e0, st));
        if (flavour >= 3) {
#define WAVES(F, PLV)                                                                                                         \
    hipLaunchKernelGGL((k_handoff_waves<F, PLV>), dim3(256), dim3(kProbeThreads), lds, st, ring_mem, ready, consumed, abort_word, \
                       errors, same_xcd, tiles, ring, mfma_per_wave, ab, sink, active)
            const int pl = tile_bytes / (16 * kProbeThreads);
            if (flavour == 3) { if (pl == 8) WAVES(3, 8); else WAVES(3, 16); }
            else { if (pl == 8) WAVES(4, 8); else WAVES(4, 16); }
#undef WAVES
        } else if (flavour == 2) hipLaunchKernelGGL(k_handoff_probe<2>, dim3(256), dim3(kProbeThreads), lds, st, ring_mem, ready, consumed, abort_word,
                                             stall, errors, same_xcd, tiles, ring, tile_vec, mfma_per_wave, ab, sink);
        else if (flavour) hipLaunchKernelGGL(k_handoff_probe<1>, dim3(256), dim3(kProbeThreads), lds, st, ring_mem, ready, consumed, abort_word,
                                        stall, errors, same_xcd, tiles, ring, tile_vec, mfma_per_wave, ab, sink);
        else hipLaunchKernelGGL(k_handoff_probe<0>, dim3(256), dim3(kProbeThreads), lds, st, ring_mem, ready, consumed, abort_word, stall,
                                errors, same_xcd, tiles, ring, tile_vec, mfma_per_wave, ab, sink);
        DG(hipEventRecord(e1, st));
        DG(hipEventSynchronize(e1));
        float ms = 0;
        DG(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best_ms) best_ms = ms;
    }
    unsigned h_err = 0, h_abort = 0;
    std::vector<unsigned long long> h_stall(512);
    DG(hipMemcpy(&h_err, errors, 4, hipMemcpyDeviceToHost));
    DG(hipMemcpy(&h_abort, abort_word, 4, hipMemcpyDeviceToHost));
    DG(hipMemcpy(h_stall.data(), stall, 512 * 8, hipMemcpyDeviceToHost));
    double sp = 0, sc = 0;
    for (int b = 0; b < 256; ++b) {      // fraction of the workgroup's own lifetime (s_memtime ticks) its lane 0 spent polling
        const int cons = same_xcd ? ((b >> 3) & 1) : (b & 1);
        const double f = h_stall[256 + b] ? (double)h_stall[b] / (double)h_stall[256 + b] : 0.0;
        (cons ? sc : sp) += f;
    }
    out[0] = (double)active * tiles * tile_bytes / (best_ms * 1e-3) / 1e9;
    out[1] = best_ms;
    out[2] = sp / 128;
    out[3] = sc / 128;
    out[4] = (double)h_err;
    out[5] = (double)h_abort;
    snprintf(msg, msg_cap, "ok");
    return 0;
#undef DG
}

}  // namespace mip
