// Packs the fp32 master parameters (separate torch tensors) into the MFMA operand streams:
// out[i] = table[i] < 0 ? 0 : params[table[i] >> 20][table[i] & 0xFFFFF], as bf16 (RNE) or fp32.
// Runs once per optimizer step / load_state_dict (610k elements: negligible).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mip {

template <typename OutT>
__global__ void __launch_bounds__(256)
k_pack(const int32_t* __restrict__ table, int64_t n, const ParamPtrs ptrs, OutT* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t e = table[i];
    float v = 0.0f;
    if (e >= 0) v = ptrs.p[e >> 20][e & 0xFFFFF];
    out[i] = (OutT)v;
}

// All operand streams of a context in ONE launch (mipnerf_set_params runs inside the captured training step: five launches of
// 5 us + their boundaries were 0.7 % of it): segment s covers elements [start[s], start[s+1]) of one global index space.
__global__ void __launch_bounds__(256) k_pack_multi(const PackSegments sg, const ParamPtrs ptrs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sg.start[sg.n]) return;
    int s = 0;
#pragma unroll
    for (int k = 1; k < kMaxPackSegments; ++k) s += (k < sg.n && i >= sg.start[k]) ? 1 : 0;
    const int64_t j = i - sg.start[s];
    const int32_t e = sg.table[s][j];
    float v = 0.0f;
    if (e >= 0) v = ptrs.p[e >> 20][e & 0xFFFFF];
    if (sg.bf16[s]) reinterpret_cast<__bf16*>(sg.out[s])[j] = (__bf16)v;
    else reinterpret_cast<float*>(sg.out[s])[j] = v;
}

hipError_t launch_pack_multi(const PackSegments& sg, const ParamPtrs& ptrs, hipStream_t st) {
    const int64_t n = sg.start[sg.n];
    hipLaunchKernelGGL(k_pack_multi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sg, ptrs);
    return hipGetLastError();
}

hipError_t launch_pack(const int32_t* table, int64_t n, const ParamPtrs& ptrs, void* out, bool bf16, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (bf16)
        hipLaunchKernelGGL((k_pack<__bf16>), dim3(grid), dim3(256), 0, st, table, n, ptrs, (__bf16*)out);
    else
        hipLaunchKernelGGL((k_pack<float>), dim3(grid), dim3(256), 0, st, table, n, ptrs, (float*)out);
    return hipGetLastError();
}

}  // namespace mip
