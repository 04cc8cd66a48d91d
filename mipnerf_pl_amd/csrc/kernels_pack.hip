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

hipError_t launch_pack(const int32_t* table, int64_t n, const ParamPtrs& ptrs, void* out, bool bf16, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (bf16)
        hipLaunchKernelGGL((k_pack<__bf16>), dim3(grid), dim3(256), 0, st, table, n, ptrs, (__bf16*)out);
    else
        hipLaunchKernelGGL((k_pack<float>), dim3(grid), dim3(256), 0, st, table, n, ptrs, (float*)out);
    return hipGetLastError();
}

}  // namespace mip
