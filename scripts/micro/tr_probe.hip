// Probe: what does ds_read_b64_tr_b16 (gfx950) return?  LDS holds lds[i] = i (16-bit); lane L passes the byte address of
// element 4 L (its own 8-byte granule in a lane-linear layout); print the 4 values each lane gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int stride_elems) {
    __shared__ short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
    __syncthreads();
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + threadIdx.x * stride_elems));
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {4, 16, 32}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("stride %d elements between lane addresses:\n", stride);
        for (int l = 0; l < 64; ++l) printf("  lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, l * stride, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
