// Micro-benchmark: HBM write bandwidth of the T-block store pattern of the training kernels (no compute).
// mode 0: consumer-friendly layout [wave_tile][block][2 KiB]  (what the kernels use)
// mode 1: producer-friendly layout [wg_tile][block][wave][2 KiB] (8 waves of a workgroup write 16 KiB contiguous)
// mode 2: linear fill by the same grid (upper bound)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(512) k(char* out, int ntiles, int nblk, int mode, int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int b = 0; b < nblk; ++b) {
            size_t off;
            if (mode == 0) off = ((size_t)(tile * 8 + wave) * nblk + b) * 2048;
            else if (mode == 1) off = (((size_t)tile * nblk + b) * 8 + wave) * 2048;
            else off = ((size_t)tile * nblk + b) * 16384 + wave * 2048;
            f4* p = reinterpret_cast<f4*>(out + off) + lane;
            if (nt) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, p + 64); }
            else { p[0] = v; p[64] = v; }
            if (mode != 2) __builtin_amdgcn_s_sleep(20);     // spread the tile's stores in time like the real kernels do
        }
    }
}
int main() {
    const int ntiles = 2048, nblk = 72;
    const size_t bytes = (size_t)ntiles * 8 * nblk * 2048;
    char* d; hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, ntiles, nblk, mode, nt);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, ntiles, nblk, mode, nt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("mode %d nt %d: %.3f ms  %.2f TB/s (%.2f GB)\n", mode, nt, ms, bytes / ms / 1e9, bytes / 1e9);
    }
    return 0;
}
