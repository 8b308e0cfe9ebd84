// Which SIMD does wavefront w of a 512-thread workgroup run on?  (HW_ID bits [5:4] = SIMD id on gfx9-family parts.)
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_simd_map.hip -o scripts/bin/probe_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    __shared__ unsigned pad[30000];                       // 120 KB: one workgroup per CU, like the 256 x 256 GEMM tile
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw + (pad[threadIdx.x] & 0);
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 8 * 4);
    k<<<4, 512>>>(d);
    unsigned h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
        printf("workgroup %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d: simd %u wave_slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
