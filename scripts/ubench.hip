// Standalone microbenchmarks (GPU box): hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o /tmp/ubench && /tmp/ubench
//  A) cycles per v_mfma_f32_16x16x32_bf16 for 8 independent accumulators, operands in registers
//  B) L2-resident streaming: 1 KB wavefront loads, NFL loads in flight per wavefront, 4 wavefronts per workgroup
// each with 1 and 256 workgroups; reports shader clock (s_memtime / s_memrealtime) as well.
// CAVEAT: only wavefront 0 of a workgroup is timed while the bytes of ALL its wavefronts are counted; wavefront 0 is the oldest and
// gets issue priority, so the B/clk/CU figures at 8 and 16 wavefronts are upper bounds (a whole workgroup's FFN phase in the real
// kernel sustains 49-61 B/clk/CU: scripts/probe_chain.hip).  The per-wavefront issue limit (~16 B/clk) is what this file is good for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_mfma(float* out, long long* t, int iters) {
    u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {5u, threadIdx.x, 7u, 9u};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int NFL, int NW>
__global__ __launch_bounds__(NW * 64) void k_stream(const u32x4* __restrict__ buf, size_t n16, unsigned* out, long long* t, int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every workgroup walks the same 2 MB (n16 16-byte elements), wavefront w takes every NW-th KB, rotated by block
    // n16 is a power of two: index arithmetic is a 32-bit mask (an earlier version used a 64-bit '%' here, ~250 clk of software
    // division per load, which capped every configuration at ~4 B/clk/wavefront and hid the memory system entirely)
    const unsigned mask = (unsigned)(n16 / 64) - 1u;
    unsigned kb = (unsigned)__builtin_amdgcn_readfirstlane(wave) + NW * (blockIdx.x % 16);
    u32x4 r[NFL];
    unsigned acc = 0;
    long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < NFL; ++i) { r[i] = buf[(size_t)(kb & mask) * 64 + lane]; kb += NW; }
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int i = 0; i < NFL; ++i) {
            acc += r[i].x ^ r[i].w;
            r[i] = buf[(size_t)(kb & mask) * 64 + lane];
            kb += NW;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < NFL; ++i) acc += r[i].y;
    long long c1 = clock64(), w1 = wall_clock64();
    out[(blockIdx.x * NW * 64 + threadIdx.x) % 65536] = acc;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
}

static void report(const char* name, long long* d_t, int grid, double per_wave_units, const char* unit) {
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), d_t, sizeof(long long) * grid * 2, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    cyc /= grid; wall /= grid;
    printf("%-34s grid %4d: %9.0f shader cycles, %7.2f us, clock %.2f GHz, %8.3f %s\n", name, grid, cyc, wall / 100.0, cyc / (wall * 10.0), per_wave_units / cyc, unit);
}

int main() {
    float* d_out; long long* d_t; unsigned* d_o2; u32x4* d_buf;
    const size_t bytes = 2u << 20;
    hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_t, 256 * 2 * 8); hipMalloc(&d_o2, 256 * 256 * 4); hipMalloc(&d_buf, bytes);
    hipMemset(d_buf, 1, bytes);
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {1, 64, 256}) {
            hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, d_out, d_t, 2000);
            hipDeviceSynchronize();
            if (rep) report("MFMA 16x16x32 bf16 (cycles/MFMA = 1/x)", d_t, grid, 16000.0, "MFMA/cycle/wave");
        }
#define STREAM(NFL, NW)                                                                                              \
    for (int grid : {1, 256}) {                                                                                      \
        for (int rep = 0; rep < 2; ++rep) {                                                                          \
            hipLaunchKernelGGL((k_stream<NFL, NW>), dim3(grid), dim3(NW * 64), 0, 0, d_buf, bytes / 16, d_o2, d_t, 4096 / NFL); \
            hipDeviceSynchronize();                                                                                  \
        }                                                                                                            \
        report("stream: in flight/wave=" #NFL " waves/CU=" #NW, d_t, grid, NW * 1024.0 * (4096 / NFL * NFL + NFL), "B/clk/CU");   \
    }
    STREAM(2, 4) STREAM(4, 4) STREAM(8, 4) STREAM(16, 4) STREAM(32, 4) STREAM(4, 8) STREAM(8, 8) STREAM(16, 8) STREAM(4, 16) STREAM(8, 16)
    return 0;
}
