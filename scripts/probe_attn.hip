// Phase timeline of the attention kernel at config 2 (GPU box).  Builds the product source with -DCFM_ATTN_STAMPS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCFM_ATTN_STAMPS -Iinclude -Iconformer-pytorch-lightning_amd/csrc \
//         scripts/probe_attn.hip conformer-pytorch-lightning_amd/csrc/abi.cpp -o scripts/bin/probe_attn
#include "attention.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

static void* dalloc(size_t bytes, int fill) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
    (void)hipMemset(p, fill, bytes);
    return p;
}

int main() {
    const int B = 32, H = 4, T = 249, dk = 64, D = H * dk;
    const int64_t M = (int64_t)B * T;
    void* qkv = dalloc((size_t)M * 3 * D * 2, 0x11);       // fused [M, 3D] 16-bit, as the macaron chain writes it
    void* pos = dalloc((size_t)B * D * 2, 0x11);
    float* uv = (float*)dalloc(2 * D * 4, 0);
    uint8_t* mask = (uint8_t*)dalloc((size_t)B * T, 1);
    void* out = dalloc((size_t)M * D * 2, 0);
    cfm_attn_desc d = {};
    d.q = qkv; d.k = (const char*)qkv + D * 2; d.v = (const char*)qkv + 2 * D * 2; d.p = pos; d.bias_u = uv; d.bias_v = uv + D;
    d.mask = mask; d.out = out;
    d.q_sb = d.k_sb = d.v_sb = (int64_t)T * 3 * D; d.q_st = d.k_st = d.v_st = 3 * D; d.k_sh = d.v_sh = dk;
    d.p_sb = D; d.p_st = 0; d.m_sb = T; d.m_sq = 0;
    d.B = B; d.H = H; d.Tq = T; d.Tk = T; d.dk = dk;
    d.q_dtype = d.kv_dtype = d.p_dtype = d.out_dtype = d.mma_dtype = CFM_BF16; d.split = 0; d.scale = 0.125f;
    for (int i = 0; i < 5; ++i) if (cfm_attention(&d, nullptr) != 0) { fprintf(stderr, "%s\n", cfm_last_error()); return 1; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) cfm_attention(&d, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int nb = 4 * H * B;   // (1-D grid of the same size when B*H is a multiple of 8)
    std::vector<long long> h(2048 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(cfm_attn_stamps), sizeof(long long) * 2048 * 8);
    printf("attention B=%d H=%d T=%d: %.2f us/launch back-to-back, %d workgroups\n", B, H, T, ms * 1000.f / reps, nb);
    const char* ph[5] = {"", "q load + bias + bd", "stage K/V/mask + barrier", "4 key tiles", "normalise + store"};
    for (int p = 1; p <= 4; ++p) {
        double s = 0; long long mx = 0;
        for (int b = 0; b < nb; ++b) { const long long dt = h[b * 8 + p] - h[b * 8 + p - 1]; s += dt; mx = dt > mx ? dt : mx; }
        printf("    %-26s %8.0f cycles mean %8lld max\n", ph[p], s / nb, mx);
    }
    long long w0 = h[6], w1 = h[7], wls = h[6];
    double wsum = 0;
    for (int b = 0; b < nb; ++b) {
        w0 = h[b * 8 + 6] < w0 ? h[b * 8 + 6] : w0; wls = h[b * 8 + 6] > wls ? h[b * 8 + 6] : wls; w1 = h[b * 8 + 7] > w1 ? h[b * 8 + 7] : w1;
        wsum += (double)(h[b * 8 + 7] - h[b * 8 + 6]);
    }
    printf("    in-kernel mean %.2f us; first start -> last start %.2f us, -> last end %.2f us (100 MHz wall clock)\n", wsum / nb / 100.0,
           (wls - w0) / 100.0, (w1 - w0) / 100.0);
    return 0;
}
