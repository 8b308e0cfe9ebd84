// Phase timeline of the row-chain kernels (GPU box).  Builds the product kernel source with -DCFM_CHAIN_STAMPS so thread 0 of
// every workgroup records the shader clock at each phase boundary, runs the three D=256 chains of a conformer block on
// random data (timing only: the numbers in the buffers are meaningless) and prints mean phase lengths in shader cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCFM_CHAIN_STAMPS -Iinclude -Iconformer-pytorch-lightning_amd/csrc \
//         scripts/probe_chain.hip conformer-pytorch-lightning_amd/csrc/abi.cpp -o scripts/bin/probe_chain
#include "rowchain.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

static void* dalloc(size_t bytes, int fill) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
    (void)hipMemset(p, fill, bytes);
    return p;
}

static const char* PHASE[8] = {"", "head: stage A tile", "head: GEMM + epilogue", "rows + LN_in", "FFN phase 1 (hidden)", "FFN phase 2 + y tile", "post norms", "tail GEMM + stores"};

// the chained launch (final chain of block i + macaron chain of block i+1): stamps 0..5 first segment, 13 its post norms, 10..12 the second
// segment's LayerNorm / phase 1 / phase 2, 6 post norms, 7 tail
static void run_seg2(const char* name, cfm_rowchain_desc d, int M) {
    d.M = M;
    const int grid = (M + 31) / 32;
    for (int i = 0; i < 5; ++i) if (cfm_rowchain(&d, nullptr) != 0) { fprintf(stderr, "%s: %s\n", name, cfm_last_error()); exit(1); }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) cfm_rowchain(&d, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(1024 * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(cfm_chain_stamps), sizeof(long long) * 1024 * 16);
    const int nb = grid < 1024 ? grid : 1024;
    static const int order[12] = {0, 1, 2, 3, 4, 5, 13, 10, 11, 12, 6, 7};
    static const char* what[12] = {"", "depthwise + BatchNorm + SiLU input stage", "head GEMM (pointwise-conv-2) + epilogue", "rows + LN_ff", "FFN phase 1 (hidden)",
                                   "FFN phase 2 + y tile", "post norms (LN_final)", "LN_ffm (next block)", "FFN_m phase 1", "FFN_m phase 2 + y tile",
                                   "post norms (LN_mha)", "tail GEMM (q|k|v) + stores"};
    printf("%-10s M=%5d grid=%4d  %7.2f us/launch (back-to-back)\n", name, M, grid, ms * 1000.f / reps);
    double tot = 0;
    for (int i = 1; i < 12; ++i) {
        double sum = 0; long long mx = 0;
        for (int b = 0; b < nb; ++b) {
            const long long dt = h[b * 16 + order[i]] - h[b * 16 + order[i - 1]];
            sum += (double)dt; mx = dt > mx ? dt : mx;
        }
        printf("    %-42s %9.0f cycles mean   %9lld max\n", what[i], sum / nb, mx);
        tot += sum / nb;
    }
    double wsum = 0;
    for (int b = 0; b < nb; ++b) wsum += (double)(h[b * 16 + 9] - h[b * 16 + 8]);
    printf("    in-kernel total %9.0f cycles mean = %.2f us mean by the 100 MHz wall clock (=> %.2f GHz)\n", tot, wsum / nb / 100.0, tot / (wsum / nb * 10.0));
}

static void run(const char* name, cfm_rowchain_desc d, int M) {
    d.M = M;
    const int grid = (M + 31) / 32;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (cfm_rowchain(&d, nullptr) != 0) { fprintf(stderr, "%s: %s\n", name, cfm_last_error()); exit(1); }
    {   // phases an instance does not have leave their stamp untouched: zero them, then inherit the previous stamp below
        std::vector<long long> zero(1024 * 16, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(cfm_chain_stamps), zero.data(), sizeof(long long) * 1024 * 16);
    }
    (void)hipEventRecord(e0, nullptr);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) cfm_rowchain(&d, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(1024 * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(cfm_chain_stamps), sizeof(long long) * 1024 * 16);
    const int nb = grid < 1024 ? grid : 1024;
    for (int b = 0; b < nb; ++b)
        for (int ph = 1; ph < 8; ++ph)
            if (h[b * 16 + ph] == 0) h[b * 16 + ph] = h[b * 16 + ph - 1];
    printf("%-10s M=%5d grid=%4d  %7.2f us/launch (back-to-back)\n", name, M, grid, ms * 1000.f / reps);
    double tot = 0;
    long long last[8] = {0};
    for (int ph = 1; ph < 8; ++ph) {
        double sum = 0; long long mx = 0;
        for (int b = 0; b < nb; ++b) {
            // a phase that does not exist in this instance leaves its stamp equal to an earlier one: walk back to the last stamp written
            long long t1 = h[b * 16 + ph], t0 = h[b * 16 + ph - 1];
            const long long dt = t1 - t0;
            sum += (double)dt; mx = dt > mx ? dt : mx;
        }
        printf("    %-24s %9.0f cycles mean   %9lld max\n", PHASE[ph], sum / nb, mx);
        tot += sum / nb;
        (void)last;
    }
    long long w_first = h[8], w_last = h[9], w_lastst = h[8];
    double wsum = 0;
    for (int b = 0; b < nb; ++b) {
        w_first = h[b * 16 + 8] < w_first ? h[b * 16 + 8] : w_first;
        w_lastst = h[b * 16 + 8] > w_lastst ? h[b * 16 + 8] : w_lastst;
        w_last = h[b * 16 + 9] > w_last ? h[b * 16 + 9] : w_last;
        wsum += (double)(h[b * 16 + 9] - h[b * 16 + 8]);
    }
    printf("    in-kernel total %9.0f cycles mean = %.2f us mean by the 100 MHz wall clock (=> %.2f GHz);  first start -> last start %.2f us, -> last end %.2f us\n",
           tot, wsum / nb / 100.0, tot / (wsum / nb * 10.0), (w_lastst - w_first) / 100.0, (w_last - w_first) / 100.0);
}

int main() {
    const int D = 256, FF = 2048, MMAX = 7968;
    cfm_rowchain_desc z = {};
    float* x = (float*)dalloc((size_t)MMAX * D * 4, 0);
    float* res = (float*)dalloc((size_t)MMAX * D * 4, 0);
    float* out = (float*)dalloc((size_t)MMAX * D * 4, 0);
    void* a16 = dalloc((size_t)MMAX * D * 2, 0);
    void* t16 = dalloc((size_t)MMAX * 768 * 2, 0);
    void* w1f = dalloc((size_t)FF * D * 2, 0x11);
    void* w2f = dalloc((size_t)FF * D * 2, 0x11);
    void* wh = dalloc((size_t)D * D * 2, 0x11);
    void* wt = dalloc((size_t)768 * D * 2, 0x11);
    float* vec = (float*)dalloc(4096 * 4, 0);
    uint8_t* mask = (uint8_t*)dalloc(MMAX, 1);

    cfm_rowchain_desc mac = z;                             // macaron: LN, FFN, +res, LN_mha, QKV
    mac.x = x; mac.ln_g = vec; mac.ln_b = vec; mac.w1f = w1f; mac.w2n = w2f; mac.b1 = vec; mac.b2 = vec; mac.ln2_g = vec; mac.ln2_b = vec;
    mac.out_f32 = out; mac.tail_w = wt; mac.tail_b = vec; mac.tail_out = t16; mac.D = D; mac.FF = FF; mac.tail_N = 768; mac.w_dtype = CFM_BF16;
    mac.alpha = 0.5f; mac.eps = 1e-5f;

    cfm_rowchain_desc cin = z;                             // conv-in: out-proj + res, LN_conv + mask, pw1 + GLU
    cin.head_a = a16; cin.head_w = wh; cin.head_b = vec; cin.head_res = res; cin.ln_g = vec; cin.ln_b = vec; cin.ln_mask = mask;
    cin.out_f32 = out; cin.tail_w = wt; cin.tail_b = vec; cin.tail_out = t16; cin.D = D; cin.tail_N = 512; cin.tail_glu = 1; cin.w_dtype = CFM_BF16;
    cin.eps = 1e-5f;

    cfm_rowchain_desc fin = z;                             // final: pw2 + mask + res, LN_ff, FFN, +res, LN_final
    fin.head_a = a16; fin.head_w = wh; fin.head_b = vec; fin.head_res = res; fin.head_mask = mask; fin.ln_g = vec; fin.ln_b = vec;
    fin.w1f = w1f; fin.w2n = w2f; fin.b1 = vec; fin.b2 = vec; fin.ln1_g = vec; fin.ln1_b = vec; fin.out_f32 = out; fin.D = D; fin.FF = FF;
    fin.w_dtype = CFM_BF16; fin.alpha = 0.5f; fin.eps = 1e-5f;

    cfm_rowchain_desc fdw = fin;                           // final with the depthwise conv + BatchNorm + SiLU in its input stage
    fdw.dw_w = vec; fdw.dw_b = vec; fdw.dw_scale = vec; fdw.dw_shift = vec; fdw.dw_T = 249; fdw.dw_K = 15;
    run("dw-final", fdw, 7968);
    {
        cfm_rowchain_desc ch = fdw;                       // + the next block's macaron chain in the same launch (SEG2)
        ch.out_f32 = nullptr;
        ch.s2_ln_g = vec; ch.s2_ln_b = vec; ch.s2_w1f = w1f; ch.s2_w2n = w2f; ch.s2_b1 = vec; ch.s2_b2 = vec; ch.s2_out_f32 = out; ch.s2_alpha = 0.5f;
        ch.ln2_g = vec; ch.ln2_b = vec; ch.tail_w = wt; ch.tail_b = vec; ch.tail_out = t16; ch.tail_N = 768;
        run_seg2("dw-final+macaron", ch, 7968);
    }

    {   // cold weights: rotate 12 weight sets (as 12 layers do), so every launch first-touches its weights on all 8 XCDs
        const int L = 12;
        std::vector<cfm_rowchain_desc> macs(L, mac), fdws(L, fdw);
        for (int l = 0; l < L; ++l) {
            void* a1 = dalloc((size_t)FF * D * 2, 0x11); void* a2 = dalloc((size_t)FF * D * 2, 0x11);
            void* a3 = dalloc((size_t)768 * D * 2, 0x11); void* a4 = dalloc((size_t)D * D * 2, 0x11);
            void* junk = dalloc((size_t)24 << 20, 0x11); (void)junk;   // spacing between the sets
            macs[l].w1f = a1; macs[l].w2n = a2; macs[l].tail_w = a3; macs[l].M = 7968;
            fdws[l].w1f = a1; fdws[l].w2n = a2; fdws[l].head_w = a4; fdws[l].M = 7968;
        }
        for (int which = 0; which < 2; ++which) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int l = 0; l < L; ++l) cfm_rowchain(which ? &fdws[l] : &macs[l], nullptr);
            (void)hipEventRecord(e0, nullptr);
            const int reps = 10;
            for (int r = 0; r < reps; ++r)
                for (int l = 0; l < L; ++l) cfm_rowchain(which ? &fdws[l] : &macs[l], nullptr);
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s, 12 rotating weight sets (L2-cold weights): %.2f us/launch back-to-back\n", which ? "dw-final" : "macaron", ms * 1000.f / (reps * L));
        }
    }

    for (int M : {32, 7968}) {
        run("macaron", mac, M);
        run("conv-in", cin, M);
        run("final", fin, M);
    }
    return 0;
}
