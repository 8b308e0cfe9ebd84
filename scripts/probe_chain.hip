// Phase timeline of the row-chain kernels (GPU box).  Builds the product kernel source with -DCFM_CHAIN_STAMPS so thread 0 of
// every workgroup records the shader clock at each phase boundary, runs the three D=256 chains of a conformer block on
// random data (timing only: the numbers in the buffers are meaningless) and prints mean phase lengths in shader cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCFM_CHAIN_STAMPS -Iinclude -Iconformer-pytorch-lightning_amd/csrc \
//         scripts/probe_chain.hip conformer-pytorch-lightning_amd/csrc/abi.cpp -o scripts/bin/probe_chain
#include "rowchain.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    if (d.cin_a) {
        double s14 = 0, s15 = 0, s1 = 0;
        for (int b = 0; b < nb; ++b) { s14 += (double)(h[b * 16 + 14] - h[b * 16 + 0]); s15 += (double)(h[b * 16 + 15] - h[b * 16 + 14]); s1 += (double)(h[b * 16 + 1] - h[b * 16 + 15]); }
        printf("    conv-in stage: context tile + out-projection %9.0f cycles, rows + LN_conv %9.0f, GLU product + depthwise stage %9.0f (inside the first line below)\n", s14 / nb, s15 / nb, s1 / nb);
    }
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

// D = 512 (config 4: 3 984 rows = 125 workgroups): the three plain chains; stamps 3 -> 4 (phase 1, first half of FF) -> 15 (phase 2) -> 14 (phase 1,
// second half) -> 5 (phase 2 + y tile)
static void run_wide(const char* name, cfm_rowchain_desc d, int M) {
    d.M = M;
    const int grid = (M + 31) / 32;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) if (cfm_rowchain(&d, nullptr) != 0) { fprintf(stderr, "%s: %s\n", name, cfm_last_error()); exit(1); }
    {
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
    const int pgrid = (d.psum_out || d.tail_pair) ? ((grid + 3) / 4) * 8 : grid;
    const int nb = pgrid < 1024 ? pgrid : 1024;
    static const int order[10] = {0, 1, 2, 3, 4, 15, 14, 5, 6, 7};
    static const char* what[10] = {"", "head: stage A tile", "head: GEMM + epilogue", "rows + LN_in", "FFN phase 1, half 0", "FFN phase 2, half 0", "FFN phase 1, half 1",
                                   "FFN phase 2, half 1 + y tile", "post norms", "tail GEMM + stores"};
    printf("%-12s D=512 M=%5d grid=%4d  %7.2f us/launch (back-to-back)\n", name, M, grid, ms * 1000.f / reps);
    for (int b = 0; b < nb; ++b)
        for (int i = 1; i < 10; ++i)
            if (h[b * 16 + order[i]] == 0) h[b * 16 + order[i]] = h[b * 16 + order[i - 1]];
    double tot = 0;
    for (int i = 1; i < 10; ++i) {
        double sum = 0; long long mx = 0;
        for (int b = 0; b < nb; ++b) {
            const long long dt = h[b * 16 + order[i]] - h[b * 16 + order[i - 1]];
            sum += (double)dt; mx = dt > mx ? dt : mx;
        }
        printf("    %-30s %9.0f cycles mean   %9lld max\n", what[i], sum / nb, mx);
        tot += sum / nb;
    }
    double wsum = 0;
    for (int b = 0; b < nb; ++b) wsum += (double)(h[b * 16 + 9] - h[b * 16 + 8]);
    printf("    in-kernel total %9.0f cycles mean = %.2f us mean by the 100 MHz wall clock\n", tot, wsum / nb / 100.0);
}

// random contents (argv[2] = 1): N(0, scale) as f32 or bf16 -- the MFMA inputs toggle as they do in the encoder (all-zero operands draw less power)
static void fill_random(void* p, size_t n, bool half, float scale, unsigned seed) {
    std::vector<float> f(half ? 0 : n);
    std::vector<unsigned short> h(half ? n : 0);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; acc += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        const float v = acc * 1.7320508f * scale;
        if (half) { unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16); } else f[i] = v;
    }
    (void)hipMemcpy(p, half ? (void*)h.data() : (void*)f.data(), n * (half ? 2 : 4), hipMemcpyHostToDevice);
}

static int main_wide(bool rnd) {
    const int D = 512, FF = 2048, MMAX = 7968;
    cfm_rowchain_desc z = {};
    float* x = (float*)dalloc((size_t)MMAX * D * 4, 0);
    float* res = (float*)dalloc((size_t)MMAX * D * 4, 0);
    float* out = (float*)dalloc((size_t)MMAX * D * 4, 0);
    void* a16 = dalloc((size_t)MMAX * D * 2, 0);
    void* t16 = dalloc((size_t)MMAX * 3 * D * 2, 0);
    void* w1f = dalloc((size_t)FF * D * 2, 0x11);
    void* w2f = dalloc((size_t)FF * D * 2, 0x11);
    void* wh = dalloc((size_t)D * D * 2, 0x11);
    void* wt = dalloc((size_t)3 * D * D * 2, 0x11);
    float* vec = (float*)dalloc(4096 * 4, 0);
    if (rnd) {
        printf("random operands\n");
        fill_random(x, (size_t)MMAX * D, false, 1.0f, 1); fill_random(res, (size_t)MMAX * D, false, 1.0f, 2); fill_random(a16, (size_t)MMAX * D, true, 1.0f, 3);
        fill_random(w1f, (size_t)FF * D, true, 0.04f, 4); fill_random(w2f, (size_t)FF * D, true, 0.02f, 5); fill_random(wh, (size_t)D * D, true, 0.04f, 6);
        fill_random(wt, (size_t)3 * D * D, true, 0.04f, 7); fill_random(vec, 4096, false, 1.0f, 8);
    }
    uint8_t* mask = (uint8_t*)dalloc(MMAX, 1);
    cfm_rowchain_desc mac = z;
    mac.x = x; mac.ln_g = vec; mac.ln_b = vec; mac.w1f = w1f; mac.w2n = w2f; mac.b1 = vec; mac.b2 = vec; mac.ln2_g = vec; mac.ln2_b = vec;
    mac.out_f32 = out; mac.tail_w = wt; mac.tail_b = vec; mac.tail_out = t16; mac.D = D; mac.FF = FF; mac.tail_N = 3 * D; mac.w_dtype = CFM_BF16;
    mac.alpha = 0.5f; mac.eps = 1e-5f;
    cfm_rowchain_desc cin = z;
    cin.head_a = a16; cin.head_w = wh; cin.head_b = vec; cin.head_res = res; cin.ln_g = vec; cin.ln_b = vec; cin.ln_mask = mask;
    cin.out_f32 = out; cin.tail_w = wt; cin.tail_b = vec; cin.tail_out = t16; cin.D = D; cin.tail_N = 2 * D; cin.tail_glu = 1; cin.w_dtype = CFM_BF16;
    cin.eps = 1e-5f;
    cfm_rowchain_desc fin = z;
    fin.head_a = a16; fin.head_w = wh; fin.head_b = vec; fin.head_res = res; fin.head_mask = mask; fin.ln_g = vec; fin.ln_b = vec;
    fin.w1f = w1f; fin.w2n = w2f; fin.b1 = vec; fin.b2 = vec; fin.ln1_g = vec; fin.ln1_b = vec; fin.out_f32 = out; fin.D = D; fin.FF = FF;
    fin.w_dtype = CFM_BF16; fin.alpha = 0.5f; fin.eps = 1e-5f;
    for (int M : {32, 3984, 7968}) {
        run_wide("macaron", mac, M);
        run_wide("conv-in", cin, M);
        run_wide("final", fin, M);
    }
    {   // the workgroup-pair launches config 4 runs (M = 3 984: 125 tiles x 2)
        float* ps = (float*)dalloc((size_t)3 * MMAX * D * 4, 0);
        cfm_rowchain_desc mh = mac;                        // macaron half: LN -> half of FF -> partial slab
        mh.ln2_g = nullptr; mh.ln2_b = nullptr; mh.tail_w = nullptr; mh.tail_b = nullptr; mh.tail_out = nullptr; mh.tail_N = 0; mh.out_f32 = nullptr; mh.psum_out = ps;
        cfm_rowchain_desc qp = z;                          // reduce + LN_mha + q|k|v, columns split over the pair
        qp.x = x; qp.psum_in = ps; qp.psum_b2 = vec; qp.psum_alpha = 0.5f; qp.out_f32 = out; qp.ln_g = vec; qp.ln_b = vec; qp.tail_w = wt; qp.tail_b = vec;
        qp.tail_out = t16; qp.tail_N = 3 * D; qp.D = D; qp.FF = FF; qp.w_dtype = CFM_BF16; qp.alpha = 1.0f; qp.eps = 1e-5f; qp.tail_pair = 1;
        cfm_rowchain_desc cp = cin;                        // conv-in, GLU tail split over the pair (rows to a third buffer)
        cp.tail_pair = 1; cp.out_f32 = ps + (size_t)2 * MMAX * D;
        cfm_rowchain_desc rw = z;                          // reduce + LN_final
        rw.x = x; rw.psum_in = ps; rw.psum_b2 = vec; rw.psum_alpha = 0.5f; rw.ln_g = vec; rw.ln_b = vec; rw.out2_f32 = out; rw.D = D; rw.FF = FF; rw.w_dtype = CFM_BF16;
        rw.alpha = 1.0f; rw.eps = 1e-5f;
        run_wide("macaron-half", mh, 3984);
        run_wide("q|k|v pair", qp, 3984);
        run_wide("conv-in pair", cp, 3984);
        run_wide("rows", rw, 3984);
    }
    {   // cold weights: 17 rotating weight sets (as the 17 blocks of config 4 do), M = 3 984
        const int L = 17;
        std::vector<cfm_rowchain_desc> macs(L, mac), cins(L, cin), fins(L, fin);
        for (int l = 0; l < L; ++l) {
            void* a1 = dalloc((size_t)FF * D * 2, 0x11); void* a2 = dalloc((size_t)FF * D * 2, 0x11);
            void* a3 = dalloc((size_t)3 * D * D * 2, 0x11); void* a4 = dalloc((size_t)D * D * 2, 0x11);
            void* b1 = dalloc((size_t)FF * D * 2, 0x11); void* b2 = dalloc((size_t)FF * D * 2, 0x11);
            void* b3 = dalloc((size_t)2 * D * D * 2, 0x11); void* b4 = dalloc((size_t)D * D * 2, 0x11);
            macs[l].w1f = a1; macs[l].w2n = a2; macs[l].tail_w = a3; macs[l].M = 3984;
            cins[l].head_w = a4; cins[l].tail_w = b3; cins[l].M = 3984;
            fins[l].w1f = b1; fins[l].w2n = b2; fins[l].head_w = b4; fins[l].M = 3984;
        }
        for (int which = 0; which < 4; ++which) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            auto go = [&](int l) {
                if (which == 0 || which == 3) cfm_rowchain(&macs[l], nullptr);
                if (which == 1 || which == 3) cfm_rowchain(&cins[l], nullptr);
                if (which == 2 || which == 3) cfm_rowchain(&fins[l], nullptr);
            };
            for (int l = 0; l < L; ++l) go(l);
            (void)hipEventRecord(e0, nullptr);
            const int reps = which == 3 ? 200 : 10;         // the whole block: long enough (0.45 s) for the clocks to settle under the load
            for (int r = 0; r < reps; ++r)
                for (int l = 0; l < L; ++l) go(l);
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            static const char* nm[4] = {"macaron", "conv-in", "final", "macaron + conv-in + final"};
            printf("%s, 17 rotating weight sets (L2-cold weights), M = 3984: %.2f us per %s back-to-back\n", nm[which], ms * 1000.f / (reps * L), which == 3 ? "block" : "launch");
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 512) return main_wide(argc > 2 && atoi(argv[2]) == 1);
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
        cfm_rowchain_desc cm = ch;                        // + this block's conv-in chain as the input stage (CIN): stamps 14 = out-proj done, 15 = LN_conv done
        float* x2 = (float*)dalloc((size_t)MMAX * D * 4, 0);
        void* wg = dalloc((size_t)2 * D * D * 2, 0x11);
        cm.cin_a = a16; cm.cin_w = wh; cm.cin_b = vec; cm.cin_res = res; cm.cin_out = x2; cm.cin_ln_g = vec; cm.cin_ln_b = vec; cm.cin_mask = mask;
        cm.cin_tail_w = wg; cm.cin_tail_b = vec; cm.head_res = x2;
        run_seg2("conv-in+dw-final+macaron", cm, 7968);
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
