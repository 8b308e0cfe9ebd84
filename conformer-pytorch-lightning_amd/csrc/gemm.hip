// gemm.hip -- C = epilogue(A[M,K] . W[N,K]^T) on gfx950 MFMA (v_mfma_f32_16x16x32_{bf16,f16}).
//
// One kernel template serves every dense contraction of the conformer block:
//   FFN w_1/w_2, fused QKV, linear_pos, linear_out, pointwise conv 1/2, front-end linear, and the
//   front-end Conv2d(D,D,3,stride 2) as an implicit GEMM over a channels-last image (CONV).
//
// Structure (256 threads = 4 wavefronts of 64 lanes, 2x2 over the BMxBN tile):
//   * both operands are K-contiguous ([M,K] activations, [N,K] nn.Linear weights), staged
//     global -> registers -> LDS in 16-byte chunks, two LDS buffers, one barrier per K tile; the
//     global loads of tile t+1 are issued before the MFMAs of tile t (async-STAGE split).
//   * LDS rows are XOR-swizzled so that the ds_read_b128 fragment reads are bank-conflict free.
//   * MFMA operand roles are SWAPPED: the weight fragment is the A operand, the activation fragment
//     the B operand, so a lane's 4 accumulator registers are 4 CONSECUTIVE OUTPUT COLUMNS of one row:
//     bias / residual loads and the output store are 8- or 16-byte vector accesses.
//   * SPLIT: activations (f32 in HBM) are split into bf16 hi+lo on the way into LDS, weights come as
//     hi/lo planes, 3 MFMAs per fragment pair (hi*lo + lo*hi + hi*hi): ~16 mantissa bits.
//   * every edge is predicated: rows clamp on load / predicate on store, K tail chunks load zeros.
#include <string>
#include <type_traits>

#include "cfm_common.h"
#include "gemm256.h"

// vector types whose address is only known to be dword aligned (rows of a [M, ldc] matrix with ldc % 4 == 2)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned int u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

struct GemmArgs {
    const void* A;
    const u16* W;
    const u16* Wlo;
    const float* bias;
    const float* res;
    const uint8_t* mask;
    void* C;
    void* Cpre;      // training: optional second output, acc + bias before the activation (row stride ld_pre, pre_dtype)
    const void* aux; // training: pre-activation (DSILU) / forward output (DRELU) of the layer whose input gradient this GEMM computes
    int64_t lda, ldc, ldr, ld_pre, ld_aux;
    int pre_dtype, aux_dtype;
    CfmDrop drop, drop2;   // output dropout (train mode): a pure function of (seed, m * N_out + n)
    int M, N, K;
    int m_begin;     // first row this launch computes (rows [m_begin, M)); 0 except for the tail of a split launch (see cfm_gemm)
    int c_dtype, act, mask_mode;
    float alpha;
    int convC, T1, F1, T2, F2;
};

// KG = 2: TWO groups of four wavefronts per tile, each walking its own half of K through its own LDS buffers; the halves meet through LDS
// before the epilogue (group 0 + group 1, a fixed order: deterministic).  For long-K products with few tiles (a feed-forward's second
// product at a training micro-batch: 300 tiles of 32 serial K steps, one workgroup per CU) -- the K chain of a workgroup is latency-bound,
// so two chains per tile nearly halve it.
template <typename HT, int BM, int BN, int BK, bool A_F32, bool SPLIT, bool CONV, int KG = 1>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void cfm_gemm_kernel(const GemmArgs g) {   // two workgroups per CU: <= 256 registers per wavefront
    constexpr int CPR = BK / 8;      // 16-byte chunks per LDS row
    constexpr int RPP = 256 / CPR;   // tile rows staged per pass of the 256 threads
    constexpr int ACH = BM / RPP;    // chunks per thread, activation tile
    constexpr int WCH = BN / RPP;    // chunks per thread, weight tile
    constexpr int FM = BM / 32;      // 16-row fragments per wave along M
    constexpr int FN = BN / 32;      // 16-col fragments per wave along N
    constexpr int RPB = 16 / CPR;    // LDS rows per 256-byte bank row
    constexpr int A_PLANE = BM * CPR;
    // (Reading the weight operand straight from a fragment-major pack, only A through LDS, was built and measured slower on the front-end --
    // conv2 250 -> 280 us: 8 more 1 KB global loads per K tile and wavefront at ~64 clk of issue each.  scripts/experiments/README.md.)
    constexpr int W_PLANE = BN * CPR;
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int BUF = (A_PLANE + W_PLANE) * NPL;
    static_assert(!SPLIT || A_F32, "split mode reads f32 activations");
    static_assert(ACH >= 1 && WCH >= 1 && (FN % 2) == 0, "tile shape");
    static_assert(KG == 1 || ((KG == 2 || KG == 4) && !CONV && !SPLIT && !A_F32), "K groups: plain 16-bit products");
    static_assert((KG - 1) * FM * FN * 256 <= 2 * BUF * KG, "the exchange area fits the staging buffers");
    __shared__ u32x4 smem_all[2 * BUF * KG];

    const int kgrp = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // wave-uniform
    u32x4* const smem = smem_all + kgrp * 2 * BUF;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so the N tiles
    // of ONE M tile are given ids 8 apart: they re-read the same activation rows from the same L2 instead of each pulling
    // them over the fabric (measured on the front-end conv: 1.49 GB of HBM traffic per launch for 0.34 GB of compulsory bytes).
    // Speed only -- any placement computes the same result.
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M - g.m_begin + BM - 1) / BM;
    const int per_group = 8 * tiles_n;
    const int grp = blockIdx.x / per_group, rem = blockIdx.x % per_group;
    const int tile_m = grp * 8 + (rem % 8);
    if (tile_m >= tiles_m) return;                      // padding blocks of the last group (uniform: before any barrier)
    const int m0 = g.m_begin + tile_m * BM;
    const int n0 = (rem / 8) * BN;

    auto lds_idx = [](int row, int c) { return row * CPR + (c ^ ((row / RPB) % CPR)); };

    // ---- per-thread staging addresses -------------------------------------------------------
    const int kc = tid % CPR;
    const int rl = tid / CPR;
    // element offsets as 32-bit values on top of the (uniform) base pointers: the loads then use SGPR-base + VGPR-offset addressing
    // instead of a 64-bit VALU add per access (the host checks that both operands are < 2^31 elements)
    unsigned a_off[ACH], w_off[WCH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        int m = m0 + i * RPP + rl;
        m = m < g.M ? m : g.M - 1;
        if constexpr (CONV) {
            const int per_b = g.T2 * g.F2;
            const int b = m / per_b;
            const int rem = m - b * per_b;
            const int t2 = rem / g.F2;
            const int f2 = rem - t2 * g.F2;
            a_off[i] = (unsigned)(((b * g.T1 + 2 * t2) * g.F1 + 2 * f2) * g.convC);
        } else {
            a_off[i] = (unsigned)((int64_t)m * g.lda);
        }
    }
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
        int n = n0 + j * RPP + rl;
        n = n < g.N ? n : g.N - 1;
        w_off[j] = (unsigned)n * (unsigned)g.K;
    }
    // implicit conv: (tap, channel) of this thread's chunk of the NEXT K tile to be requested, advanced by BK per request
    // (gload is called for K tiles 0, 1, 2, ... in order) -- no integer division in the loop
    int cv_ci = kc * 8, cv_tap = 0;
    if constexpr (CONV) {
        cv_tap = cv_ci / g.convC;
        cv_ci -= cv_tap * g.convC;
    }

    // register prefetch ring: PF K-tiles in flight per workgroup.  The K loops here are SHORT (4 tiles at K=256) and the
    // grids small (250-1000 workgroups on 256 CUs), so occupancy cannot hide global-load latency; depth has to.
    constexpr int PF = A_F32 ? 2 : 3;
    u32x4 ra[PF][A_F32 ? 1 : ACH];
    f32x4 fa[PF][A_F32 ? ACH : 1][2];
    u32x4 rw[PF][WCH];
    u32x4 rwl[PF][SPLIT ? WCH : 1];
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};

    const int nkt_all = (g.K + BK - 1) / BK;
    const int nkt = (nkt_all + KG - 1) / KG;              // K tiles per group (both groups run the same count: the barriers are shared;
    const int kt_base = kgrp * nkt;                       //  a tile past K loads zeros)
    auto gload = [&](int kt, auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        const int k0 = (kt + kt_base) * BK + kc * 8;
        const bool kv = k0 < g.K;
        unsigned koff = (unsigned)k0;
        if constexpr (CONV) {
            const int k3 = cv_tap / 3;                       // cv_tap <= 8: a multiply-shift, not a division
            const int f3 = cv_tap - 3 * k3;
            koff = (unsigned)((k3 * g.F1 + f3) * g.convC + cv_ci);
            cv_ci += BK;
            while (cv_ci >= g.convC) {
                cv_ci -= g.convC;
                ++cv_tap;
            }
        }
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            if constexpr (A_F32) {
                const f32x4* p = (const f32x4*)((const float*)g.A + (a_off[i] + koff));
                fa[S][i][0] = kv ? p[0] : zf;
                fa[S][i][1] = kv ? p[1] : zf;
            } else {
                ra[S][i] = kv ? *(const u32x4*)((const u16*)g.A + (a_off[i] + koff)) : z4;
            }
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            rw[S][j] = kv ? *(const u32x4*)(g.W + (w_off[j] + (unsigned)k0)) : z4;
            if constexpr (SPLIT) rwl[S][j] = kv ? *(const u32x4*)(g.Wlo + (w_off[j] + (unsigned)k0)) : z4;
        }
    };

    auto lstore = [&](int buf, auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        u32x4* As = smem + buf * BUF;
        u32x4* Ws = As + A_PLANE * NPL;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int idx = lds_idx(i * RPP + rl, kc);
            if constexpr (SPLIT) {
                u32x4 hi, lo;
                split8(fa[S][i][0], fa[S][i][1], hi, lo);
                As[idx] = hi;
                As[A_PLANE + idx] = lo;
            } else if constexpr (A_F32) {
                As[idx] = pack8<HT>(fa[S][i][0], fa[S][i][1]);
            } else {
                As[idx] = ra[S][i];
            }
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int idx = lds_idx(j * RPP + rl, kc);
            Ws[idx] = rw[S][j];
            if constexpr (SPLIT) Ws[W_PLANE + idx] = rwl[S][j];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const u32x4* As = smem + buf * BUF;
        const u32x4* Ws = As + A_PLANE * NPL;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int c = kk * 4 + (lane >> 4);
            u32x4 af[FM], wf[FN];
            u32x4 afl[SPLIT ? FM : 1], wfl[SPLIT ? FN : 1];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int idx = lds_idx(wr * (BM / 2) + i * 16 + (lane & 15), c);
                af[i] = As[idx];
                if constexpr (SPLIT) afl[i] = As[A_PLANE + idx];
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int idx = lds_idx(wc * (BN / 2) + j * 16 + (lane & 15), c);
                wf[j] = Ws[idx];
                if constexpr (SPLIT) wfl[j] = Ws[W_PLANE + idx];
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (SPLIT) {
                        acc[i][j] = HT::mfma(wf[j], afl[i], acc[i][j]);
                        acc[i][j] = HT::mfma(wfl[j], af[i], acc[i][j]);
                    }
                    acc[i][j] = HT::mfma(wf[j], af[i], acc[i][j]);
                }
        }
    };

    // ---- main loop: 2 LDS buffers, 1 barrier per K tile, PF tiles of global loads in flight ------------------
    auto prologue = [&](auto s) {
        if (decltype(s)::value < nkt) gload(decltype(s)::value, s);
    };
    auto body = [&](int kt, auto s) {           // tile kt lives in register slot s = kt % PF
        lstore(kt & 1, s);                       // waits (compiler-counted vmcnt) for tile kt only
        if (kt + PF < nkt) gload(kt + PF, s);    // refill the slot that was just drained
        __syncthreads();
        compute(kt & 1);
    };
    prologue(std::integral_constant<int, 0>{});
    prologue(std::integral_constant<int, 1>{});
    if constexpr (PF > 2) prologue(std::integral_constant<int, 2>{});
    // The epilogue's bias and row-mask reads are requested HERE, behind the first K tiles: issued in the epilogue they expose one more
    // full memory latency per workgroup (microseconds on a loaded chip, against a 4-8 us K loop at K = 256..512).
    const bool glu = g.act == CFM_ACT_GLU;
    const int q4 = (lane >> 4) * 4;
    f32x4 bias_r[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wc * (BN / 2) + j * 16 + q4;
        if (g.bias && col + 3 < g.N) bias_r[j] = *(const f32x4*)(g.bias + col);
        else if (g.bias && col + 1 < g.N) bias_r[j] = (f32x4){g.bias[col], g.bias[col + 1], 0.f, 0.f};   // N % 4 == 2: the last pair
        else bias_r[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bool keep_r[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = m0 + wr * (BM / 2) + i * 16 + (lane & 15);
        keep_r[i] = (g.mask && row < g.M) ? (g.mask[row] != 0) : true;
    }
    for (int kt0 = 0; kt0 < nkt; kt0 += PF) {
        body(kt0, std::integral_constant<int, 0>{});
        if (kt0 + 1 < nkt) body(kt0 + 1, std::integral_constant<int, 1>{});
        if constexpr (PF > 2) {
            if (kt0 + 2 < nkt) body(kt0 + 2, std::integral_constant<int, 2>{});
        }
    }

    if constexpr (KG > 1) {                               // the other groups' partial sums -> LDS -> group 0, added in group order
        __syncthreads();                                  // every wavefront is done reading its last K tile
        f32x4* const xch = (f32x4*)smem_all;
        if (kgrp > 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) xch[((kgrp - 1) * FM * FN + i * FN + j) * 256 + tid] = acc[i][j];
        }
        __syncthreads();
        if (kgrp > 0) return;
#pragma unroll
        for (int q = 0; q < KG - 1; ++q)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] += xch[(q * FM * FN + i * FN + j) * 256 + tid];
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // All global READS of the epilogue are issued in one batch before the first store: C may alias `residual`
    // (in-place accumulate) and the compiler cannot prove it does not alias `bias`, so a load placed after a store is
    // serialised behind it -- 16 fragments x one exposed L2 latency each (measured: +6 us per launch for the bias alone).
    f32x4 res_r[FM][FN];
    const bool dact = g.act == CFM_ACT_DSILU || g.act == CFM_ACT_DRELU;
    auto store_pre = [&](int row, int col, const f32x4& pv) {   // training: the pre-activation, 4 columns (N % 4 == 0 checked by the host)
        const int64_t po = (int64_t)row * g.ld_pre + col;
        if (g.pre_dtype == CFM_F32) *(f32x4*)((float*)g.Cpre + po) = pv;
        else if (g.pre_dtype == CFM_BF16) *(u32x2*)((u16*)g.Cpre + po) = (u32x2){pack2<BF16>(pv[0], pv[1]), pack2<BF16>(pv[2], pv[3])};
        else *(u32x2*)((u16*)g.Cpre + po) = (u32x2){pack2<F16>(pv[0], pv[1]), pack2<F16>(pv[2], pv[3])};
    };
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = m0 + wr * (BM / 2) + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int cb = n0 + wc * (BN / 2) + j * 16;
            const int col = cb + q4;
            const int ocol = glu ? (cb >> 1) + q4 : col;
            const bool live = g.res && row < g.M && col < g.N && !(glu && (j & 1));
            res_r[i][j] = live ? *(const f32x4*)(g.res + (int64_t)row * g.ldr + ocol) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (dact && row < g.M && col + 3 < g.N) {        // backward epilogue: the activation's argument shares the residual's registers
                const int64_t ao = (int64_t)row * g.ld_aux + col;
                if (g.aux_dtype == CFM_F32) {
                    res_r[i][j] = *(const f32x4*)((const float*)g.aux + ao);
                } else {
                    const u32x2 r2 = *(const u32x2*)((const u16*)g.aux + ao);
                    if (g.aux_dtype == CFM_BF16)
                        res_r[i][j] = (f32x4){BF16::to_f32((u16)(r2.x & 0xffffu)), BF16::to_f32((u16)(r2.x >> 16)), BF16::to_f32((u16)(r2.y & 0xffffu)), BF16::to_f32((u16)(r2.y >> 16))};
                    else
                        res_r[i][j] = (f32x4){F16::to_f32((u16)(r2.x & 0xffffu)), F16::to_f32((u16)(r2.x >> 16)), F16::to_f32((u16)(r2.y & 0xffffu)), F16::to_f32((u16)(r2.y >> 16))};
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = m0 + wr * (BM / 2) + i * 16 + (lane & 15);
        if (row >= g.M) continue;
        const bool keep = keep_r[i];
        const bool in_dead = !keep && g.mask_mode == 1;  // masked INPUT row: x.W = 0, bias/act still apply
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if (glu && (j & 1)) continue;
            const int cb = n0 + wc * (BN / 2) + j * 16;  // first GEMM column of this fragment
            const int col = cb + q4;
            if (col >= g.N) continue;
            f32x4 v = in_dead ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j];
            v += bias_r[j];
            if (g.Cpre && col + 3 < g.N) store_pre(row, col, v);
            int ocol = col;
            if (glu) {
                constexpr int JN_MAX = FN - 1;
                const int jn = j + 1 <= JN_MAX ? j + 1 : JN_MAX;  // FN is even; keeps the index static
                f32x4 gt = in_dead ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][jn];
                gt += bias_r[jn];
                if (g.Cpre && col + 16 + 3 < g.N) store_pre(row, col + 16, gt);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= sigmoidf_(gt[r]);
                ocol = (cb >> 1) + q4;
            } else if (g.act == CFM_ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = siluf_(v[r]);
            } else if (g.act == CFM_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (g.act == CFM_ACT_DSILU) {               // d silu(z)/dz = s (1 + z (1 - s)), s = sigmoid(z)
                if (g.drop.thresh) {                           // the gradient arrives at dropout(silu(z)): same mask as the forward's hidden
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = cfm_drop(g.drop, (unsigned)row * (unsigned)g.N + (unsigned)(col + r), v[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = res_r[i][j][r], sg = sigmoidf_(z);
                    v[r] *= g.alpha * sg * (1.f + z * (1.f - sg));
                }
            } else if (g.act == CFM_ACT_DRELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = res_r[i][j][r] > 0.f ? g.alpha * v[r] : 0.f;
            }
            if (g.drop.thresh && g.act != CFM_ACT_DSILU) {
                const unsigned nout = glu ? (unsigned)g.N >> 1 : (unsigned)g.N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned e = (unsigned)row * nout + (unsigned)(ocol + r);
                    v[r] = cfm_drop(g.drop, e, v[r]);
                    if (g.drop2.thresh) v[r] = cfm_drop(g.drop2, e, v[r]);
                }
            }
            if (!keep && g.mask_mode == 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g.res) v = res_r[i][j] + g.alpha * v;
            const int64_t o = (int64_t)row * g.ldc + ocol;
            if (col + 3 < g.N) {
                // one 16-byte (f32) / 8-byte (16 bit) store per lane.  When ldc is only a multiple of 2 the address is 8- / 4-byte aligned:
                // the under-aligned vector types keep it ONE global_store_dwordx4 / dwordx2 (global memory needs dword alignment only);
                // splitting into pairs cost 10 % on the joint's vocabulary projection (2224 -> 2028 us at ldc 5002 vs 5004)
                if (g.c_dtype == CFM_F32) {
                    *(f32x4_a4*)((float*)g.C + o) = v;
                } else if (g.c_dtype == CFM_BF16) {
                    *(u32x2_a4*)((u16*)g.C + o) = (u32x2){pack2<BF16>(v[0], v[1]), pack2<BF16>(v[2], v[3])};
                } else {
                    *(u32x2_a4*)((u16*)g.C + o) = (u32x2){pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3])};
                }
            } else if (col + 1 < g.N) {
                // N % 4 == 2 (a vocabulary of 5002 columns): the last column pair of the row
                if (g.c_dtype == CFM_F32) *(f32x2_a4*)((float*)g.C + o) = (f32x2){v[0], v[1]};
                else if (g.c_dtype == CFM_BF16) *(unsigned*)((u16*)g.C + o) = pack2<BF16>(v[0], v[1]);
                else *(unsigned*)((u16*)g.C + o) = pack2<F16>(v[0], v[1]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PERSISTENT variant for short-K products with many tiles (the transducer joint's vocabulary projection: M = B*T'*U = 163 k rows,
// K = 512, N = 5002 -> 51 k tiles of 8 K steps).  Measured on the kernel above (scripts/exp_gemm_k.py, 128 x 128 tile, two
// workgroups per CU): a workgroup lives 6.8 us + 1.19 us per K step -- launch, the first loads' latency and the epilogue are 40 % of
// its life at K = 512 and the other resident workgroup cannot use what it leaves idle.  Here 512 workgroups (two per CU) stay resident
// and walk the tile list b, b + 512, b + 1024, ... of the same XCD-aware order; the register prefetch ring runs ACROSS tile boundaries
// (the first K steps of the next tile are requested while the current tile still computes), the bias of a tile is requested at its
// first K step, and the epilogue is stores only.  16-bit operands, bias + SiLU/ReLU epilogues; same K order per output as the kernel
// above (bit-identical results).  Every lambda is force-inlined: left to the inliner's heuristics the state they capture (the
// prefetch ring, the accumulators) went to scratch memory and the kernel ran 10x slower.
template <typename HT, int BM, int BN, int BK>
__global__ __launch_bounds__(256, 2) void cfm_gemm_pers_kernel(const GemmArgs g) {
    constexpr int CPR = BK / 8, RPP = 256 / CPR, ACH = BM / RPP, WCH = BN / RPP, FM = BM / 32, FN = BN / 32, RPB = 16 / CPR;
    constexpr int A_PLANE = BM * CPR, W_PLANE = BN * CPR, BUF = A_PLANE + W_PLANE, PF = 3;
    __shared__ u32x4 smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kc = tid % CPR, rl = tid / CPR;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int per_group = 8 * tiles_n;
    const int total_vb = ((tiles_m + 7) / 8) * per_group;       // tile ids of the XCD-aware order, the last group padded
    const int G = gridDim.x;                                    // a multiple of 8: a workgroup's tiles all sit on its own XCD
    auto decode = [&](int vb, int& m0, int& n0) __attribute__((always_inline)) {
        const int grp = vb / per_group, rem = vb - grp * per_group;
        const int tm = grp * 8 + (rem & 7);
        m0 = tm * BM;
        n0 = (rem >> 3) * BN;
        return tm < tiles_m;
    };
    int ntile = ((int)blockIdx.x < total_vb) ? (total_vb - 1 - (int)blockIdx.x) / G + 1 : 0;
    if (ntile > 0) {                                            // padding ids exist only in the last group = this workgroup's last id
        int m0_, n0_;
        if (!decode((int)blockIdx.x + (ntile - 1) * G, m0_, n0_)) --ntile;
    }
    const int nkt = (g.K + BK - 1) / BK;
    const int V = ntile * nkt;                                  // K steps of this workgroup over all its tiles
    if (V == 0) return;

    auto lds_idx = [](int row, int c) __attribute__((always_inline)) { return row * CPR + (c ^ ((row / RPB) % CPR)); };

    // ---- load side: runs PF K steps ahead of the compute side, across tile boundaries -------------------------------------------
    int l_vb = blockIdx.x, l_kt = 0;
    unsigned a_off[ACH], w_off[WCH];
    auto l_setup = [&]() __attribute__((always_inline)) {
        int m0, n0;
        decode(l_vb, m0, n0);
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            int m = m0 + i * RPP + rl;
            m = m < g.M ? m : g.M - 1;
            a_off[i] = (unsigned)((int64_t)m * g.lda);
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            int n = n0 + j * RPP + rl;
            n = n < g.N ? n : g.N - 1;
            w_off[j] = (unsigned)n * (unsigned)g.K;
        }
    };
    l_setup();
    u32x4 ra[PF][ACH], rw[PF][WCH];
    auto gload = [&](auto slot_c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
        // UNCONDITIONAL loads (K % BK == 0 is required, rows are clamped, and past the last tile the ring re-reads clamped rows that
        // nobody stages): a predicated or branched-around load makes the compiler's s_waitcnt bookkeeping merge paths and fall back
        // to vmcnt(0) before every LDS store -- a prefetch distance of ONE K step instead of PF
        const int k0 = l_kt * BK + kc * 8;
#pragma unroll
        for (int i = 0; i < ACH; ++i) ra[S][i] = *(const u32x4*)((const u16*)g.A + (a_off[i] + (unsigned)k0));
#pragma unroll
        for (int j = 0; j < WCH; ++j) rw[S][j] = *(const u32x4*)(g.W + (w_off[j] + (unsigned)k0));
        if (++l_kt == nkt) {                                    // uniform
            l_kt = 0;
            l_vb += G;
            l_setup();
        }
    };
    auto lstore = [&](int buf, auto slot_c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
        u32x4* As = smem + buf * BUF;
        u32x4* Ws = As + A_PLANE;
#pragma unroll
        for (int i = 0; i < ACH; ++i) As[lds_idx(i * RPP + rl, kc)] = ra[S][i];
#pragma unroll
        for (int j = 0; j < WCH; ++j) Ws[lds_idx(j * RPP + rl, kc)] = rw[S][j];
    };

    // ---- compute side ---------------------------------------------------------------------------------------------------------------
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const u32x4* As = smem + buf * BUF;
        const u32x4* Ws = As + A_PLANE;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int c = kk * 4 + (lane >> 4);
            u32x4 af[FM], wf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = As[lds_idx(wr * (BM / 2) + i * 16 + (lane & 15), c)];
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = Ws[lds_idx(wc * (BN / 2) + j * 16 + (lane & 15), c)];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = HT::mfma(wf[j], af[i], acc[i][j]);
        }
    };
    int c_vb = blockIdx.x, c_kt = 0, c_m0 = 0, c_n0 = 0;
    const int q4 = (lane >> 4) * 4;
    f32x4 bias_r[FN];
    auto tile_begin = [&]() __attribute__((always_inline)) {                                   // the tile's coordinates and its bias, requested a whole K loop early
        decode(c_vb, c_m0, c_n0);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int col = c_n0 + wc * (BN / 2) + j * 16 + q4;
            if (g.bias && col + 3 < g.N) bias_r[j] = *(const f32x4*)(g.bias + col);
            else if (g.bias && col + 1 < g.N) bias_r[j] = (f32x4){g.bias[col], g.bias[col + 1], 0.f, 0.f};
            else bias_r[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto tile_end = [&]() __attribute__((always_inline)) {                                     // stores only
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = c_m0 + wr * (BM / 2) + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int col = c_n0 + wc * (BN / 2) + j * 16 + q4;
                f32x4 v = acc[i][j] + bias_r[j];
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (g.act == CFM_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = siluf_(v[r]);
                } else if (g.act == CFM_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (row >= g.M) continue;
                const int64_t o = (int64_t)row * g.ldc + col;
                if (col + 3 < g.N) {
                    if (g.c_dtype == CFM_F32) *(f32x4_a4*)((float*)g.C + o) = v;
                    else if (g.c_dtype == CFM_BF16) *(u32x2_a4*)((u16*)g.C + o) = (u32x2){pack2<BF16>(v[0], v[1]), pack2<BF16>(v[2], v[3])};
                    else *(u32x2_a4*)((u16*)g.C + o) = (u32x2){pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3])};
                } else if (col + 1 < g.N) {
                    if (g.c_dtype == CFM_F32) *(f32x2_a4*)((float*)g.C + o) = (f32x2){v[0], v[1]};
                    else if (g.c_dtype == CFM_BF16) *(unsigned*)((u16*)g.C + o) = pack2<BF16>(v[0], v[1]);
                    else *(unsigned*)((u16*)g.C + o) = pack2<F16>(v[0], v[1]);
                }
            }
        }
    };
    auto body = [&](int v, auto s) __attribute__((always_inline)) {                            // K step v of the workgroup lives in register slot v % PF
        lstore(v & 1, s);
        gload(s);
        __syncthreads();
        if (c_kt == 0) tile_begin();
        compute(v & 1);
        if (++c_kt == nkt) {
            tile_end();
            c_kt = 0;
            c_vb += G;
        }
    };
    gload(std::integral_constant<int, 0>{});
    gload(std::integral_constant<int, 1>{});
    gload(std::integral_constant<int, 2>{});
    for (int v0 = 0; v0 < V; v0 += PF) {
        body(v0, std::integral_constant<int, 0>{});
        if (v0 + 1 < V) body(v0 + 1, std::integral_constant<int, 1>{});
        if (v0 + 2 < V) body(v0 + 2, std::integral_constant<int, 2>{});
    }
}

// ---------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int CFM_PERSIST_GRID = 512;   // two resident workgroups on each of the 256 CUs (a multiple of 8: XCD round-robin)

template <typename HT, int BM, int BN, int BK, bool A_F32, bool SPLIT, bool CONV, int KG = 1>
int launch(const GemmArgs& a, hipStream_t s, const char* name) {
    const int tiles = (((a.M - a.m_begin + BM - 1) / BM + 7) / 8) * 8 * ((a.N + BN - 1) / BN);   // M tiles padded to a multiple of 8 (XCD groups)
    static const std::string nm = std::string(name) + "_" + std::to_string(BM) + "x" + std::to_string(BN) + (KG > 1 ? "_k" + std::to_string(KG) : std::string());
    const double rows = a.M - a.m_begin;
    const double flops = 2.0 * rows * (double)a.N * a.K;  // algorithmic (the 3 passes of SPLIT are not counted)
    const double bytes = rows * a.K * (A_F32 ? 4 : 2) + (double)a.N * a.K * 2 * (SPLIT ? 2 : 1) +
                         rows * a.N * (a.c_dtype == CFM_F32 ? 4 : 2);
    CfmProfScope prof(nm.c_str(), s, flops, bytes);
    CFM_LAUNCH((cfm_gemm_kernel<HT, BM, BN, BK, A_F32, SPLIT, CONV, KG>), dim3(tiles), dim3(256 * KG), 0, s, a);
    return cfm_launch_status(nm.c_str());
}

template <typename HT>
int launch_persistent(const GemmArgs& a, hipStream_t s, const char* name) {
    constexpr int BM = 128, BN = 128;
    const int total = (((a.M + BM - 1) / BM + 7) / 8) * 8 * ((a.N + BN - 1) / BN);
    const int grid = total < CFM_PERSIST_GRID ? total : CFM_PERSIST_GRID;
    static const std::string nm = std::string(name) + "_persistent_128x128";
    CfmProfScope prof(nm.c_str(), s, 2.0 * a.M * (double)a.N * a.K,
                      (double)a.M * a.K * 2 + (double)a.N * a.K * 2 + (double)a.M * a.N * (a.c_dtype == CFM_F32 ? 4 : 2));
    CFM_LAUNCH((cfm_gemm_pers_kernel<HT, BM, BN, 64>), dim3(grid), dim3(256), 0, s, a);
    return cfm_launch_status(nm.c_str());
}

template <typename HT, bool A_F32, bool SPLIT, bool CONV>
int pick_tile(const GemmArgs& a, int tile, hipStream_t s, const char* base) {
    constexpr int BK = SPLIT ? 32 : 64;
    if constexpr (!SPLIT && !A_F32 && !CONV) {
        // persistent workgroups: plain 16-bit products whose tiles are short (K <= 1024) and many (>= 8 per resident workgroup)
        const bool plain = !a.res && !a.mask && a.act != CFM_ACT_GLU && a.m_begin == 0 && !a.Cpre && a.act < CFM_ACT_DSILU && !a.drop.thresh;
        const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        if (tile == 7 || (tile == 0 && plain && a.K % BK == 0 && a.K <= 1024 && t128 >= 8L * CFM_PERSIST_GRID)) {
            if (!plain || a.K % BK) return cfm_fail(CFM_ERR_ARG, "cfm_gemm: the persistent tile takes bias / SiLU / ReLU epilogues only and K %% 64 == 0");
            return launch_persistent<HT>(a, s, base);
        }
    }
    const bool kgroups_ok = tile < 0;                   // CFM_TILE_AUTO_TRAIN: automatic choice, K-group tiles allowed (see cfm.h)
    if (tile < 0) tile = 0;
    if (tile == 0) {
        // fill the 256 CUs: prefer the biggest tile that still yields >= ~1 workgroup per CU
        const long t128 = (long)((a.M - a.m_begin + 127) / 128) * ((a.N + 127) / 128);
        const long t64x128 = (long)((a.M - a.m_begin + 63) / 64) * ((a.N + 127) / 128);
        tile = t128 >= 224 ? 1 : (t64x128 >= 224 ? 2 : 3);
        if constexpr (!SPLIT) {
            // short K and one to two rounds of 128 x 128 tiles (a feed-forward's first product at a training micro-batch: 304 tiles of 4 K steps,
            // the second round nearly empty): 32 x 64 tiles, measured 11.2 vs 14.9 us at M = 2 380, N = 2 048 (scripts/bench_gemm_tiles.py)
            if (tile == 1 && a.K <= 256 && t128 < 512) tile = 5;                   // narrow outputs (N = D) at training batch sizes: 64 x 64 tiles fill under half the CUs
            const long t64 = (long)((a.M - a.m_begin + 63) / 64) * ((a.N + 63) / 64), t32x64 = (long)((a.M - a.m_begin + 31) / 32) * ((a.N + 63) / 64);
            if (tile == 3 && t64 < 192 && t32x64 >= 96) tile = 5;
            // deep K and about one 64 x 128 workgroup per CU (the front-end Linear: M = 7 968, N = 256, K = 4 864 -> 250 workgroups of 76 K steps,
            // each alone on its CU and bound by its own load -> LDS -> MFMA chain): two 64 x 64 workgroups per CU overlap; measured 40.8 vs 44.8 us
            // there and 75.8 vs 83.2 us at M = 3 984, N = 512, K = 9 728 (config 4); 32 x 64 is slower again (L2 bytes per FLOP)
            if (tile == 2 && a.K >= 2048 && t64x128 < 384 && t64 >= 448) tile = 3;      // (round 3: also at K = 2 048 -- config 4's second feed-forward product, M = 3 984, N = 512: 19.7 vs 21.8 us)
            // one to two rounds of 128 x 128 tiles over a short K (config 4, M = 3 984, K = 512; scripts/bench_gemm_tiles.py): more, smaller tiles overlap
            // their prologues -- q|k|v (N = 1 536) 15.8 us on 128 x 64 against 19.7, pointwise-conv-1 (N = 1 024) 12.4 us on 64 x 64 against 15.9; at two
            // full rounds (N = 2 048: 512 tiles) 128 x 128 stays best.  Tiles without K groups keep the K order: the bits do not depend on the choice
            if (tile == 1 && a.K > 256 && a.K <= 512 && t128 < 448) tile = t128 >= 320 ? 4 : 3;
            // short K at a training micro-batch (M ~ 2 400: q|k|v 7.5 -> 6.3 us, pointwise-conv-1 6.5 -> 5.5 us): 32 x 64 tiles up to ~10 per CU
            if (tile != 5 && a.K <= 768 && t32x64 >= 96 && t32x64 <= 2560) tile = 5;
            // long K on few tiles (a feed-forward's second product and its input gradient at a training micro-batch: N = 256, K = 2 048):
            // K groups.  Measured at M = 2 380 (scripts/bench_gemm_tiles.py): 32 x 64 13.7 us, 32 x 64 k2 12.7, 64 x 64 14.8, 64 x 64 k2 11.7,
            // 64 x 64 k4 11.5, 32 x 64 k4 16.4, 64 x 128 k2 17.5 -- the gain is modest: the K chain is not what bounds these launches, and inside
            // the config-3 step it does not show at all (13.36 ms with, 13.38 ms without, same box, alternating runs)
            // -- only on request (tile -1, the training paths): a K-group tile regroups the K sum (half + half), so with it in the plain automatic
            // choice a batch of 2 would no longer reproduce the first two utterances of a batch of 16 bit for bit (the tile follows M)
            if constexpr (!A_F32 && !CONV) {
                if (kgroups_ok && (tile == 5 || tile == 3) && a.K >= 1024 && t64 <= 512) tile = 11;
                // an accumulation WINDOW of micro-batches (round 3: M ~ 2 800 .. 4 800 rows per launch): wide outputs over a short K have enough
                // 64 x 64 tiles for several per CU and half the L2 bytes per FLOP of 32 x 64 -- measured (scripts/bench_gemm_tiles.py) at
                // M = 3 400 / 4 400: N = 2 048: 13.9 / 17.8 us against 15.9 / 18.8; N = 768: 7.7 / 8.7 against 8.6 / 9.8; narrower outputs keep 32 x 64
                if (kgroups_ok && tile == 5 && a.K <= 256 && ((a.N >= 1024 && t64 >= 1400) || (a.N >= 768 && t64 >= 600))) tile = 3;
            }
        }
    }
    switch (tile) {
        case 1: return launch<HT, 128, 128, BK, A_F32, SPLIT, CONV>(a, s, base);
        case 2: return launch<HT, 64, 128, BK, A_F32, SPLIT, CONV>(a, s, base);
        case 3: return launch<HT, 64, 64, BK, A_F32, SPLIT, CONV>(a, s, base);
        case 4: return launch<HT, 128, 64, BK, A_F32, SPLIT, CONV>(a, s, base);
        case 5: if constexpr (!SPLIT) return launch<HT, 32, 64, BK, A_F32, SPLIT, CONV>(a, s, base); else break;
        case 6: if constexpr (!SPLIT) return launch<HT, 32, 128, BK, A_F32, SPLIT, CONV>(a, s, base); else break;
        case 9: if constexpr (!SPLIT && !A_F32 && !CONV) return launch<HT, 32, 64, BK, false, false, false, 2>(a, s, base); else break;
        case 10: if constexpr (!SPLIT && !A_F32 && !CONV) return launch<HT, 64, 64, BK, false, false, false, 4>(a, s, base); else break;
        case 11: if constexpr (!SPLIT && !A_F32 && !CONV) return launch<HT, 64, 64, BK, false, false, false, 2>(a, s, base); else break;
        default: break;
    }
    return cfm_fail(CFM_ERR_ARG, "cfm_gemm: unknown tile id %d", tile);
}

}  // namespace

extern "C" int cfm_gemm(const cfm_gemm_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d && d->A && d->W && d->C, "cfm_gemm: null pointer");
    CFM_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "cfm_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    CFM_CHECK_ARG(d->K % 8 == 0, "cfm_gemm: K=%d must be a multiple of 8", d->K);
    CFM_CHECK_ARG(d->N % 2 == 0, "cfm_gemm: N=%d must be a multiple of 2", d->N);
    const bool n4 = d->N % 4 == 0 && d->ldc % 4 == 0;    // otherwise: pair stores, and no epilogue that reads 4-column vectors
    CFM_CHECK_ARG(n4 || (!d->residual && d->act != CFM_ACT_GLU), "cfm_gemm: residual / GLU need N and ldc to be multiples of 4");
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_gemm: w_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(d->a_dtype == CFM_F32 || d->a_dtype == d->w_dtype, "cfm_gemm: a_dtype must be f32 or equal w_dtype");
    CFM_CHECK_ARG(d->c_dtype >= CFM_F32 && d->c_dtype <= CFM_F16, "cfm_gemm: bad c_dtype");
    CFM_CHECK_ARG(d->act >= CFM_ACT_NONE && d->act <= CFM_ACT_DRELU, "cfm_gemm: bad activation");
    const bool dact = d->act == CFM_ACT_DSILU || d->act == CFM_ACT_DRELU;
    CFM_CHECK_ARG(!dact || (d->aux && !d->residual && d->N % 4 == 0 && d->ld_aux % 4 == 0 && d->aux_dtype >= CFM_F32 && d->aux_dtype <= CFM_F16),
                  "cfm_gemm: a backward activation epilogue needs aux [M,N] (ld_aux %% 4 == 0), N %% 4 == 0 and no residual");
    CFM_CHECK_ARG(!d->C_pre || (d->N % 4 == 0 && d->ld_pre % 4 == 0 && d->pre_dtype >= CFM_F32 && d->pre_dtype <= CFM_F16 && !dact),
                  "cfm_gemm: C_pre needs N %% 4 == 0 and ld_pre %% 4 == 0");
    CFM_CHECK_ARG(d->mask_mode == 0 || d->mask_mode == 1, "cfm_gemm: bad mask_mode");
    CFM_CHECK_ARG(d->act != CFM_ACT_GLU || d->N % 32 == 0, "cfm_gemm: GLU needs N %% 32 == 0 (N=%d)", d->N);
    CFM_CHECK_ARG(d->ldc % 2 == 0, "cfm_gemm: ldc=%lld must be a multiple of 2", (long long)d->ldc);
    CFM_CHECK_ARG(!d->residual || (d->ldr % 4 == 0), "cfm_gemm: ldr must be a multiple of 4");
    const bool split = d->W_lo != nullptr;
    CFM_CHECK_ARG(!split || (d->a_dtype == CFM_F32 && d->w_dtype == CFM_BF16),
                  "cfm_gemm: split mode needs f32 activations and bf16 weight planes");
    const bool conv = d->conv_C > 0;
    if (conv) {
        CFM_CHECK_ARG(d->conv_C % 8 == 0 && d->K == 9 * d->conv_C, "cfm_gemm: conv needs C %% 8 == 0 and K == 9*C");
        CFM_CHECK_ARG(d->conv_T2 == (d->conv_T1 - 3) / 2 + 1 && d->conv_F2 == (d->conv_F1 - 3) / 2 + 1 && d->conv_T2 > 0 &&
                          d->conv_F2 > 0,
                      "cfm_gemm: conv output shape does not match a 3x3 stride-2 convolution");
        CFM_CHECK_ARG(d->M % (d->conv_T2 * d->conv_F2) == 0, "cfm_gemm: conv M must be B*T2*F2");
    } else {
        CFM_CHECK_ARG(d->lda % 8 == 0, "cfm_gemm: lda=%lld must be a multiple of 8", (long long)d->lda);
    }
    CFM_CHECK_ARG((int64_t)d->N * d->K < ((int64_t)1 << 31), "cfm_gemm: weight matrix too large for 32-bit element offsets");
    CFM_CHECK_ARG(conv ? (int64_t)(d->M / (d->conv_T2 * d->conv_F2)) * d->conv_T1 * d->conv_F1 * d->conv_C < ((int64_t)1 << 31)
                       : ((int64_t)(d->M - 1) * d->lda + d->K) < ((int64_t)1 << 31),
                  "cfm_gemm: activation operand too large for 32-bit element offsets");
    GemmArgs a;
    a.A = d->A; a.W = (const u16*)d->W; a.Wlo = (const u16*)d->W_lo; a.bias = d->bias; a.res = d->residual;
    a.mask = d->row_mask; a.C = d->C; a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr;
    CFM_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f && d->drop2_p >= 0.f && d->drop2_p < 1.f, "cfm_gemm: dropout probabilities must be in [0, 1)");
    CFM_CHECK_ARG((d->drop_p == 0.f && d->drop2_p == 0.f) || (d->N % 4 == 0 && (int64_t)d->M * d->N < ((int64_t)1 << 32)),
                  "cfm_gemm: dropout needs N %% 4 == 0 and fewer than 2^32 output elements");
    CFM_CHECK_ARG(d->drop2_p == 0.f || d->drop_p > 0.f, "cfm_gemm: drop2 is a second mask on top of drop");
    a.drop = cfm_make_drop(d->drop_p, d->drop_seed); a.drop2 = cfm_make_drop(d->drop2_p, d->drop2_seed);
    a.Cpre = d->C_pre; a.ld_pre = d->ld_pre; a.pre_dtype = d->pre_dtype; a.aux = dact ? d->aux : nullptr; a.ld_aux = d->ld_aux; a.aux_dtype = d->aux_dtype;
    a.M = d->M; a.N = d->N; a.K = d->K; a.m_begin = 0; a.c_dtype = d->c_dtype; a.act = d->act; a.alpha = d->alpha; a.mask_mode = d->mask_mode;
    a.convC = d->conv_C; a.T1 = d->conv_T1; a.F1 = d->conv_F1; a.T2 = d->conv_T2; a.F2 = d->conv_F2;
    hipStream_t s = (hipStream_t)stream;
    const bool a32 = d->a_dtype == CFM_F32;
    long head256 = 0;
    {   // 256 x 256 tile with LDS-DMA staging (gemm256.hip): tile id 8, or chosen by a two-line cost model when it can run
        const bool can256 = !split && !a32 && !d->residual && !d->row_mask && d->act != CFM_ACT_GLU && !d->C_pre && !dact && d->drop_p == 0.f;
        Gemm256Args b;
        b.A = (const u16*)d->A; b.W = (const u16*)d->W; b.bias = d->bias; b.C = d->C; b.lda = d->lda; b.ldc = d->ldc;
        b.M = d->M; b.N = d->N; b.K = d->K; b.c_dtype = d->c_dtype; b.act = d->act;
        b.convC = d->conv_C; b.T1 = d->conv_T1; b.F1 = d->conv_F1; b.T2 = d->conv_T2; b.F2 = d->conv_F2;
        if (d->tile == 8) {
            CFM_CHECK_ARG(can256, "cfm_gemm: the 256x256 tile takes 16-bit operands and bias / SiLU / ReLU epilogues only");
            return cfm_gemm256_launch(b, d->w_dtype == CFM_BF16, s);
        }
        if (d->tile == 0 && can256 && cfm_gemm256_eligible(b, d->w_dtype == CFM_BF16)) {
            // measured on MI355X (scripts/exp_gemm_k.py): a 256 x 256 tile takes ~7 us + 1.8 us per K step and 256 run at a time (whole
            // rounds: one workgroup per CU); a 128 x 128 tile ~6.8 us + 1.19 us per K step, 512 at a time, dealt continuously
            const double nk = d->K / 64.0;
            const long t256 = (long)((d->M + 255) / 256) * ((d->N + 255) / 256), t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
            const double c256 = (double)((t256 + 255) / 256) * (7.0 + 1.8 * nk);
            const double r128 = t128 / 512.0;
            const double c128 = (r128 < 1.0 ? 1.0 : r128) * (6.8 + 1.19 * nk);
            // third option: whole rounds on the 256 x 256 tile, the partial last round (e.g. 592 tiles on 256 CUs = 2.3 rounds) as
            // 128 x 128 tiles over the remaining rows -- two launches, split at an M-tile boundary
            const long tn256 = (d->N + 255) / 256;
            const long full_mt = ((t256 / 256) * 256) / tn256;                 // 256-row M tiles covered by the whole rounds
            const long rem_rows = d->M - full_mt * 256;
            double chyb = 1e30;
            if (full_mt >= 8 && rem_rows > 0) {
                const double rr = (double)((rem_rows + 127) / 128) * ((d->N + 127) / 128) / 512.0;
                chyb = (double)(t256 / 256) * (7.0 + 1.8 * nk) + (rr < 1.0 ? 1.0 : rr) * (6.8 + 1.19 * nk);
            }
            if (t256 >= 128 && chyb < c256 && chyb < c128) {
                head256 = full_mt * 256;                                      // rows [0, head256) here, the rest below on 128 x 128 tiles
                Gemm256Args h = b;
                h.M = (int)head256;
                if (int rc = cfm_gemm256_launch(h, d->w_dtype == CFM_BF16, s)) return rc;
            } else if (t256 >= 128 && c256 < c128) {
                return cfm_gemm256_launch(b, d->w_dtype == CFM_BF16, s);
            }
        }
    }
    a.m_begin = (int)head256;
    if (split) {
        return conv ? pick_tile<BF16, true, true, true>(a, d->tile, s, "gemm_conv_bf16x3")
                    : pick_tile<BF16, true, true, false>(a, d->tile, s, "gemm_bf16x3");
    }
    if (d->w_dtype == CFM_BF16) {
        if (conv) {
            CFM_CHECK_ARG(!a32, "cfm_gemm: conv mode reads a 16-bit image unless split");
            return pick_tile<BF16, false, false, true>(a, d->tile, s, "gemm_conv_bf16");
        }
        return a32 ? pick_tile<BF16, true, false, false>(a, d->tile, s, "gemm_bf16_a32")
                   : pick_tile<BF16, false, false, false>(a, d->tile, s, "gemm_bf16");
    }
    if (conv) {
        CFM_CHECK_ARG(!a32, "cfm_gemm: conv mode reads a 16-bit image unless split");
        return pick_tile<F16, false, false, true>(a, d->tile, s, "gemm_conv_f16");
    }
    return a32 ? pick_tile<F16, true, false, false>(a, d->tile, s, "gemm_f16_a32")
               : pick_tile<F16, false, false, false>(a, d->tile, s, "gemm_f16");
}
