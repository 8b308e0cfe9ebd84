// gemm256.hip -- C = act(A[M,K] . W[N,K]^T + bias) on 256 x 256 tiles with LDS-DMA staging (gfx950).
//
// The 128 x 128 kernel of gemm.hip is bound by the bytes it moves per MFMA, not by the MFMA pipe: per K step a CU spends ~512 clk
// of its vector-memory path on the global loads and ~768 clk of LDS on the register->LDS writes and the fragment reads, and those
// add up (measured 1 420 clk per K step against 512 clk of MFMA, scripts/exp_gemm_k.py).  This kernel moves fewer bytes per MFMA -- a
// wavefront owns 128 x 64 of the output, 0.375 KB of fragment reads per MFMA instead of 0.5, and a 256 x 256 tile loads half the
// global bytes per MFMA of a 128 x 128 one -- and drops the register round trip of the staging altogether:
//   * 512 threads = 8 wavefronts as 2 (M) x 4 (N); tile 256 x 256, BK = 64; one workgroup per CU (128 KB of LDS, two K tiles);
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no VGPR destination, no ds_write); the LDS image of an operand
//     tile is lane-linear ([256 rows][8 x 16 B]), so the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is
//     applied to the per-lane SOURCE address and again on the read (position p of row r holds chunk p ^ ((r >> 1) & 7));
//   * two-phase loop: request K tile t+1, compute K tile t, then vmcnt(0) + barrier -- one barrier per K tile, the whole next tile
//     in flight behind 128 MFMAs per wavefront;
//   * swapped MFMA roles as everywhere in this library (weights are the A operand): a lane ends up with 4 consecutive output
//     columns of one row, stored as one 16- / 8-byte access;
//   * the activation operand may be the implicit im2col view of a channels-last image (3 x 3 / stride 2, C % 64 == 0): the
//     gather is just a different per-lane source offset.
// 16-bit operands, K % 64 == 0, bias + ReLU / SiLU epilogues.  Same K order per output element as gemm.hip, so results are
// bit-identical to its kernels.
#include <string>
#include <type_traits>

#include "cfm_common.h"
#include "gemm256.h"

namespace {

typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned int u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

#define CFM_INL __attribute__((always_inline))

template <typename HT, bool CONV>
__global__ __launch_bounds__(512) void cfm_gemm256_kernel(const Gemm256Args g) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int OP = BM * (BK / 8);                      // 16-byte chunks of one operand tile (32 KB)
    constexpr int BUF = 2 * OP;                            // activation tile + weight tile
    constexpr int FM = 8, FN = 4;                          // 16-row / 16-column fragments of a wavefront's 128 x 64 block
    __shared__ u32x4 smem[2 * BUF];                        // the ONLY LDS object: 2 K tiles x 64 KB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // XCD-aware tile order (as gemm.hip): the N tiles of one M tile get ids 8 apart and share an L2
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int per_group = 8 * tiles_n;
    const int grp = blockIdx.x / per_group, rem = blockIdx.x - grp * per_group;
    const int tile_m = grp * 8 + (rem & 7);
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = (rem >> 3) * BN;

    // ---- staging: 4 + 4 LDS-DMA requests per thread per K tile.  Request i of wavefront w fills the 1 KB block (i*8 + w) of an operand
    //      tile = its rows 8*(i*8+w) .. +7; lane L lands at position p = L & 7 of row r = 8*blk + (L >> 3) and fetches chunk p ^ ((r>>1)&7)
    unsigned a_off[4], w_off[4];
    auto setup = [&](int tm0, int tn0) CFM_INL {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (i * 8 + wave) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int m = tm0 + r;
            m = m < g.M ? m : g.M - 1;
            if constexpr (CONV) {
                const int per_b = g.T2 * g.F2;
                const int b = m / per_b;
                const int q = m - b * per_b;
                const int t2 = q / g.F2;
                const int f2 = q - t2 * g.F2;
                a_off[i] = ((unsigned)(((b * g.T1 + 2 * t2) * g.F1 + 2 * f2) * g.convC) + c * 8) * 2u;
            } else {
                a_off[i] = ((unsigned)((int64_t)m * g.lda) + c * 8) * 2u;
            }
            int n = tn0 + r;
            n = n < g.N ? n : g.N - 1;
            w_off[i] = ((unsigned)n * (unsigned)g.K + c * 8) * 2u;
        }
    };
    auto stage = [&](int buf, int kt) CFM_INL {
        // BYTE offsets as 32-bit values on top of the uniform base pointers (the host checks both operands are < 4 GB): the request is
        // then `global_load_lds_dwordx4 voff, s[base]`; element offsets would be widened to 64-bit VGPR pairs (8 pairs, and they spilled)
        unsigned ka = (unsigned)(kt * BK) * 2u;
        if constexpr (CONV) {                              // a K tile lies inside one tap (convC % 64 == 0)
            const int tap = (kt * BK) / g.convC;
            const int ci = kt * BK - tap * g.convC;
            const int k3 = tap / 3, f3 = tap - 3 * k3;
            ka = (unsigned)((k3 * g.F1 + f3) * g.convC + ci) * 2u;
        }
        const unsigned kw = (unsigned)(kt * BK) * 2u;
        u32x4* const As = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g.A + (a_off[i] + ka)),
                                             (__attribute__((address_space(3))) void*)(As + (i * 8 + wave) * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g.W + (w_off[i] + kw)),
                                             (__attribute__((address_space(3))) void*)(As + OP + (i * 8 + wave) * 64), 16, 0, 0);
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int sw = (lane >> 1) & 7;                         // ((row >> 1) & 7) of this lane's fragment row (fragment bases are multiples of 16)
    const int a_row = wr * 128 + (lane & 15), w_row = wc * 64 + (lane & 15);
    auto compute = [&](int buf) CFM_INL {
        const u32x4* const As = smem + buf * BUF;
        const u32x4* const Ws = As + OP;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int c = (kk * 4 + (lane >> 4)) ^ sw;
            u32x4 af[FM], wf[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[j] = Ws[(w_row + j * 16) * 8 + c];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = As[(a_row + i * 16) * 8 + c];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = HT::mfma(wf[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
    };
    const int q4 = (lane >> 4) * 4;
    f32x4 bias_r[FN];
    auto load_bias = [&](int tn0) CFM_INL {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int col = tn0 + wc * 64 + j * 16 + q4;
            if (g.bias && col + 3 < g.N) bias_r[j] = *(const f32x4*)(g.bias + col);
            else if (g.bias && col + 1 < g.N) bias_r[j] = (f32x4){g.bias[col], g.bias[col + 1], 0.f, 0.f};
            else bias_r[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // Epilogue: one 64-bit row pointer per fragment row and constant column offsets, one dtype branch around the whole tile.
    auto store_tile = [&](int tm0, int tn0, auto dt_c) CFM_INL {
        constexpr int DT = decltype(dt_c)::value;
        constexpr int ESZ = DT == CFM_F32 ? 4 : 2;
        const int col0 = tn0 + wc * 64 + q4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = tm0 + wr * 128 + i * 16 + (lane & 15);
            char* const rowp = (char*)g.C + ((int64_t)row * g.ldc + col0) * ESZ;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
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
                const int col = col0 + j * 16;
                char* const p = rowp + j * 16 * ESZ;
                if (col + 3 < g.N) {
                    if constexpr (DT == CFM_F32) *(f32x4_a4*)p = v;
                    else if constexpr (DT == CFM_BF16) *(u32x2_a4*)p = (u32x2){pack2<BF16>(v[0], v[1]), pack2<BF16>(v[2], v[3])};
                    else *(u32x2_a4*)p = (u32x2){pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3])};
                } else if (col + 1 < g.N) {
                    if constexpr (DT == CFM_F32) *(f32x2_a4*)p = (f32x2){v[0], v[1]};
                    else if constexpr (DT == CFM_BF16) *(unsigned*)p = pack2<BF16>(v[0], v[1]);
                    else *(unsigned*)p = pack2<F16>(v[0], v[1]);
                }
            }
        }
    };
    auto epilogue = [&](int tm0, int tn0) CFM_INL {
        if (g.c_dtype == CFM_F32) store_tile(tm0, tn0, std::integral_constant<int, CFM_F32>{});
        else if (g.c_dtype == CFM_BF16) store_tile(tm0, tn0, std::integral_constant<int, CFM_BF16>{});
        else store_tile(tm0, tn0, std::integral_constant<int, CFM_F16>{});
    };

    const int nkt = g.K / BK;
    setup(m0, n0);
    stage(0, 0);
    load_bias(n0);                                          // requested behind the first K tile, used after the loop
    __syncthreads();                                        // vmcnt(0) + barrier: K tile 0 is in LDS
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);      // the buffer read one step ago; every wavefront is past that step's barrier
        compute(kt & 1);
        __syncthreads();                                    // retires the DMA of K tile kt+1 and this step's reads
    }
    epilogue(m0, n0);
    // Two deeper-pipelined forms were built, verified bit-identical and measured SLOWER; neither is kept:
    //  * BK = 32 with FOUR LDS buffers, the DMA of K step t+3 requested while step t computes, a counted `s_waitcnt vmcnt(8)` and a raw
    //    s_barrier (two K steps stay in flight across every barrier): 1 752 vs 1 582 us at K = 2 048, 239 vs 227 us on the front-end
    //    convolution -- twice the barriers per MFMA cost more than the longer prefetch distance gains, i.e. the step is not waiting for
    //    memory but for its own lock-step of LDS reads and MFMAs (all 8 wavefronts read, then all multiply);
    //  * a PERSISTENT form (one workgroup per CU walking the tile list, the next tile's first K tile requested during the last K step,
    //    epilogue inside the loop): 618 vs 588 us at K = 512, 1 409 vs 1 307 us on the joint's projection, 238 vs 227 us on the front-end
    //    convolution -- the in-loop epilogue costs ~20 spilled registers and its stores sit in front of the vmcnt(0) of the next barrier, which is worse than a fresh workgroup's launch + first-tile latency.
    //  * a PING-PONG rotation: the wavefronts of row group 1 (one per SIMD, next to one of group 0: scripts/probe_simd_map.hip) run their
    //    loop rotated by one MFMA cluster -- the cluster of a step's second K slice is issued at the top of the next step from fragments
    //    already in registers -- so that one wavefront of a SIMD reads while the other multiplies.  Bit-identical; +8 % at K = 4 096 on
    //    random data (1 253 vs 1 145 TFLOP/s) but, A/B in one session, 158 vs 156 us on the front-end convolution and 1 328 vs 1 336 us on the
    //    joint's projection, with 9-13 spilled registers.  Knock-outs of the lock-step loop at K = 4 096 (us per K step per tile): MFMA +
    //    barrier alone 1.45 (the MFMA roofline at the ~1.9 GHz the chip holds under this load, plus 0.33 of barrier and loop), + DMA 1.88, +
    //    fragment reads 2.06: the pieces add up instead of overlapping, and the rotation recovers only 0.1 of the 0.6.
    // What is left on the table is the fully phased schedule of the programming guide (half-tile staging, per-phase counted waits, two
    // barriers per phase), which it measures at 1.3-1.5 PFLOP/s on 8 192-cubed products.
}

template <typename HT, bool CONV>
int launch256(const Gemm256Args& a, hipStream_t s, const char* name) {
    const int tiles = (((a.M + 255) / 256 + 7) / 8) * 8 * ((a.N + 255) / 256);
    static const std::string nm = std::string(name) + "_256x256";
    CfmProfScope prof(nm.c_str(), s, 2.0 * a.M * (double)a.N * a.K,
                      (double)a.M * a.K * 2 + (double)a.N * a.K * 2 + (double)a.M * a.N * (a.c_dtype == CFM_F32 ? 4 : 2));
    CFM_LAUNCH((cfm_gemm256_kernel<HT, CONV>), dim3(tiles), dim3(512), 0, s, a);
    return cfm_launch_status(nm.c_str());
}

}  // namespace

bool cfm_gemm256_eligible(const Gemm256Args& a, bool w_bf16) {
    (void)w_bf16;
    if (a.K % 64 != 0 || a.K < 64) return false;
    const int64_t a_elems = a.convC > 0 ? (int64_t)(a.M / (a.T2 * a.F2)) * a.T1 * a.F1 * a.convC : (int64_t)(a.M - 1) * a.lda + a.K;
    if (a_elems * 2 >= ((int64_t)1 << 32) || (int64_t)a.N * a.K * 2 >= ((int64_t)1 << 32)) return false;   // 32-bit byte offsets
    if (a.convC > 0 && a.convC % 64 != 0) return false;
    return true;
}

int cfm_gemm256_launch(const Gemm256Args& a, bool w_bf16, hipStream_t s) {
    if (!cfm_gemm256_eligible(a, w_bf16)) return cfm_fail(CFM_ERR_ARG, "cfm_gemm: the 256x256 tile needs K %% 64 == 0 (and conv C %% 64 == 0)");
    if (a.convC > 0) return w_bf16 ? launch256<BF16, true>(a, s, "gemm_conv_bf16") : launch256<F16, true>(a, s, "gemm_conv_f16");
    return w_bf16 ? launch256<BF16, false>(a, s, "gemm_bf16") : launch256<F16, false>(a, s, "gemm_f16");
}
