// frontend.hip -- the two stride-2 convolutions of the front-end in ONE kernel (gfx950): Conv2d(1,C,3,2) + ReLU is recomputed inside the
// A-operand producer of Conv2d(C,C,3,2) + ReLU, so its [B,T1,F1,C] output (319 MB at config 2) is never written or read.
// Replaces convolution.py:60-63 for the 16-bit precision modes in eval mode.
//
// The second convolution is the implicit GEMM of gemm256.hip: rows m = (b, t2, f2), N = C output channels (one 256-wide tile covers
// all of N for C <= 256; C = 512 -- config 4 -- runs two workgroups per row tile, blockIdx.y = the 256-wide half of N, each producing the A tiles
// itself: the same +12.5 % of matrix work per workgroup), K = 9 taps x C input channels walked as 9 C / 64 tiles of one tap x 64 channels.  There the A tile of a K step
// (256 rows x 64 channels of conv1's output at tap (dt, df)) arrives by LDS-DMA from memory; here the workgroup COMPUTES it:
//   h1[b, 2 t2 + dt, 2 f2 + df, c] = relu( sum_{i,j} w1[c,i,j] * x[b, 4 t2 + 2 dt + i, 4 f2 + 2 df + j] + b1[c] )
// as one K = 32 MFMA per 16 rows x 16 channels with exactly the operands of cfm_conv1_mma_kernel (convmod.hip: taps of input row i in K
// group i, the f32 bias as a bf16 hi/lo pair times (1, 1) in K group 3, everything rounded to the mode's 16-bit type, zero
// accumulator), so every h1 value -- and with the same K order in the main loop every output -- is BIT-IDENTICAL to the two-kernel path.
// A 256 x 64 A tile costs 64 such MFMAs against the 512 of the K step it feeds (+12.5 % matrix work, 8 per wavefront), 32 v_max + 16
// v_cvt_pk and 4 ds_write_b128 per wavefront; it replaces 4 of the 8 LDS-DMA requests per thread and halves the kernel's L2 traffic.
// The x window a workgroup needs (a few dozen KB of f32) stays in L1 / L2; each lane fetches 3 floats per 16 rows and tap, one tap ahead.
//   * the channel a MFMA output row stands for is permuted (row 4q + r of fragment j = channel 16 q + 4 j + r of the 64-channel
//     slab) so that a lane ends up with 16 consecutive channels of its row: two 16-byte LDS writes, conflict-free under the tile's XOR
//     swizzle (position p of row r holds chunk p ^ ((r >> 1) & 7), as gemm256.hip);
//   * conv1's weight fragments for all C channels live in an 8 KB LDS table built once per workgroup;
//   * (measured: a third weight buffer with two K tiles in flight -- asm-issued LDS-DMA, counted waits -- does NOT help the 96-row tail tile: 48 vs
//     45 us.  Its K step is not waiting for the DMA but paying fixed per-step costs that do not shrink with the tile: the 32 KB weight tile
//     through the CU's vector-memory path, 8 + 6 fragment reads per wavefront, the barrier; ~1.25 us per step whatever is in flight.)
//   * FM = 16-row fragments per wavefront along M: tiles of 32 FM rows.  The host runs whole rounds of 256-row tiles and gives the last
//     partial round to a smaller FM so that it, too, is one workgroup per CU (the two-kernel path did the same with 128 x 128 tiles).
#include <string>
#include <type_traits>

#include "cfm_common.h"

namespace {

typedef unsigned int u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

#define CFM_INL __attribute__((always_inline))

struct Conv12Args {
    const float* x;        // [B, T, F] f32
    const float* w1;       // [9, C] f32, tap-major
    const float* b1;       // [C]
    const u16* W;          // [C, 9 C] 16-bit, K order (dt, df, ci)
    const float* b2;       // [C]
    u16* y;                // [M, C] 16-bit = [B, T2, F2, C]
    const float* cm_mean;  // [F] or null
    const float* cm_istd;  // [F] or null
    int T, F, C, T2, F2, M;
    int m_begin, m_end;    // this launch's rows
};

template <typename HT, int FM>
__global__ __launch_bounds__(512) void cfm_conv12_kernel(const Conv12Args g) {
    constexpr int BM = 32 * FM, BN = 256, BK = 64;
    constexpr int OPA = BM * (BK / 8), OPW = BN * (BK / 8);   // 16-byte chunks of the two operand tiles
    constexpr int BUF = OPA + OPW;
    constexpr int FN = 4;
    constexpr int NU = 2 * FM;                               // 16-row production units of an A tile
    constexpr int U = (NU + 7) / 8;                          // units per wavefront
    __shared__ u32x4 smem[2 * BUF + 1024];                   // two K tiles + conv1's weight fragments (u32x2 [<= 8 slabs][4][64 lanes])
    u32x2* const w1tab = (u32x2*)(smem + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, kg = lane >> 4;
    const int m0 = g.m_begin + blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;                          // this workgroup's output channels (C > 256: two halves)
    const int N = g.C, K = 9 * g.C;
    const int slabs = g.C >> 6;

    // ---- conv1's weight fragments (A operand: row l15 = channel 16 (l15 >> 2) + 4 j + (l15 & 3) of slab cs; K group kg = input row kg, 3 taps)
    for (int e = tid; e < slabs * 4 * 64; e += 512) {
        const int ln = e & 63, j = (e >> 6) & 3, cs = e >> 8;
        const int r = ln & 15, q = ln >> 4;
        const int ch = cs * 64 + 16 * (r >> 2) + 4 * j + (r & 3);
        u32x2 v = (u32x2){0u, 0u};
        if (q < 3) {
            const float a = g.w1[(3 * q) * g.C + ch], b = g.w1[(3 * q + 1) * g.C + ch], c = g.w1[(3 * q + 2) * g.C + ch];
            v.x = pack2<HT>(a, b);
            v.y = pack2<HT>(c, 0.f);
        } else {
            const float bv = g.b1[ch];
            const float hi = HT::to_f32(HT::from_f32(bv));
            v.x = pack2<HT>(hi, bv - hi);
        }
        w1tab[e] = v;
    }

    // ---- weight-tile staging (LDS-DMA, as gemm256.hip): request i of wavefront w fills rows 8 (8 i + w) .. + 7
    unsigned w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int n = n0 + r < N ? n0 + r : N - 1;
        w_off[i] = ((unsigned)n * (unsigned)K + c * 8) * 2u;
    }
    auto stageW = [&](int buf, int kt) CFM_INL {
        const unsigned kw = (unsigned)(kt * BK) * 2u;
        u32x4* const Ws = smem + buf * BUF + OPA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g.W + (w_off[i] + kw)),
                                             (__attribute__((address_space(3))) void*)(Ws + (i * 8 + wave) * 64), 16, 0, 0);
    };

    // ---- A-tile production.  Unit u = rows 16 u .. 16 u + 15 of the tile; lane (l15, kg) gathers input row 4 t2 + 2 dt + kg, columns 4 f2 + 2 df ..
    int xbase[U], fcol[U];
#pragma unroll
    for (int ui = 0; ui < U; ++ui) {
        const int u = wave + 8 * ui;
        int m = m0 + u * 16 + l15;
        m = m < g.M ? m : g.M - 1;
        const int per_b = g.T2 * g.F2;
        const int b = m / per_b;
        const int q = m - b * per_b;
        const int t2 = q / g.F2;
        const int f2 = q - t2 * g.F2;
        fcol[ui] = 4 * f2;
        xbase[ui] = (b * g.T + 4 * t2 + (kg < 3 ? kg : 0)) * g.F + 4 * f2;
    }
    float xraw[U][3];
    auto fetch = [&](int tap) CFM_INL {                      // tap = 3 dt + df
        const int dt = tap / 3, df = tap - 3 * dt;
#pragma unroll
        for (int ui = 0; ui < U; ++ui) {
            const float* xr = g.x + (xbase[ui] + 2 * dt * g.F + 2 * df);
            float v0 = xr[0], v1 = xr[1], v2 = xr[2];
            if (g.cm_mean) {
                const int f = fcol[ui] + 2 * df;
                v0 -= g.cm_mean[f]; v1 -= g.cm_mean[f + 1]; v2 -= g.cm_mean[f + 2];
                if (g.cm_istd) { v0 *= g.cm_istd[f]; v1 *= g.cm_istd[f + 1]; v2 *= g.cm_istd[f + 2]; }
            }
            xraw[ui][0] = v0; xraw[ui][1] = v1; xraw[ui][2] = v2;
        }
    };
    u32x4 xcur[U];
    auto convert = [&]() CFM_INL {
#pragma unroll
        for (int ui = 0; ui < U; ++ui) {
            xcur[ui] = (u32x4){0u, 0u, 0u, 0u};
            if (kg < 3) {
                xcur[ui].x = pack2<HT>(xraw[ui][0], xraw[ui][1]);
                xcur[ui].y = pack2<HT>(xraw[ui][2], 0.f);
            } else {
                xcur[ui].x = pack2<HT>(1.0f, 1.0f);          // times (bias_hi, bias_lo)
            }
        }
    };
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto produce = [&](int buf, int cs) CFM_INL {
        u32x4* const As = smem + buf * BUF;
        u32x4 wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x2 t = w1tab[(cs * 4 + j) * 64 + lane];
            wf[j] = (u32x4){t.x, t.y, 0u, 0u};
        }
#pragma unroll
        for (int ui = 0; ui < U; ++ui) {
            const int u = wave + 8 * ui;
            if (u < NU) {                                     // wave-uniform
                f32x4 a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = HT::mfma(wf[j], xcur[ui], zero4);
                    a[j].x = fmaxf(a[j].x, 0.f); a[j].y = fmaxf(a[j].y, 0.f); a[j].z = fmaxf(a[j].z, 0.f); a[j].w = fmaxf(a[j].w, 0.f);
                }
                const int row = u * 16 + l15;
                const int sw = (row >> 1) & 7;
                As[row * 8 + ((2 * kg) ^ sw)] = pack8<HT>(a[0], a[1]);
                As[row * 8 + ((2 * kg + 1) ^ sw)] = pack8<HT>(a[2], a[3]);
            }
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = zero4;
    const int sw = (lane >> 1) & 7;
    const int a_row = wr * (16 * FM) + l15, w_row = wc * 64 + l15;
    auto compute = [&](int buf) CFM_INL {
        const u32x4* const As = smem + buf * BUF;
        const u32x4* const Ws = As + OPA;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int c = (kk * 4 + kg) ^ sw;
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

    // ---- prologue: tap 0 in xcur, tap 1 on its way, K tile 0 in LDS
    fetch(0);
    stageW(0, 0);
    convert();
    fetch(1);
    __syncthreads();                                         // the weight-fragment table is complete
    produce(0, 0);
    const int q4 = kg * 4;
    f32x4 bias_r[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wc * 64 + j * 16 + q4;
        bias_r[j] = col + 3 < N ? *(const f32x4*)(g.b2 + col) : zero4;
    }
    __syncthreads();
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int cs = 0; cs < 8; ++cs) {
            if (cs < slabs) {                                // wave-uniform: C = 64 .. 512
                const int kt = tap * slabs + cs;
                const int cur = kt & 1;
                const bool last_slab = cs == slabs - 1;
                if (!(tap == 8 && last_slab)) {
                    stageW(cur ^ 1, kt + 1);
                    if (!last_slab) {
                        produce(cur ^ 1, cs + 1);
                    } else {
                        convert();                            // the tap fetched one tap ago
                        if (tap + 2 < 9) fetch(tap + 2);
                        produce(cur ^ 1, 0);
                    }
                }
                compute(cur);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: + bias, ReLU, 8-byte stores (a lane owns 4 consecutive output channels of a row)
    const int col0 = n0 + wc * 64 + q4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = m0 + wr * (16 * FM) + i * 16 + l15;
        u16* const rowp = g.y + ((int64_t)row * g.C + col0);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            f32x4 v = acc[i][j] + bias_r[j];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (row < g.m_end && col0 + j * 16 + 3 < N) *(u32x2_a4*)(rowp + j * 16) = (u32x2){pack2<HT>(v.x, v.y), pack2<HT>(v.z, v.w)};
        }
    }
}

template <typename HT, int FM>
int launch(const Conv12Args& a, hipStream_t s, const char* name) {
    const int rows = a.m_end - a.m_begin, BM = 32 * FM;
    const int tiles = (rows + BM - 1) / BM, nh = (a.C + 255) / 256;
    CfmProfScope prof(name, s, 2.0 * rows * (double)a.C * (9.0 * a.C + 32.0 * nh), (double)rows * a.C * 2 + 9.0 * a.C * a.C * 2);
    CFM_LAUNCH((cfm_conv12_kernel<HT, FM>), dim3(tiles, nh), dim3(512), 0, s, a);
    return cfm_launch_status(name);
}

template <typename HT>
int run(Conv12Args a, int cus, hipStream_t s, const char* nm_big, const char* nm_tail) {
    // whole rounds of 256-row tiles, then the remainder as ONE round of the smallest tile that fits it on the chip's CUs
    const int M = a.M;
    cus /= (a.C + 255) / 256;                                 // C = 512: two workgroups (halves of N) per row tile
    const int whole = (M / (256 * cus)) * cus;               // 256-row tiles in whole rounds
    int rest = M - whole * 256;
    if (rest > 0 && rest > 128 * cus) {                       // more than half a round left: it goes on 256-row tiles as well
        a.m_begin = 0; a.m_end = M;
        return launch<HT, 8>(a, s, nm_big);
    }
    if (whole > 0) {
        a.m_begin = 0; a.m_end = whole * 256;
        if (int rc = launch<HT, 8>(a, s, nm_big)) return rc;
    }
    if (rest > 0) {
        a.m_begin = whole * 256; a.m_end = M;
        if (rest <= 64 * cus) return launch<HT, 2>(a, s, nm_tail);
        if (rest <= 96 * cus) return launch<HT, 3>(a, s, nm_tail);
        return launch<HT, 4>(a, s, nm_tail);
    }
    return CFM_OK;
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}

}  // namespace

extern "C" int cfm_conv12_supported(int32_t C, int32_t y_dtype) { return C % 64 == 0 && C >= 64 && (C <= 256 || C == 512) && cfm_is16(y_dtype); }

extern "C" int cfm_conv12_relu(const float* x, const float* w1, const float* b1, const void* w2, const float* b2, void* y, int32_t y_dtype, int32_t B,
                               int32_t T, int32_t F, int32_t C, const float* cmvn_mean, const float* cmvn_istd, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && w1 && b1 && w2 && b2 && y, "cfm_conv12_relu: null pointer");
    CFM_CHECK_ARG(cfm_conv12_supported(C, y_dtype), "cfm_conv12_relu: C must be 64, 128, 192, 256 or 512 and y 16-bit (the type its products are rounded to)");
    const int T1 = (T - 3) / 2 + 1, F1 = (F - 3) / 2 + 1;
    const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
    CFM_CHECK_ARG(B > 0 && T >= 7 && F >= 7 && T2 >= 1 && F2 >= 1, "cfm_conv12_relu: the input is shorter than two 3x3 stride-2 windows");
    CFM_CHECK_ARG(!cmvn_istd || cmvn_mean, "cfm_conv12_relu: cmvn_istd without cmvn_mean");
    const int64_t M = (int64_t)B * T2 * F2;
    CFM_CHECK_ARG((int64_t)B * T * F < ((int64_t)1 << 31) && M * C < ((int64_t)1 << 31), "cfm_conv12_relu: 32-bit element offsets");
    Conv12Args a;
    a.x = x; a.w1 = w1; a.b1 = b1; a.W = (const u16*)w2; a.b2 = b2; a.y = (u16*)y; a.cm_mean = cmvn_mean; a.cm_istd = cmvn_istd;
    a.T = T; a.F = F; a.C = C; a.T2 = T2; a.F2 = F2; a.M = (int)M; a.m_begin = 0; a.m_end = (int)M;
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == CFM_BF16) return run<BF16>(a, num_cus(), s, "conv12_bf16_256", "conv12_bf16_tail");
    return run<F16>(a, num_cus(), s, "conv12_f16_256", "conv12_f16_tail");
}
