// ffnsplit.hip -- the feed-forward of a conformer block for FEW rows (gfx950): the hidden dimension split across workgroups.
//
// The row chains (rowchain.hip) give every 32-row tile to one workgroup, which streams ALL of W1 and W2 (2 MB at D = 256, FF = 2048)
// through its CU's vector-memory path: ~25 us per feed-forward however few rows there are.  That is the right shape when the tiles fill
// the chip (config 2: 249 tiles); a streaming step of 64 streams x 16 frames has 32 tiles and a single stream has one (BASELINE config 5,
// SURVEY 8 row S).  Here a tile's feed-forward is split over FF / 256 workgroups instead:
//
//     workgroup (tile t, slice g):   P_g[32 rows, D] = act( LN(x_t) . W1[slice g]^T + b1[slice g] ) . W2[:, slice g]^T        (256 KB of weights)
//
// and the G partial slabs are summed -- with bias, alpha, the residual and the following LayerNorm(s) -- by the INPUT STAGE of whichever
// launch of this file runs next: a launch boundary, so no protocol between workgroups and a fixed summation order (g = 0, 1, ...:
// bitwise reproducible).  Three modes share the input stage and the first product:
//     ROWS   rows only:  x = res + alpha (sum_g P_g + b2) -> [LN1] -> rows_out -> [LN2] -> rows2_out              (norm_final, after_norm)
//     PROJ   rows, then  out16[:, slice g] = LN(rows) . W[slice g]^T + b                                          (norm_mha + fused q|k|v)
//     FFN    rows, then  P_g as above                                                                              (both feed-forwards)
// 8 wavefronts; wavefront w owns 32 of the slice's 256 columns in both products; weights come fragment-major straight from memory
// into MFMA A operands (one wavefront-load = one fragment, as rowchain.hip), the LayerNorm tile and the hidden tile pass through LDS
// (16 KB each, 16-byte chunks XOR-swizzled by the row: conflict-free ds_read_b128 of 16 rows).  All 16 weight fragments of a
// product are requested before its first MFMA, and product 2's before the hidden tile is written: a workgroup is a short latency
// chain (rows -> LN -> 32 MFMAs -> SiLU -> 32 MFMAs), so everything that does not depend on the chain is in flight from the start.
// D = 256 only (the config-2 / config-5 width).
// (Measured and dropped: letting a block's reduce + norm_final ride in the NEXT block's macaron feed-forward launch -- one launch less per block.
// All 8 slices then repeat the reduce (8 slabs x 32 KB each) in front of their LayerNorm: that launch went 8.9 -> 16.8 us against the 5.7 us of
// the rows launch it saved; config-5 step 1.011 -> 1.052 ms.  The reduce belongs where one workgroup per tile does it: modes 0 and 1.)
#include <string>
#include <type_traits>

#include "cfm_common.h"

namespace {

constexpr int FS_D = 256, FS_RBM = 32, FS_NW = 8, FS_SL = 256, FS_KS = FS_D / 32;

struct FfnSplitArgs {
    const float* x;            // [M, D] rows, or the residual of the reduce
    const float* psum;         // null, or [g_in][M][D] partial slabs to add
    const float* pb2;          // reduce: second bias of the product the slabs belong to
    const float *ln1_g, *ln1_b;  // optional LayerNorm on the reduced rows
    const float *ln2_g, *ln2_b;  // optional second LayerNorm for rows2_out
    float* rows_out;           // optional (written by slice 0)
    float* rows2_out;          // optional (written by slice 0)
    const float *ln_g, *ln_b;  // the LayerNorm feeding product 1
    const u16* w1;             // fragment-major [n1 / 16][D / 32][64][8]
    const float* b1;
    const u16* w2;             // fragment-major [D / 16][FF / 32][64][8]
    float* psum_out;           // FFN: [n1 / 256][M][D]
    u16* out16;                // PROJ: [M, ldo]
    int64_t ldo;
    int M, g_in, act, FF;
    float palpha, eps;
    // PROJ as the fused q|k|v projection of a streaming step: columns [D, 3D) also go, rounded to the 16-bit type and widened again (the very
    // values cfm_kv_ring_write would copy), into the K/V ring f32 [B, H, ring_T, 2 dk] at slot (offsets[b] + t) mod ring_T
    float* ring;
    const int* ring_off;
    int ring_T, ring_H, ring_dk, ring_Tq;
};

__device__ __forceinline__ int fs_swz(int row, int chunk) { return row * 32 + (chunk ^ (row & 15)); }   // 16-byte chunk index in a [32][256] 16-bit tile

template <typename HT, int MODE>   // 0 rows, 1 proj, 2 ffn
__global__ __launch_bounds__(FS_NW * 64) void cfm_ffn_split_kernel(const FfnSplitArgs a) {
    __shared__ u32x4 xn_s[FS_RBM * 32];
    __shared__ u32x4 h_s[MODE == 2 ? FS_RBM * 32 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int row0 = blockIdx.x * FS_RBM;
    const int slice = blockIdx.y;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- product 1's weight fragments: requested first (they depend on nothing)
    u32x4 wf1[2][FS_KS];
    f32x4 bias1[2];
    if constexpr (MODE > 0) {
        const int nfrag0 = (slice * FS_SL + wave * 32) / 16;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int kk = 0; kk < FS_KS; ++kk) wf1[f][kk] = *(const u32x4*)(a.w1 + ((int64_t)((nfrag0 + f) * FS_KS + kk) * 64 + lane) * 8);
            bias1[f] = *(const f32x4*)(a.b1 + slice * FS_SL + wave * 32 + f * 16 + 4 * g4);
        }
    }

    // ---- rows: wavefront w owns rows 4 w .. 4 w + 3 of the tile, lane owns columns 4 lane .. + 3
    constexpr int RPW = FS_RBM / FS_NW;
    f32x4 v[RPW];
    int grow[RPW];
    bool rok[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = row0 + wave * RPW + rr;
        rok[rr] = r < a.M;
        grow[rr] = rok[rr] ? r : a.M - 1;
        v[rr] = *(const f32x4*)(a.x + (int64_t)grow[rr] * FS_D + 4 * lane);
    }
    if (a.psum) {                                           // uniform
        f32x4 acc[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) acc[rr] = *(const f32x4*)(a.psum + (int64_t)grow[rr] * FS_D + 4 * lane);
        for (int g = 1; g < a.g_in; ++g)
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) acc[rr] += *(const f32x4*)(a.psum + ((int64_t)g * a.M + grow[rr]) * FS_D + 4 * lane);
        const f32x4 b2 = *(const f32x4*)(a.pb2 + 4 * lane);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) v[rr] = v[rr] + a.palpha * (acc[rr] + b2);
    }
    auto layernorm = [&](const float* gp, const float* bp) __attribute__((always_inline)) {
        const f32x4 gam = *(const f32x4*)(gp + 4 * lane), bet = *(const f32x4*)(bp + 4 * lane);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const float mean = wave_sum((v[rr].x + v[rr].y) + (v[rr].z + v[rr].w)) * (1.0f / FS_D);
            const f32x4 d = v[rr] - mean;
            const float rstd = __builtin_amdgcn_rsqf(wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / FS_D) + a.eps);
            v[rr] = d * rstd * gam + bet;
        }
    };
    if (a.ln1_g) layernorm(a.ln1_g, a.ln1_b);
    if (a.rows_out && slice == 0) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
            if (rok[rr]) *(f32x4*)(a.rows_out + (int64_t)grow[rr] * FS_D + 4 * lane) = v[rr];
    }
    if constexpr (MODE == 0) {
        if (a.rows2_out) {
            layernorm(a.ln2_g, a.ln2_b);
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
                if (rok[rr]) *(f32x4*)(a.rows2_out + (int64_t)grow[rr] * FS_D + 4 * lane) = v[rr];
        }
        return;
    } else {
        layernorm(a.ln_g, a.ln_b);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {                  // 8 bytes per lane: half of chunk (lane >> 1)
            const int r = wave * RPW + rr;
            u32x2* const dst = (u32x2*)(xn_s + fs_swz(r, lane >> 1)) + (lane & 1);
            *dst = (u32x2){pack2<HT>(v[rr].x, v[rr].y), pack2<HT>(v[rr].z, v[rr].w)};
        }
        // ---- product 2's weight fragments (FFN): requested before the barrier
        u32x4 wf2[2][FS_KS];
        if constexpr (MODE == 2) {
            const int ks2 = a.FF / 32;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int kk = 0; kk < FS_KS; ++kk)
                    wf2[f][kk] = *(const u32x4*)(a.w2 + ((int64_t)((wave * 2 + f) * ks2 + slice * FS_KS + kk) * 64 + lane) * 8);
        }
        __syncthreads();

        // ---- product 1: acc[f][mf] = W1 fragment (16 columns of the slice) x LN tile fragment (16 rows), K = D
        f32x4 acc[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[f][0] = acc[f][1] = zero4;
#pragma unroll
        for (int kk = 0; kk < FS_KS; ++kk) {
            const u32x4 x0 = xn_s[fs_swz(l15, kk * 4 + g4)], x1 = xn_s[fs_swz(16 + l15, kk * 4 + g4)];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                acc[f][0] = HT::mfma(wf1[f][kk], x0, acc[f][0]);
                acc[f][1] = HT::mfma(wf1[f][kk], x1, acc[f][1]);
            }
        }
        // a lane holds 4 consecutive columns (slice column wave*32 + f*16 + 4 g4 ..) of row mf*16 + l15
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                f32x4 h = acc[f][mf] + bias1[f];
                if (a.act == CFM_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = siluf_(h[r]);
                } else if (a.act == CFM_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
                }
                const u32x2 pk = (u32x2){pack2<HT>(h.x, h.y), pack2<HT>(h.z, h.w)};
                const int row = mf * 16 + l15, col = wave * 32 + f * 16 + 4 * g4;
                if constexpr (MODE == 1) {
                    if (row0 + row < a.M) {
                        *(u32x2*)(a.out16 + (int64_t)(row0 + row) * a.ldo + slice * FS_SL + col) = pk;
                        if (a.ring && slice > 0) {                      // uniform: slice 1 = keys, slice 2 = values (D = 256 = one slice each)
                            const int m = row0 + row, b = m / a.ring_Tq, t = m - b * a.ring_Tq;
                            const int slot = (a.ring_off[b] + t) % a.ring_T;
                            const int hh = col / a.ring_dk, e = col - hh * a.ring_dk + (slice == 2 ? a.ring_dk : 0);
                            const f32x4 rv = (f32x4){HT::to_f32((u16)(pk.x & 0xffffu)), HT::to_f32((u16)(pk.x >> 16)), HT::to_f32((u16)(pk.y & 0xffffu)), HT::to_f32((u16)(pk.y >> 16))};
                            *(f32x4*)(a.ring + (((int64_t)b * a.ring_H + hh) * a.ring_T + slot) * (2 * a.ring_dk) + e) = rv;
                        }
                    }
                } else {
                    *((u32x2*)(h_s + fs_swz(row, col >> 3)) + ((col >> 2) & 1)) = pk;
                }
            }
        if constexpr (MODE == 2) {
            __syncthreads();
            // ---- product 2: P[rows, 32 output columns of this wavefront] over the slice's 256 hidden columns
            f32x4 p[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f) p[f][0] = p[f][1] = zero4;
#pragma unroll
            for (int kk = 0; kk < FS_KS; ++kk) {
                const u32x4 h0 = h_s[fs_swz(l15, kk * 4 + g4)], h1 = h_s[fs_swz(16 + l15, kk * 4 + g4)];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    p[f][0] = HT::mfma(wf2[f][kk], h0, p[f][0]);
                    p[f][1] = HT::mfma(wf2[f][kk], h1, p[f][1]);
                }
            }
            float* const slab = a.psum_out + (int64_t)slice * a.M * FS_D;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const int row = row0 + mf * 16 + l15;
                    if (row < a.M) *(f32x4*)(slab + (int64_t)row * FS_D + wave * 32 + f * 16 + 4 * g4) = p[f][mf];
                }
        }
    }
}

template <typename HT>
int launch_mode(const FfnSplitArgs& a, int mode, int slices, hipStream_t s) {
    const dim3 grid((unsigned)((a.M + FS_RBM - 1) / FS_RBM), (unsigned)slices), block(FS_NW * 64);
    const bool bf = std::is_same<HT, BF16>::value;
    if (mode == 0) {
        CfmProfScope prof(bf ? "ffnsplit_rows_bf16" : "ffnsplit_rows_f16", s, 0.0, (double)a.M * FS_D * 4 * (2 + a.g_in));
        CFM_LAUNCH((cfm_ffn_split_kernel<HT, 0>), grid, block, 0, s, a);
    } else if (mode == 1) {
        CfmProfScope prof(bf ? "ffnsplit_proj_bf16" : "ffnsplit_proj_f16", s, 2.0 * a.M * FS_D * (double)slices * FS_SL, (double)a.M * FS_D * 4 * (1 + a.g_in) * slices);
        CFM_LAUNCH((cfm_ffn_split_kernel<HT, 1>), grid, block, 0, s, a);
    } else {
        CfmProfScope prof(bf ? "ffnsplit_ffn_bf16" : "ffnsplit_ffn_f16", s, 4.0 * a.M * FS_D * (double)slices * FS_SL, (double)a.M * FS_D * 4 * (2 + a.g_in) * slices);
        CFM_LAUNCH((cfm_ffn_split_kernel<HT, 2>), grid, block, 0, s, a);
    }
    return cfm_launch_status("cfm_ffn_split");
}

}  // namespace

extern "C" int cfm_ffn_split_supported(int32_t D, int32_t FF) { return D == FS_D && FF > 0 && FF % FS_SL == 0; }

extern "C" int cfm_ffn_split(const cfm_ffn_split_desc* d, cfm_stream_t stream) {
    CFM_CHECK_ARG(d && d->x, "cfm_ffn_split: null pointer");
    CFM_CHECK_ARG(d->D == FS_D && d->M > 0, "cfm_ffn_split: D = %d has no instance (256), M = %d", d->D, d->M);
    CFM_CHECK_ARG(d->mode >= 0 && d->mode <= 2, "cfm_ffn_split: mode %d (0 rows, 1 projection, 2 feed-forward)", d->mode);
    CFM_CHECK_ARG(d->w_dtype == CFM_BF16 || d->w_dtype == CFM_F16, "cfm_ffn_split: w_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(!d->psum || (d->psum_splits > 0 && d->psum_b2), "cfm_ffn_split: partial slabs need their count and the second bias");
    CFM_CHECK_ARG((d->ln1_g == nullptr) == (d->ln1_b == nullptr) && (d->ln2_g == nullptr) == (d->ln2_b == nullptr), "cfm_ffn_split: a LayerNorm needs gain and bias");
    CFM_CHECK_ARG(!d->rows2_out || (d->mode == 0 && d->ln2_g), "cfm_ffn_split: rows2_out is the rows mode's second LayerNorm output");
    FfnSplitArgs a = {};
    a.x = d->x; a.psum = d->psum; a.pb2 = d->psum_b2; a.g_in = d->psum_splits; a.palpha = d->psum_alpha;
    a.ln1_g = d->ln1_g; a.ln1_b = d->ln1_b; a.ln2_g = d->ln2_g; a.ln2_b = d->ln2_b; a.rows_out = d->rows_out; a.rows2_out = d->rows2_out;
    a.M = d->M; a.eps = d->eps; a.act = d->act;
    int slices = 1;
    if (d->mode > 0) {
        CFM_CHECK_ARG(d->ln_g && d->ln_b && d->w1 && d->b1 && d->N1 > 0 && d->N1 % FS_SL == 0, "cfm_ffn_split: product 1 needs its LayerNorm, weights, bias and N1 %% 256 == 0");
        CFM_CHECK_ARG(d->act == CFM_ACT_NONE || d->act == CFM_ACT_SILU || d->act == CFM_ACT_RELU, "cfm_ffn_split: act must be none, SiLU or ReLU");
        a.ln_g = d->ln_g; a.ln_b = d->ln_b; a.w1 = (const u16*)d->w1; a.b1 = d->b1;
        slices = d->N1 / FS_SL;
        if (d->mode == 1) {
            CFM_CHECK_ARG(d->out16 && d->ldo >= d->N1 && d->ldo % 4 == 0, "cfm_ffn_split: the projection needs out16 with a row stride >= N1, multiple of 4");
            a.out16 = (u16*)d->out16; a.ldo = d->ldo;
            if (d->kv_ring) {
                CFM_CHECK_ARG(d->N1 == 3 * FS_D && d->ring_offsets && d->ring_H > 0 && FS_D % d->ring_H == 0 && (FS_D / d->ring_H) % 4 == 0 && d->ring_Tq > 0 &&
                              d->M % d->ring_Tq == 0 && d->ring_T >= d->ring_Tq,
                              "cfm_ffn_split: the K/V ring needs the fused q|k|v projection (N1 = 3 D), offsets, H | D with dk %% 4 == 0, M %% T == 0 and ring_T >= T");
                a.ring = d->kv_ring; a.ring_off = d->ring_offsets; a.ring_T = d->ring_T; a.ring_H = d->ring_H; a.ring_dk = FS_D / d->ring_H; a.ring_Tq = d->ring_Tq;
            }
        } else {
            CFM_CHECK_ARG(d->w2 && d->psum_out && d->psum_out != d->psum, "cfm_ffn_split: the feed-forward needs w2 and its own partial slabs");
            a.w2 = (const u16*)d->w2; a.psum_out = d->psum_out; a.FF = d->N1;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    return d->w_dtype == CFM_BF16 ? launch_mode<BF16>(a, d->mode, slices, s) : launch_mode<F16>(a, d->mode, slices, s);
}
