// attention_bwd.hip -- backward of the fused attention (attention.py:81-96 under autograd), gfx950.
//
//   P = softmax_j(scale * q'_i . k_j  [masked])      (q' = q + pos_bias_u, added by the projection's bias in train mode; the batch
//   O = P . V                                          path's positional term is constant along a row: softmax-invariant, no gradient)
//   delta_i = dO_i . O_i
//   dS = P * (dO . V^T - delta) * scale;   dQ = dS . K;   dK = dS^T . Q';   dV = P^T . dO
//
// Flash-style recomputation: the forward kept only the row log-sum-exp (cfm_attn_desc.lse), so P is rebuilt tile by tile as
// exp(scale * s - lse) and nothing of size Tq x Tk reaches memory.  Two kernels with the forward kernel's structure (cfm_attn_kernel:
// 256 threads, 4 wavefronts x 16 rows, 64-wide tiles staged in LDS, swapped MFMA orientation so a lane owns one query / key column):
//   cfm_attn_bwd_dq_kernel   workgroup = (b, h, 64 queries), loops over key tiles:   S^T = K Q'^T, dP^T = V dO^T, dQ^T += K^T dS^T
//   cfm_attn_bwd_dkv_kernel  workgroup = (b, h, 64 keys),    loops over query tiles: S = Q' K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q'^T dS
// Each computes the scores twice over (5 products instead of the minimal 4 + a cross-workgroup reduction) in exchange for no atomics:
// bitwise reproducible.  A fully masked row (lse = -inf) has P = 0 and contributes nothing -- the reference's masked_fill(mask, 0.0)
// after the softmax makes such a row a constant (attention.py:91-92).
// SPLIT evaluates every product as hi*hi + lo*hi + hi*lo on bf16 planes (f32-accurate mode, f32 tensors in memory).
#include "cfm_common.h"
#include "attn_common.h"

struct AttnBwdArgs {
    const void *q, *k, *v, *out, *dout;
    const float* lse;
    const uint8_t* mask;
    void *dq, *dkk, *dv;
    float* delta;
    int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, m_sb, m_sq;
    int B, H, Tq, Tk, dk;
    int io_dt, do_dt;
    float scale;
    CfmDrop drop;
};

namespace {

__global__ void cfm_attn_delta_kernel(const AttnBwdArgs a) {
    const int64_t n = (int64_t)a.B * a.H * a.Tq;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const int i = (int)(id % a.Tq);
    const int h = (int)((id / a.Tq) % a.H);
    const int b = (int)(id / ((int64_t)a.Tq * a.H));
    const int64_t o = ((int64_t)b * a.Tq + i) * ((int64_t)a.H * a.dk) + (int64_t)h * a.dk;
    float s = 0.f;
    for (int d = 0; d < a.dk; d += 8) {
        float x[8], y[8];
        load8f(a.out, a.io_dt, o + d, a.dk - d, x);            // 16-byte loads when aligned
        load8f(a.dout, a.do_dt, o + d, a.dk - d, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(x[e], y[e], s);
    }
    a.delta[id] = s;
}

// 64 rows x dk of `src` (row r at element offset base + r*stride, rows >= nrows read as zero) -> row-major swizzled planes `rm` (may be
// null) and transposed planes `tr` [d][row] (may be null)
template <typename HT, bool SPLIT>
__device__ __forceinline__ void stage_tile(const void* src, int dt, int64_t base, int64_t stride, int r0, int nrows, int dk, u32x4* rm, u16* tr, int tid) {
    constexpr int KS_PLANE = KT * 8, VT_PLANE = DKP * VSTR;
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        const int id = pss * 256 + tid;
        const int row = id >> 3, c = id & 7;
        const int rr = r0 + row;
        const int d0 = c * 8;
        const int nv = rr < nrows ? dk - d0 : 0;
        float f[8];
        load8f(src, dt, base + (int64_t)(rr < nrows ? rr : 0) * stride + d0, nv, f);
        if (rm) {
            u32x4 hi, lo;
            pack_planes<HT, SPLIT>(f, hi, lo);
            rm[k_swz(row, c)] = hi;
            if constexpr (SPLIT) rm[KS_PLANE + k_swz(row, c)] = lo;
        }
        if (tr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (SPLIT) {
                    const u16 hb = BF16::from_f32(f[j]);
                    tr[(d0 + j) * VSTR + row] = hb;
                    tr[VT_PLANE + (d0 + j) * VSTR + row] = BF16::from_f32(f[j] - BF16::to_f32(hb));
                } else {
                    tr[(d0 + j) * VSTR + row] = HT::from_f32(f[j]);
                }
            }
        }
    }
}

// acc[f] (f = 0..3: 16-row fragments of the LDS tile) += tile[f*16 + .., :] . frag^T   -- the "first product" of both kernels
template <typename HT, bool SPLIT>
__device__ __forceinline__ void tile_dot(const u32x4* tile, const u32x4 (&fr)[2], const u32x4 (&frl)[2], f32x4 (&acc)[4], int l15, int g) {
    constexpr int KS_PLANE = KT * 8;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int idx = k_swz(f * 16 + l15, kk * 4 + g);
            const u32x4 t = tile[idx];
            if constexpr (SPLIT) {
                acc[f] = HT::mfma(t, frl[kk], acc[f]);
                acc[f] = HT::mfma(tile[KS_PLANE + idx], fr[kk], acc[f]);
            }
            acc[f] = HT::mfma(t, fr[kk], acc[f]);
        }
    }
}

// acc[fd] += T^T[fd*16 + .., 32-step k2] . b^T  with T^T a transposed tile [d][row] (stride VSTR) and b the packed accumulator fragment
template <typename HT, bool SPLIT>
__device__ __forceinline__ void tr_dot(const u16* tr, int k2, const u32x4& bh, const u32x4& bl, f32x4 (&acc)[4], int l15, int g) {
    constexpr int VT_PLANE = DKP * VSTR;
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) {
        const int o0 = (fd * 16 + l15) * VSTR + (2 * k2) * 16 + g * 4;
        const u32x2 v0 = *(const u32x2*)(tr + o0);
        const u32x2 v1 = *(const u32x2*)(tr + o0 + 16);
        const u32x4 th = {v0.x, v0.y, v1.x, v1.y};
        if constexpr (SPLIT) {
            const u32x2 w0 = *(const u32x2*)(tr + VT_PLANE + o0);
            const u32x2 w1 = *(const u32x2*)(tr + VT_PLANE + o0 + 16);
            const u32x4 tl = {w0.x, w0.y, w1.x, w1.y};
            acc[fd] = HT::mfma(th, bl, acc[fd]);
            acc[fd] = HT::mfma(tl, bh, acc[fd]);
        }
        acc[fd] = HT::mfma(th, bh, acc[fd]);
    }
}

template <typename HT, bool SPLIT>
__device__ __forceinline__ void row_frags(const void* src, int dt, int64_t off, int dk, int g, u32x4 (&hi)[2], u32x4 (&lo)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int d0 = kk * 32 + g * 8;
        float f[8];
        load8f(src, dt, off + d0, dk - d0, f);
        pack_planes<HT, SPLIT>(f, hi[kk], lo[kk]);
    }
}

__device__ __forceinline__ void store_row4(void* base, int dt, int64_t off, const f32x4& o) {
    if (dt == CFM_F32) *(f32x4*)((float*)base + off) = o;
    else if (dt == CFM_BF16) *(u32x2*)((u16*)base + off) = (u32x2){pack2<BF16>(o.x, o.y), pack2<BF16>(o.z, o.w)};
    else *(u32x2*)((u16*)base + off) = (u32x2){pack2<F16>(o.x, o.y), pack2<F16>(o.z, o.w)};
}

template <typename HT, bool SPLIT>
__global__ __launch_bounds__(256) void cfm_attn_bwd_dq_kernel(const AttnBwdArgs a) {
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int KS_PLANE = KT * 8, VT_PLANE = DKP * VSTR;
    __shared__ u32x4 Ks[KS_PLANE * NPL];
    __shared__ u32x4 Vs[KS_PLANE * NPL];
    __shared__ __attribute__((aligned(16))) u16 Kt[VT_PLANE * NPL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * QT + wave * 16 + l15;
    const int qc = qi < a.Tq ? qi : a.Tq - 1;
    const int dk = a.dk;
    u32x4 qf[2], qfl[2], dof[2], dofl[2];
    row_frags<HT, SPLIT>(a.q, a.io_dt, (int64_t)b * a.q_sb + (int64_t)qc * a.q_st + h * dk, dk, g, qf, qfl);
    row_frags<HT, SPLIT>(a.dout, a.do_dt, ((int64_t)b * a.Tq + qc) * ((int64_t)a.H * dk) + (int64_t)h * dk, dk, g, dof, dofl);
    const int64_t rid = ((int64_t)b * a.H + h) * a.Tq + qc;
    const float lse_q = qi < a.Tq ? a.lse[rid] : -INFINITY;
    const float delta_q = a.delta[rid];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t kb = (int64_t)b * a.k_sb + (int64_t)h * dk, vb = (int64_t)b * a.v_sb + (int64_t)h * dk;
    const int ntiles = (a.Tk + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * KT;
        __syncthreads();
        stage_tile<HT, SPLIT>(a.k, a.io_dt, kb, a.k_st, k0, a.Tk, dk, Ks, Kt, tid);
        stage_tile<HT, SPLIT>(a.v, a.io_dt, vb, a.v_st, k0, a.Tk, dk, Vs, nullptr, tid);
        __syncthreads();
        f32x4 s[4], dp[4];
        tile_dot<HT, SPLIT>(Ks, qf, qfl, s, l15, g);       // s[f][r]: key k0 + f*16 + g*4 + r, query qi
        tile_dot<HT, SPLIT>(Vs, dof, dofl, dp, l15, g);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kj = k0 + f * 16 + g * 4 + r;
                bool ok = kj < a.Tk && lse_q != -INFINITY;
                if (ok && a.mask) ok = a.mask[(int64_t)b * a.m_sb + (int64_t)qc * a.m_sq + kj] != 0;
                const float p = ok ? __expf(s[f][r] * a.scale - lse_q) : 0.f;
                float dpe = dp[f][r];                          // d loss / d P through the forward's dropout mask
                if (a.drop.thresh) dpe = cfm_drop(a.drop, (unsigned)rid * (unsigned)a.Tk + (unsigned)kj, dpe);
                s[f][r] = p * (dpe - delta_q) * a.scale;       // dS
            }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const float pf[8] = {s[2 * k2][0], s[2 * k2][1], s[2 * k2][2], s[2 * k2][3], s[2 * k2 + 1][0], s[2 * k2 + 1][1], s[2 * k2 + 1][2], s[2 * k2 + 1][3]};
            u32x4 ph, pl;
            pack_planes<HT, SPLIT>(pf, ph, pl);
            tr_dot<HT, SPLIT>(Kt, k2, ph, pl, acc, l15, g);   // dQ^T[d, q] += K^T[d, key] dS^T[key, q]
        }
    }
    if (qi < a.Tq) {
        const int64_t ob = (int64_t)b * a.q_sb + (int64_t)qi * a.q_st + h * dk;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            const int d = fd * 16 + g * 4;
            if (d < dk) store_row4(a.dq, a.io_dt, ob + d, acc[fd]);
        }
    }
}

template <typename HT, bool SPLIT>
__global__ __launch_bounds__(256) void cfm_attn_bwd_dkv_kernel(const AttnBwdArgs a) {
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int KS_PLANE = KT * 8, VT_PLANE = DKP * VSTR;
    __shared__ u32x4 Qs[KS_PLANE * NPL];
    __shared__ u32x4 Os[KS_PLANE * NPL];
    __shared__ __attribute__((aligned(16))) u16 Qt[VT_PLANE * NPL];
    __shared__ __attribute__((aligned(16))) u16 Ot[VT_PLANE * NPL];
    __shared__ float Ls[QT], Ds[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kj = blockIdx.x * KT + wave * 16 + l15;
    const int kc = kj < a.Tk ? kj : a.Tk - 1;
    const int dk = a.dk;
    u32x4 kf[2], kfl[2], vf[2], vfl[2];
    row_frags<HT, SPLIT>(a.k, a.io_dt, (int64_t)b * a.k_sb + (int64_t)kc * a.k_st + h * dk, dk, g, kf, kfl);
    row_frags<HT, SPLIT>(a.v, a.io_dt, (int64_t)b * a.v_sb + (int64_t)kc * a.v_st + h * dk, dk, g, vf, vfl);
    f32x4 acc_k[4], acc_v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_k[i] = acc_v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t qb = (int64_t)b * a.q_sb + (int64_t)h * dk, ob = (int64_t)b * a.Tq * ((int64_t)a.H * dk) + (int64_t)h * dk;
    const int64_t rb = ((int64_t)b * a.H + h) * a.Tq;
    const int ntiles = (a.Tq + QT - 1) / QT;
    for (int t = 0; t < ntiles; ++t) {
        const int q0 = t * QT;
        __syncthreads();
        stage_tile<HT, SPLIT>(a.q, a.io_dt, qb, a.q_st, q0, a.Tq, dk, Qs, Qt, tid);
        stage_tile<HT, SPLIT>(a.dout, a.do_dt, ob, (int64_t)a.H * dk, q0, a.Tq, dk, Os, Ot, tid);
        if (tid < QT) {
            const int qq = q0 + tid;
            Ls[tid] = qq < a.Tq ? a.lse[rb + qq] : -INFINITY;
            Ds[tid] = qq < a.Tq ? a.delta[rb + qq] : 0.f;
        }
        __syncthreads();
        f32x4 s[4], dp[4];
        tile_dot<HT, SPLIT>(Qs, kf, kfl, s, l15, g);       // s[f][r]: query q0 + f*16 + g*4 + r, key kj
        tile_dot<HT, SPLIT>(Os, vf, vfl, dp, l15, g);
        float pv[4][4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = f * 16 + g * 4 + r, qq = q0 + ql;
                const float L = Ls[ql];
                bool ok = qq < a.Tq && kj < a.Tk && L != -INFINITY;
                if (ok && a.mask) ok = a.mask[(int64_t)b * a.m_sb + (int64_t)qq * a.m_sq + kj] != 0;
                const float p = ok ? __expf(s[f][r] * a.scale - L) : 0.f;
                float pd = p, dpe = dp[f][r];
                if (a.drop.thresh) {
                    const unsigned e = (unsigned)(rb + (qq < a.Tq ? qq : 0)) * (unsigned)a.Tk + (unsigned)(kj < a.Tk ? kj : 0);
                    pd = cfm_drop(a.drop, e, p);               // the probabilities the forward multiplied V with
                    dpe = cfm_drop(a.drop, e, dpe);
                }
                pv[f][r] = pd;
                s[f][r] = p * (dpe - Ds[ql]) * a.scale;        // dS
            }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const float pf[8] = {pv[2 * k2][0], pv[2 * k2][1], pv[2 * k2][2], pv[2 * k2][3], pv[2 * k2 + 1][0], pv[2 * k2 + 1][1], pv[2 * k2 + 1][2], pv[2 * k2 + 1][3]};
            const float sf[8] = {s[2 * k2][0], s[2 * k2][1], s[2 * k2][2], s[2 * k2][3], s[2 * k2 + 1][0], s[2 * k2 + 1][1], s[2 * k2 + 1][2], s[2 * k2 + 1][3]};
            u32x4 ph, pl, sh, sl;
            pack_planes<HT, SPLIT>(pf, ph, pl);
            pack_planes<HT, SPLIT>(sf, sh, sl);
            tr_dot<HT, SPLIT>(Ot, k2, ph, pl, acc_v, l15, g);   // dV^T[d, key] += dO^T[d, q] P[q, key]
            tr_dot<HT, SPLIT>(Qt, k2, sh, sl, acc_k, l15, g);   // dK^T[d, key] += Q'^T[d, q] dS[q, key]
        }
    }
    if (kj < a.Tk) {
        const int64_t ko = (int64_t)b * a.k_sb + (int64_t)kj * a.k_st + h * dk, vo = (int64_t)b * a.v_sb + (int64_t)kj * a.v_st + h * dk;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            const int d = fd * 16 + g * 4;
            if (d < dk) {
                store_row4(a.dkk, a.io_dt, ko + d, acc_k[fd]);
                store_row4(a.dv, a.io_dt, vo + d, acc_v[fd]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Fast forms for what a bf16 / fp16 training step actually runs: d_k = 64, 16-bit q / k / v / dO rows (16-byte aligned), no mask or a
// key-validity mask (B, 1, Tk).  Same products, same element-wise arithmetic, same MFMA order as the general kernels above --
// bit-identical results (tests) -- but
//   * tiles are staged as they lie in memory (16-byte pieces into the swizzled row-major image, no f32 round trip) and the TRANSPOSED
//     operands of the second products are read straight out of that image with ds_read_b64_tr_b16: no transposed copy is written
//     (the general kernels spend 32 two-byte LDS stores per thread and tile on it);
//   * two LDS buffers and a register prefetch: the next tile's global loads are in flight while this one is multiplied, one barrier per
//     tile instead of two with the loads exposed in between;
//   * mask bytes: the dkv kernel's lane owns ONE key (loaded once), the dq kernel stages the tile's 64 validity bytes with the tile;
//   * 16 KB of LDS per buffer pair: several workgroups per CU.
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// A-operand fragment of the transposed tile: column d = fd*16 + l15, rows 32 k2 + 4 g .. + 3 and + 16 (the k order of the packed accumulators)
__device__ __forceinline__ u32x4 tr_frag64(const u32x4* tile, int k2, int fd, int l15, int g) {
    const int row = k2 * 32 + 4 * g + (l15 >> 2);
    const int col = fd * 16 + (l15 & 3) * 4;
    const unsigned char* pa = (const unsigned char*)tile + row * 128 + (((col >> 3) ^ ((row >> 1) & 7)) << 4) + (col & 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa + 16 * 128));
    const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
    return (u32x4){lo2.x, lo2.y, hi2.x, hi2.y};
}

// this thread's two 16-byte pieces of a 64-row x 64-column 16-bit tile (rows r0 .. r0+63 of src; rows >= nrows read as zero)
__device__ __forceinline__ void tile_load64(const u16* src, int64_t base, int64_t stride, int r0, int nrows, int tid, u32x4 (&r)[2]) {
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        const int id = pss * 256 + tid, row = id >> 3, c = id & 7;
        const int rr = r0 + row;
        const u32x4 v = *(const u32x4*)(src + base + (int64_t)(rr < nrows ? rr : 0) * stride + c * 8);
        r[pss] = rr < nrows ? v : (u32x4){0u, 0u, 0u, 0u};
    }
}
__device__ __forceinline__ void tile_store64(u32x4* tile, int tid, const u32x4 (&r)[2]) {
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        const int id = pss * 256 + tid;
        tile[k_swz(id >> 3, id & 7)] = r[pss];
    }
}
template <typename HT>
__device__ __forceinline__ void tile_dot64(const u32x4* tile, const u32x4 (&fr)[2], f32x4 (&acc)[4], int l15, int g) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc[f] = HT::mfma(tile[k_swz(f * 16 + l15, kk * 4 + g)], fr[kk], acc[f]);
    }
}

// MFULL: a (B, Tq, Tk) mask (chunk mask & padding, utils.py:96-160: the dynamic-chunk training recipe of train.sh:52-53) -- the byte of
// (query, key) is read from memory by the lane that owns the pair (4 consecutive keys / queries per fragment); the key-validity row stays in LDS
template <typename HT, bool MFULL>
__device__ __forceinline__ void attn_bwd_dq_fast_body(const AttnBwdArgs& a, const int block_x, const int block_y, const int block_z) {
    constexpr int TP = KT * 8;                           // u32x4 per tile
    __shared__ u32x4 Ks[2][TP], Vs[2][TP];
    __shared__ __attribute__((aligned(16))) uint8_t Ms[2][KT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int b = block_z, h = block_y;
    const int qi = block_x * QT + wave * 16 + l15;
    const int qc = qi < a.Tq ? qi : a.Tq - 1;
    constexpr int dk = 64;
    const u16 *qp = (const u16*)a.q, *kp = (const u16*)a.k, *vp = (const u16*)a.v, *dop = (const u16*)a.dout;
    u32x4 qf[2], dof[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        qf[kk] = *(const u32x4*)(qp + (int64_t)b * a.q_sb + (int64_t)qc * a.q_st + h * dk + kk * 32 + g * 8);
        dof[kk] = *(const u32x4*)(dop + ((int64_t)b * a.Tq + qc) * ((int64_t)a.H * dk) + (int64_t)h * dk + kk * 32 + g * 8);
    }
    const int64_t rid = ((int64_t)b * a.H + h) * a.Tq + qc;
    const float lse_q = qi < a.Tq ? a.lse[rid] : -INFINITY;
    const float delta_q = a.delta[rid];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t kb = (int64_t)b * a.k_sb + (int64_t)h * dk, vb = (int64_t)b * a.v_sb + (int64_t)h * dk;
    const int ntiles = (a.Tk + KT - 1) / KT;
    u32x4 rk[2], rv[2];
    uint8_t rm = 1;
    auto fetch = [&](int t) __attribute__((always_inline)) {
        tile_load64(kp, kb, a.k_st, t * KT, a.Tk, tid, rk);
        tile_load64(vp, vb, a.v_st, t * KT, a.Tk, tid, rv);
        if (tid < KT) {
            const int kj = t * KT + tid;
            rm = kj < a.Tk ? 1 : 0;
            if (!MFULL && rm && a.mask) rm = a.mask[(int64_t)b * a.m_sb + kj] != 0;
        }
    };
    const uint8_t* const mrow = MFULL ? a.mask + (int64_t)b * a.m_sb + (int64_t)qc * a.m_sq : nullptr;   // this lane's query row of the mask
    auto put = [&](int buf) __attribute__((always_inline)) {
        tile_store64(Ks[buf], tid, rk);
        tile_store64(Vs[buf], tid, rv);
        if (tid < KT) Ms[buf][tid] = rm;
    };
    fetch(0);
    put(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * KT;
        if (t + 1 < ntiles) fetch(t + 1);
        f32x4 s[4], dp[4];
        uint8_t mq[MFULL ? 16 : 1];
        if constexpr (MFULL) {                            // requested before the products, used after them
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kj = k0 + f * 16 + g * 4 + r;
                    mq[f * 4 + r] = mrow[kj < a.Tk ? kj : a.Tk - 1];
                }
        }
        tile_dot64<HT>(Ks[buf], qf, s, l15, g);          // s[f][r]: key k0 + f*16 + g*4 + r, query qi
        tile_dot64<HT>(Vs[buf], dof, dp, l15, g);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const unsigned mb = *(const unsigned*)(&Ms[buf][f * 16 + g * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kj = k0 + f * 16 + g * 4 + r;
                bool ok = ((mb >> (8 * r)) & 0xffu) != 0 && lse_q != -INFINITY;
                if constexpr (MFULL) ok = ok && mq[f * 4 + r] != 0;
                const float p = ok ? __expf(s[f][r] * a.scale - lse_q) : 0.f;
                float dpe = dp[f][r];
                if (a.drop.thresh) dpe = cfm_drop(a.drop, (unsigned)rid * (unsigned)a.Tk + (unsigned)kj, dpe);
                s[f][r] = p * (dpe - delta_q) * a.scale;   // dS
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const u32x4 ph = pack8<HT>(s[2 * k2], s[2 * k2 + 1]);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) acc[fd] = HT::mfma(tr_frag64(Ks[buf], k2, fd, l15, g), ph, acc[fd]);   // dQ^T[d, q] += K^T[d, key] dS^T[key, q]
        }
        if (t + 1 < ntiles) put(buf ^ 1);
        __syncthreads();
    }
    if (qi < a.Tq) {
        const int64_t ob = (int64_t)b * a.q_sb + (int64_t)qi * a.q_st + h * dk;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) store_row4(a.dq, a.io_dt, ob + fd * 16 + g * 4, acc[fd]);
    }
}

// one conformer window's micro-batches in one launch (cfm_attention_bwd_group): see cfm_attn2_group_kernel in attention.hip
constexpr int ATTN_GROUP_MAX = 8;
struct AttnBwdGroupArgs {
    AttnBwdArgs a[ATTN_GROUP_MAX];
    int first[ATTN_GROUP_MAX + 1];     // first workgroup of each problem (grid of this launch)
    int gx[ATTN_GROUP_MAX];            // its grid x extent (query or key tiles); y = H, z = B
    int n;
};

__device__ __forceinline__ int attn_group_pick(const AttnBwdGroupArgs& G, int b) {
    int idx = 0;
#pragma unroll
    for (int i = 1; i < ATTN_GROUP_MAX; ++i)
        if (i < G.n && b >= G.first[i]) idx = i;          // uniform
    return idx;
}

template <typename HT, bool MFULL>
__global__ __launch_bounds__(256) void cfm_attn_bwd_dq_fast_kernel(const AttnBwdArgs a) {
    attn_bwd_dq_fast_body<HT, MFULL>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

template <typename HT, bool MFULL>
__global__ __launch_bounds__(256) void cfm_attn_bwd_dq_group_kernel(const AttnBwdGroupArgs G) {
    const int idx = attn_group_pick(G, (int)blockIdx.x);
    const int rel = (int)blockIdx.x - G.first[idx], gx = G.gx[idx], H = G.a[idx].H;
    attn_bwd_dq_fast_body<HT, MFULL>(G.a[idx], rel % gx, (rel / gx) % H, rel / (gx * H));
}

template <typename HT, bool MFULL>
__device__ __forceinline__ void attn_bwd_dkv_fast_body(const AttnBwdArgs& a, const int block_x, const int block_y, const int block_z) {
    constexpr int TP = QT * 8;
    __shared__ u32x4 Qs[2][TP], Os[2][TP];
    __shared__ __attribute__((aligned(16))) float Ls[2][QT], Ds[2][QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int b = block_z, h = block_y;
    const int kj = block_x * KT + wave * 16 + l15;
    const int kc = kj < a.Tk ? kj : a.Tk - 1;
    constexpr int dk = 64;
    const u16 *qp = (const u16*)a.q, *kp = (const u16*)a.k, *vp = (const u16*)a.v, *dop = (const u16*)a.dout;
    u32x4 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        kf[kk] = *(const u32x4*)(kp + (int64_t)b * a.k_sb + (int64_t)kc * a.k_st + h * dk + kk * 32 + g * 8);
        vf[kk] = *(const u32x4*)(vp + (int64_t)b * a.v_sb + (int64_t)kc * a.v_st + h * dk + kk * 32 + g * 8);
    }
    bool key_ok = kj < a.Tk;
    if (!MFULL && key_ok && a.mask) key_ok = a.mask[(int64_t)b * a.m_sb + kj] != 0;
    const uint8_t* const mcol = MFULL ? a.mask + (int64_t)b * a.m_sb + kc : nullptr;      // this lane's key column of the mask (row stride m_sq)
    f32x4 acc_k[4], acc_v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_k[i] = acc_v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t qb = (int64_t)b * a.q_sb + (int64_t)h * dk, ob = (int64_t)b * a.Tq * ((int64_t)a.H * dk) + (int64_t)h * dk;
    const int64_t rb = ((int64_t)b * a.H + h) * a.Tq;
    const int ntiles = (a.Tq + QT - 1) / QT;
    u32x4 rq[2], ro[2];
    float rl = -INFINITY, rd = 0.f;
    auto fetch = [&](int t) __attribute__((always_inline)) {
        tile_load64(qp, qb, a.q_st, t * QT, a.Tq, tid, rq);
        tile_load64(dop, ob, (int64_t)a.H * dk, t * QT, a.Tq, tid, ro);
        if (tid < QT) {
            const int qq = t * QT + tid;
            const int qx = qq < a.Tq ? qq : a.Tq - 1;
            const float l = a.lse[rb + qx], d = a.delta[rb + qx];
            rl = qq < a.Tq ? l : -INFINITY;
            rd = qq < a.Tq ? d : 0.f;
        }
    };
    auto put = [&](int buf) __attribute__((always_inline)) {
        tile_store64(Qs[buf], tid, rq);
        tile_store64(Os[buf], tid, ro);
        if (tid < QT) { Ls[buf][tid] = rl; Ds[buf][tid] = rd; }
    };
    fetch(0);
    put(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, q0 = t * QT;
        if (t + 1 < ntiles) fetch(t + 1);
        f32x4 s[4], dp[4];
        uint8_t mk[MFULL ? 16 : 1];
        if constexpr (MFULL) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + f * 16 + g * 4 + r;
                    mk[f * 4 + r] = mcol[(int64_t)(qq < a.Tq ? qq : a.Tq - 1) * a.m_sq];
                }
        }
        tile_dot64<HT>(Qs[buf], kf, s, l15, g);          // s[f][r]: query q0 + f*16 + g*4 + r, key kj
        tile_dot64<HT>(Os[buf], vf, dp, l15, g);
        f32x4 pv[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 L4 = *(const f32x4*)(&Ls[buf][f * 16 + g * 4]), D4 = *(const f32x4*)(&Ds[buf][f * 16 + g * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + f * 16 + g * 4 + r;
                const float L = L4[r];
                bool ok = key_ok && L != -INFINITY;         // (queries past Tq carry L = -inf)
                if constexpr (MFULL) ok = ok && mk[f * 4 + r] != 0;
                const float p = ok ? __expf(s[f][r] * a.scale - L) : 0.f;
                float pd = p, dpe = dp[f][r];
                if (a.drop.thresh) {
                    const unsigned e = (unsigned)(rb + (qq < a.Tq ? qq : 0)) * (unsigned)a.Tk + (unsigned)(kj < a.Tk ? kj : 0);
                    pd = cfm_drop(a.drop, e, p);
                    dpe = cfm_drop(a.drop, e, dpe);
                }
                pv[f][r] = pd;
                s[f][r] = p * (dpe - D4[r]) * a.scale;     // dS
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const u32x4 ph = pack8<HT>(pv[2 * k2], pv[2 * k2 + 1]), sh = pack8<HT>(s[2 * k2], s[2 * k2 + 1]);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                acc_v[fd] = HT::mfma(tr_frag64(Os[buf], k2, fd, l15, g), ph, acc_v[fd]);   // dV^T[d, key] += dO^T[d, q] P[q, key]
                acc_k[fd] = HT::mfma(tr_frag64(Qs[buf], k2, fd, l15, g), sh, acc_k[fd]);   // dK^T[d, key] += Q'^T[d, q] dS[q, key]
            }
        }
        if (t + 1 < ntiles) put(buf ^ 1);
        __syncthreads();
    }
    if (kj < a.Tk) {
        const int64_t ko = (int64_t)b * a.k_sb + (int64_t)kj * a.k_st + h * dk, vo = (int64_t)b * a.v_sb + (int64_t)kj * a.v_st + h * dk;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            store_row4(a.dkk, a.io_dt, ko + fd * 16 + g * 4, acc_k[fd]);
            store_row4(a.dv, a.io_dt, vo + fd * 16 + g * 4, acc_v[fd]);
        }
    }
}

template <typename HT, bool MFULL>
__global__ __launch_bounds__(256) void cfm_attn_bwd_dkv_fast_kernel(const AttnBwdArgs a) {
    attn_bwd_dkv_fast_body<HT, MFULL>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

template <typename HT, bool MFULL>
__global__ __launch_bounds__(256) void cfm_attn_bwd_dkv_group_kernel(const AttnBwdGroupArgs G) {
    const int idx = attn_group_pick(G, (int)blockIdx.x);
    const int rel = (int)blockIdx.x - G.first[idx], gx = G.gx[idx], H = G.a[idx].H;
    attn_bwd_dkv_fast_body<HT, MFULL>(G.a[idx], rel % gx, (rel / gx) % H, rel / (gx * H));
}

// delta_i = dO_i . O_i for d_k = 64 16-bit rows: the row's sixteen 16-byte pieces requested together (the general kernel's loads sit behind
// run-time dtype branches and go out one latency after the other)
template <typename HT>
__device__ __forceinline__ void attn_delta_fast_body(const AttnBwdArgs& a, const int block_x) {
    const int64_t n = (int64_t)a.B * a.H * a.Tq;
    const int64_t id = (int64_t)block_x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const int i = (int)(id % a.Tq);
    const int h = (int)((id / a.Tq) % a.H);
    const int b = (int)(id / ((int64_t)a.Tq * a.H));
    const int64_t o = ((int64_t)b * a.Tq + i) * ((int64_t)a.H * 64) + (int64_t)h * 64;
    u32x4 x[8], y[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        x[c] = *(const u32x4*)((const u16*)a.out + o + c * 8);
        y[c] = *(const u32x4*)((const u16*)a.dout + o + c * 8);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {                          // same element order as the general kernel: bit-identical
        const unsigned xw[4] = {x[c].x, x[c].y, x[c].z, x[c].w}, yw[4] = {y[c].x, y[c].y, y[c].z, y[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s = fmaf(HT::to_f32((u16)(xw[e] & 0xffffu)), HT::to_f32((u16)(yw[e] & 0xffffu)), s);
            s = fmaf(HT::to_f32((u16)(xw[e] >> 16)), HT::to_f32((u16)(yw[e] >> 16)), s);
        }
    }
    a.delta[id] = s;
}

template <typename HT>
__global__ void cfm_attn_delta_fast_kernel(const AttnBwdArgs a) {
    attn_delta_fast_body<HT>(a, (int)blockIdx.x);
}

template <typename HT>
__global__ void cfm_attn_delta_group_kernel(const AttnBwdGroupArgs G) {
    const int idx = attn_group_pick(G, (int)blockIdx.x);
    attn_delta_fast_body<HT>(G.a[idx], (int)blockIdx.x - G.first[idx]);
}

static bool g_attn_bwd_general_only = false;      // tests: force the general kernels

template <typename HT>
bool bwd_fast_ok(const AttnBwdArgs& a) {
    const auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return !g_attn_bwd_general_only && a.dk == 64 && a.io_dt == HT::kId && a.do_dt == HT::kId && a.q_st % 8 == 0 &&
           a.k_st % 8 == 0 && a.v_st % 8 == 0 && a.q_sb % 8 == 0 && a.k_sb % 8 == 0 && a.v_sb % 8 == 0 && al16(a.q) && al16(a.k) && al16(a.v) && al16(a.dout) &&
           al16(a.out);
}

// three launches for n problems: delta, dq, dkv -- each over all problems (callers have checked bwd_fast_ok for every one)
template <typename HT>
int launch_bwd_group(AttnBwdGroupArgs& G, hipStream_t s, const char* n_dq, const char* n_dkv) {
    const bool mfull = G.a[0].mask && G.a[0].m_sq != 0;     // the caller groups problems of one mask kind
    double fl = 0.0, by = 0.0, bd = 0.0;
    int first = 0;
    for (int i = 0; i < G.n; ++i) {
        const AttnBwdArgs& a = G.a[i];
        fl += 2.0 * a.B * a.H * (double)a.Tq * a.Tk * a.dk;
        by += (double)a.B * a.H * (a.Tq + a.Tk) * a.dk * cfm_elt_size(a.io_dt) * 3;
        bd += 2.0 * a.B * a.Tq * a.H * a.dk * cfm_elt_size(a.io_dt);
        G.first[i] = first;
        first += (int)(((int64_t)a.B * a.H * a.Tq + 255) / 256);
    }
    for (int i = G.n; i <= ATTN_GROUP_MAX; ++i) G.first[i] = first;
    {
        CfmProfScope prof("attn_bwd_delta_group", s, 0.0, bd);
        CFM_LAUNCH((cfm_attn_delta_group_kernel<HT>), dim3((unsigned)first), dim3(256), 0, s, G);
        if (int rc = cfm_launch_status("cfm_attention_bwd_group (delta)")) return rc;
    }
    first = 0;
    for (int i = 0; i < G.n; ++i) {
        const AttnBwdArgs& a = G.a[i];
        G.first[i] = first;
        G.gx[i] = (a.Tq + QT - 1) / QT;
        first += G.gx[i] * a.H * a.B;
    }
    for (int i = G.n; i <= ATTN_GROUP_MAX; ++i) G.first[i] = first;
    {
        CfmProfScope prof(n_dq, s, 3.0 * fl, by);
        if (mfull) CFM_LAUNCH((cfm_attn_bwd_dq_group_kernel<HT, true>), dim3((unsigned)first), dim3(256), 0, s, G);
        else CFM_LAUNCH((cfm_attn_bwd_dq_group_kernel<HT, false>), dim3((unsigned)first), dim3(256), 0, s, G);
        if (int rc = cfm_launch_status("cfm_attention_bwd_group (dq)")) return rc;
    }
    first = 0;
    for (int i = 0; i < G.n; ++i) {
        const AttnBwdArgs& a = G.a[i];
        G.first[i] = first;
        G.gx[i] = (a.Tk + KT - 1) / KT;
        first += G.gx[i] * a.H * a.B;
    }
    for (int i = G.n; i <= ATTN_GROUP_MAX; ++i) G.first[i] = first;
    CfmProfScope prof(n_dkv, s, 4.0 * fl, by);
    if (mfull) CFM_LAUNCH((cfm_attn_bwd_dkv_group_kernel<HT, true>), dim3((unsigned)first), dim3(256), 0, s, G);
    else CFM_LAUNCH((cfm_attn_bwd_dkv_group_kernel<HT, false>), dim3((unsigned)first), dim3(256), 0, s, G);
    return cfm_launch_status("cfm_attention_bwd_group (dkv)");
}

template <typename HT, bool SPLIT>
int launch_bwd(const AttnBwdArgs& a, hipStream_t s, const char* n_dq, const char* n_dkv) {
    const double fl = 2.0 * a.B * a.H * (double)a.Tq * a.Tk * a.dk;
    const double by = (double)a.B * a.H * (a.Tq + a.Tk) * a.dk * cfm_elt_size(a.io_dt) * 3;
    bool fast = false;
    if constexpr (!SPLIT) fast = bwd_fast_ok<HT>(a);
    {
        CfmProfScope prof("attn_bwd_delta", s, 0.0, 2.0 * a.B * a.Tq * a.H * a.dk * cfm_elt_size(a.io_dt));
        const int64_t n = (int64_t)a.B * a.H * a.Tq;
        if constexpr (!SPLIT) {
            if (fast) CFM_LAUNCH((cfm_attn_delta_fast_kernel<HT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
            else CFM_LAUNCH(cfm_attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        } else {
            CFM_LAUNCH(cfm_attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        }
        if (int rc = cfm_launch_status("cfm_attention_bwd (delta)")) return rc;
    }
    {
        CfmProfScope prof(n_dq, s, 3.0 * fl, by);
        const dim3 grid((unsigned)((a.Tq + QT - 1) / QT), (unsigned)a.H, (unsigned)a.B);
        if constexpr (!SPLIT) {
            if (fast && a.mask && a.m_sq != 0) CFM_LAUNCH((cfm_attn_bwd_dq_fast_kernel<HT, true>), grid, dim3(256), 0, s, a);
            else if (fast) CFM_LAUNCH((cfm_attn_bwd_dq_fast_kernel<HT, false>), grid, dim3(256), 0, s, a);
            else CFM_LAUNCH((cfm_attn_bwd_dq_kernel<HT, SPLIT>), grid, dim3(256), 0, s, a);
        } else {
            CFM_LAUNCH((cfm_attn_bwd_dq_kernel<HT, SPLIT>), grid, dim3(256), 0, s, a);
        }
        if (int rc = cfm_launch_status("cfm_attention_bwd (dq)")) return rc;
    }
    CfmProfScope prof(n_dkv, s, 4.0 * fl, by);
    const dim3 grid((unsigned)((a.Tk + KT - 1) / KT), (unsigned)a.H, (unsigned)a.B);
    if constexpr (!SPLIT) {
        if (fast && a.mask && a.m_sq != 0) CFM_LAUNCH((cfm_attn_bwd_dkv_fast_kernel<HT, true>), grid, dim3(256), 0, s, a);
        else if (fast) CFM_LAUNCH((cfm_attn_bwd_dkv_fast_kernel<HT, false>), grid, dim3(256), 0, s, a);
        else CFM_LAUNCH((cfm_attn_bwd_dkv_kernel<HT, SPLIT>), grid, dim3(256), 0, s, a);
    } else {
        CFM_LAUNCH((cfm_attn_bwd_dkv_kernel<HT, SPLIT>), grid, dim3(256), 0, s, a);
    }
    return cfm_launch_status("cfm_attention_bwd (dkv)");
}

}  // namespace

extern "C" void cfm_attention_bwd_force_general(int32_t on) { g_attn_bwd_general_only = on != 0; }

static int attn_bwd_args(const cfm_attn_bwd_desc* d, AttnBwdArgs& a) {
    CFM_CHECK_ARG(d && d->q && d->k && d->v && d->out && d->dout && d->lse && d->grad_q && d->grad_k && d->grad_v && d->delta, "cfm_attention_bwd: null pointer");
    CFM_CHECK_ARG(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "cfm_attention_bwd: empty problem");
    CFM_CHECK_ARG(d->dk > 0 && d->dk <= DKP && d->dk % 4 == 0, "cfm_attention_bwd: need dk %% 4 == 0 and dk <= 64 (dk=%d)", d->dk);
    CFM_CHECK_ARG(d->B <= 65535 && d->H <= 65535, "cfm_attention_bwd: B,H must fit a grid dimension");
    CFM_CHECK_ARG(d->mma_dtype == CFM_BF16 || d->mma_dtype == CFM_F16, "cfm_attention_bwd: mma_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(!d->split || d->mma_dtype == CFM_BF16, "cfm_attention_bwd: split mode uses bf16 planes");
    CFM_CHECK_ARG(d->io_dtype >= CFM_F32 && d->io_dtype <= CFM_F16 && d->dout_dtype >= CFM_F32 && d->dout_dtype <= CFM_F16, "cfm_attention_bwd: bad dtype");
    a.q = d->q; a.k = d->k; a.v = d->v; a.out = d->out; a.dout = d->dout; a.lse = d->lse; a.mask = d->mask;
    a.dq = d->grad_q; a.dkk = d->grad_k; a.dv = d->grad_v; a.delta = d->delta;
    a.q_sb = d->q_sb; a.q_st = d->q_st; a.k_sb = d->k_sb; a.k_st = d->k_st; a.v_sb = d->v_sb; a.v_st = d->v_st; a.m_sb = d->m_sb; a.m_sq = d->m_sq;
    a.B = d->B; a.H = d->H; a.Tq = d->Tq; a.Tk = d->Tk; a.dk = d->dk; a.io_dt = d->io_dtype; a.do_dt = d->dout_dtype; a.scale = d->scale;
    CFM_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f, "cfm_attention_bwd: dropout probability must be in [0, 1)");
    a.drop = cfm_make_drop(d->drop_p, d->drop_seed);
    return CFM_OK;
}

extern "C" int cfm_attention_bwd(const cfm_attn_bwd_desc* d, cfm_stream_t stream) {
    AttnBwdArgs a;
    if (int rc = attn_bwd_args(d, a)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d->split) return launch_bwd<BF16, true>(a, s, "attn_bwd_dq_bf16x3", "attn_bwd_dkv_bf16x3");
    if (d->mma_dtype == CFM_BF16) return launch_bwd<BF16, false>(a, s, "attn_bwd_dq_bf16", "attn_bwd_dkv_bf16");
    return launch_bwd<F16, false>(a, s, "attn_bwd_dq_f16", "attn_bwd_dkv_f16");
}

// n attention backward problems (the micro-batches of a training window) in three launches instead of 3 n when every one takes the d_k = 64 /
// 16-bit fast kernels (no (B,Tq,Tk) mask); otherwise problem after problem.  Same results either way.
extern "C" int cfm_attention_bwd_group(const cfm_attn_bwd_desc* descs, int32_t n, cfm_stream_t stream) {
    CFM_CHECK_ARG(descs && n > 0, "cfm_attention_bwd_group: no problems");
    hipStream_t s = (hipStream_t)stream;
    AttnBwdGroupArgs G;
    bool groupable = n >= 2 && n <= ATTN_GROUP_MAX;
    for (int i = 0; i < n && groupable; ++i) {
        if (int rc = attn_bwd_args(&descs[i], G.a[i])) return rc;
        const bool mf = G.a[i].mask && G.a[i].m_sq != 0, mf0 = G.a[0].mask && G.a[0].m_sq != 0;
        groupable = !descs[i].split && descs[i].mma_dtype == descs[0].mma_dtype && mf == mf0 &&
                    (descs[i].mma_dtype == CFM_BF16 ? bwd_fast_ok<BF16>(G.a[i]) : bwd_fast_ok<F16>(G.a[i]));
    }
    if (!groupable) {
        for (int i = 0; i < n; ++i)
            if (int rc = cfm_attention_bwd(&descs[i], stream)) return rc;
        return CFM_OK;
    }
    G.n = n;
    for (int i = n; i < ATTN_GROUP_MAX; ++i) { G.a[i] = G.a[0]; G.gx[i] = 1; }
    if (descs[0].mma_dtype == CFM_BF16) return launch_bwd_group<BF16>(G, s, "attn_bwd_dq_group_bf16", "attn_bwd_dkv_group_bf16");
    return launch_bwd_group<F16>(G, s, "attn_bwd_dq_group_f16", "attn_bwd_dkv_group_f16");
}
