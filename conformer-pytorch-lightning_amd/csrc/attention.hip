// attention.hip -- fused (relative-position) multi-head self-attention for gfx950.
//
//   out[b,i,h,:] = softmax_j( scale * ((q_i+u_h).k_j + (q_i+v_h).p_{b,j}) ) . v_j      (attention.py:81-96)
//
// with the reference's masking rule (masked score = -inf, fully masked row -> zero context) and its
// un-shifted positional term (p_{b,j} is one broadcast row in the batch path, an absolute
// position-by-key table in the streaming path).  Nothing of size Tq x Tk ever reaches HBM: scores,
// probabilities and the running max/sum live in registers (online softmax over 64-key tiles).
//
// Work split: one 256-thread workgroup per (batch, head, 64-query tile); each of its 4 wavefronts
// owns 16 queries.  Both MFMA products are evaluated in SWAPPED orientation,
//     S^T[key, q] = K~[key, :] . Q~[q, :]^T          O^T[d, q] = V^T[d, key] . P^T[key, q]
// so that a lane always owns ONE query column (q = lane & 15): the row max / row sum reductions are
// 15 register ops + 2 shuffles, the O rescale is lane-local, and P goes from the S accumulators into
// the next MFMA's B operand with no LDS round trip (the k-order inside a 32-key step is permuted the
// same way on the V^T fragment).  K~ / P~ tiles are staged row-major + XOR swizzle (ds_read_b128,
// conflict-free); V is staged transposed with a padded row stride (ds_read_b64, conflict-free).
//
// SPLIT evaluates every product as hi*hi + lo*hi + hi*lo on bf16 hi/lo planes (f32-accurate mode).
#include <stdlib.h>

#include "cfm_common.h"
#include "attn_common.h"

struct AttnArgs {
    const void* q;
    const void* k;
    const void* v;
    const void* p;
    const float* bias_u;
    const float* bias_v;
    const uint8_t* mask;
    void* out;
    int64_t q_sb, q_st, k_sb, k_st, k_sh, v_sb, v_st, v_sh, p_sb, p_st, m_sb, m_sq;
    int B, H, Tq, Tk, dk;
    int q_dtype, kv_dtype, p_dtype, out_dtype;
    float scale;
    float* lse;      // optional f32 [B,H,Tq]: log-sum-exp of each row's scaled masked scores (training)
    CfmDrop drop;    // dropout on the probabilities (training): element ((b*H + h)*Tq + i)*Tk + j
};

namespace {

template <typename HT, bool HAS_POS, bool SPLIT>
__global__ __launch_bounds__(256) void cfm_attn_kernel(const AttnArgs a) {
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int KS_PLANE = KT * 8;                // u32x4 per K~ plane
    constexpr int VT_PLANE = DKP * VSTR;            // u16 per V^T plane
    __shared__ u32x4 Ks[KS_PLANE * NPL];
    __shared__ u32x4 Ps[HAS_POS ? KS_PLANE * NPL : 1];
    __shared__ __attribute__((aligned(16))) u16 Vt[VT_PLANE * NPL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * QT + wave * 16 + (lane & 15);
    const int qc = qi < a.Tq ? qi : a.Tq - 1;  // clamped for loads; stores are predicated on qi
    const int dk = a.dk;

    // ---- Q~ fragments (B operand of S^T = K~ . Q~^T): lane holds q=lane&15, d = kk*32 + g*8 + j ----
    u32x4 qu[2], qul[2], qv[HAS_POS ? 2 : 1], qvl[HAS_POS ? 2 : 1];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int d0 = kk * 32 + g * 8;
        float f[8], fu[8], fv[8];
        load8f(a.q, a.q_dtype, (int64_t)b * a.q_sb + (int64_t)qc * a.q_st + h * dk + d0, dk - d0, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = d0 + j < dk;
            fu[j] = f[j] + ((ok && a.bias_u) ? a.bias_u[h * dk + d0 + j] : 0.f);
            if constexpr (HAS_POS) fv[j] = f[j] + ((ok && a.bias_v) ? a.bias_v[h * dk + d0 + j] : 0.f);
        }
        pack_planes<HT, SPLIT>(fu, qu[kk], qul[kk]);
        if constexpr (HAS_POS) pack_planes<HT, SPLIT>(fv, qv[kk], qvl[kk]);
    }

    f32x4 acc_o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (a.Tk + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * KT;
        __syncthreads();  // previous tile fully consumed
        // ---- stage K~ (and P~) rows, V transposed ---------------------------------------------
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int id = pss * 256 + tid;
            const int key = id >> 3, c = id & 7;
            const int kj = k0 + key;
            const int d0 = c * 8;
            const int nv = kj < a.Tk ? dk - d0 : 0;
            float f[8];
            u32x4 hi, lo;
            load8f(a.k, a.kv_dtype, (int64_t)b * a.k_sb + (int64_t)h * a.k_sh + (int64_t)(kj < a.Tk ? kj : 0) * a.k_st + d0, nv, f);
            pack_planes<HT, SPLIT>(f, hi, lo);
            Ks[k_swz(key, c)] = hi;
            if constexpr (SPLIT) Ks[KS_PLANE + k_swz(key, c)] = lo;
            if constexpr (HAS_POS) {
                load8f(a.p, a.p_dtype, (int64_t)b * a.p_sb + (int64_t)(kj < a.Tk ? kj : 0) * a.p_st + h * dk + d0, nv, f);
                pack_planes<HT, SPLIT>(f, hi, lo);
                Ps[k_swz(key, c)] = hi;
                if constexpr (SPLIT) Ps[KS_PLANE + k_swz(key, c)] = lo;
            }
            load8f(a.v, a.kv_dtype, (int64_t)b * a.v_sb + (int64_t)h * a.v_sh + (int64_t)(kj < a.Tk ? kj : 0) * a.v_st + d0, nv, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (SPLIT) {
                    const u16 hb = BF16::from_f32(f[j]);
                    Vt[(d0 + j) * VSTR + key] = hb;
                    Vt[VT_PLANE + (d0 + j) * VSTR + key] = BF16::from_f32(f[j] - BF16::to_f32(hb));
                } else {
                    Vt[(d0 + j) * VSTR + key] = HT::from_f32(f[j]);
                }
            }
        }
        __syncthreads();

        // ---- S^T tile: 4 key fragments of 16 ---------------------------------------------------
        f32x4 s[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int idx = k_swz(f * 16 + (lane & 15), kk * 4 + g);
                const u32x4 kf = Ks[idx];
                if constexpr (SPLIT) {
                    s[f] = HT::mfma(kf, qul[kk], s[f]);
                    s[f] = HT::mfma(Ks[KS_PLANE + idx], qu[kk], s[f]);
                }
                s[f] = HT::mfma(kf, qu[kk], s[f]);
                if constexpr (HAS_POS) {
                    const u32x4 pf = Ps[idx];
                    if constexpr (SPLIT) {
                        s[f] = HT::mfma(pf, qvl[kk], s[f]);
                        s[f] = HT::mfma(Ps[KS_PLANE + idx], qv[kk], s[f]);
                    }
                    s[f] = HT::mfma(pf, qv[kk], s[f]);
                }
            }
        }
        // ---- scale, mask, online softmax (lane owns query qi; keys k0 + f*16 + g*4 + r) ---------
        float tmax = -INFINITY;
        float sv[4][4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kj = k0 + f * 16 + g * 4 + r;
                bool ok = kj < a.Tk;
                if (ok && a.mask) ok = a.mask[(int64_t)b * a.m_sb + (int64_t)qc * a.m_sq + kj] != 0;
                const float x = ok ? s[f][r] * a.scale : -INFINITY;
                sv[f][r] = x;
                tmax = fmaxf(tmax, x);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        float alpha = 1.f;
        if (m_new != -INFINITY) alpha = __expf(m_run - m_new);  // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (m_new == -INFINITY) ? 0.f : __expf(sv[f][r] - m_new);
                sv[f][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_o[i] *= alpha;
        if (a.drop.thresh) {                               // dropout(softmax(..)): the normaliser is the undropped row's
            const unsigned rowbase = (unsigned)(((int64_t)b * a.H + h) * a.Tq + qc) * (unsigned)a.Tk;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) sv[f][r] = cfm_drop(a.drop, rowbase + (unsigned)(k0 + f * 16 + g * 4 + r), sv[f][r]);
        }

        // ---- O^T += V^T . P^T over the tile's two 32-key steps ------------------------------------
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            float pf[8] = {sv[2 * k2][0], sv[2 * k2][1], sv[2 * k2][2], sv[2 * k2][3],
                           sv[2 * k2 + 1][0], sv[2 * k2 + 1][1], sv[2 * k2 + 1][2], sv[2 * k2 + 1][3]};
            u32x4 ph, pl;
            pack_planes<HT, SPLIT>(pf, ph, pl);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                const int d = fd * 16 + (lane & 15);
                const int o0 = d * VSTR + (2 * k2) * 16 + g * 4;
                const u32x2 v0 = *(const u32x2*)(Vt + o0);
                const u32x2 v1 = *(const u32x2*)(Vt + o0 + 16);
                const u32x4 vf = {v0.x, v0.y, v1.x, v1.y};
                if constexpr (SPLIT) {
                    const u32x2 w0 = *(const u32x2*)(Vt + VT_PLANE + o0);
                    const u32x2 w1 = *(const u32x2*)(Vt + VT_PLANE + o0 + 16);
                    const u32x4 vl = {w0.x, w0.y, w1.x, w1.y};
                    acc_o[fd] = HT::mfma(vf, pl, acc_o[fd]);
                    acc_o[fd] = HT::mfma(vl, ph, acc_o[fd]);
                }
                acc_o[fd] = HT::mfma(vf, ph, acc_o[fd]);
            }
        }
    }

    // ---- normalise and store: lane holds O[q=qi][d = fd*16 + g*4 + r] --------------------------------
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;  // fully masked row -> zeros (attention.py:92)
    if (a.lse && qi < a.Tq && g == 0) a.lse[((int64_t)b * a.H + h) * a.Tq + qi] = l_tot > 0.f ? m_run + logf(l_tot) : -INFINITY;
    if (qi < a.Tq) {
        const int64_t ob = ((int64_t)b * a.Tq + qi) * ((int64_t)a.H * dk) + (int64_t)h * dk;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            const int d = fd * 16 + g * 4;
            if (d >= dk) continue;  // dk % 4 == 0: the 4 columns are valid together
            const f32x4 o = acc_o[fd] * inv;
            if (a.out_dtype == CFM_F32)
                *(f32x4*)((float*)a.out + ob + d) = o;
            else if (a.out_dtype == CFM_BF16)
                *(u32x2*)((u16*)a.out + ob + d) = (u32x2){pack2<BF16>(o.x, o.y), pack2<BF16>(o.z, o.w)};
            else
                *(u32x2*)((u16*)a.out + ob + d) = (u32x2){pack2<F16>(o.x, o.y), pack2<F16>(o.z, o.w)};
        }
    }
}

// =============================================================================================
// v2: d_k == 64, 16-bit q/p, non-split.  Same maths and the same swapped MFMA orientation as cfm_attn_kernel, but
//   * K, V (and the per-key positional rows) of up to 256 keys are staged ONCE into LDS with 16-byte copies and one
//     barrier; the 4 wavefronts then walk the key tiles independently (v1: restage + 2 barriers per 64 keys);
//   * V stays ROW-major in LDS and is consumed column-wise with ds_read_b64_tr_b16 (the hardware transpose read) --
//     v1 transposed it with 2-byte LDS writes;
//   * the batch path's positional term (one p row per utterance) is a per-query constant:
//     bd_i = round(q_i + v) . p_b is one 64-long dot product per query, not an MFMA against a broadcast tile;
//   * masks are staged into LDS once (a key-validity row when there is no mask / a (B,1,Tk) mask; 64 rows for (B,Tq,Tk)).
// =============================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Phase stamps for scripts/probe_attn.hip (built with -DCFM_ATTN_STAMPS; never defined in the product build).
#ifdef CFM_ATTN_STAMPS
__device__ long long cfm_attn_stamps[2048 * 8];
#define CFM_ASTAMP(i) do { if (threadIdx.x == 0) { const int bid_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
        if (bid_ < 2048) cfm_attn_stamps[bid_ * 8 + (i)] = (i) >= 6 ? wall_clock64() : clock64(); } } while (0)
#else
#define CFM_ASTAMP(i) do { } while (0)
#endif

constexpr int SK = 256;        // keys resident per super-tile
constexpr int V2STR = 72;      // V row stride in 16-bit elements (144 B)
constexpr int MLSTR = SK + 4;  // mask row stride in bytes

// 8 consecutive elements -> 16-bit x8.  The element type is a template argument: a run-time branch around each load makes the
// compiler wait at every join, which serialised the 16 K/V loads of a thread (6 k cycles to ISSUE them, measured).
template <typename HT, bool F32>
__device__ __forceinline__ u32x4 load_row8(const void* base, unsigned off) {
    if constexpr (F32) {
        const f32x4 a = *(const f32x4*)((const float*)base + off);
        const f32x4 b = *(const f32x4*)((const float*)base + off + 4);
        return pack8<HT>(a, b);
    } else {
        return *(const u32x4*)((const u16*)base + off);
    }
}

template <typename HT, int PMODE, bool MFULL, bool KVF32, int NWQ = 4>
__device__ __forceinline__ void attn2_body(const AttnArgs& a, const int block_x) {
    constexpr int QTL = 16 * NWQ, NTH = 64 * NWQ, NST = SK * 8 / NTH;   // queries and threads per workgroup (NWQ wavefronts x 16 queries), staging pieces per thread
    __shared__ u32x4 Kl[SK * 8];
    __shared__ u32x4 Pl[PMODE == 2 ? SK * 8 : 1];
    __shared__ __attribute__((aligned(16))) u16 Vl[SK * V2STR];
    __shared__ __attribute__((aligned(16))) uint8_t Ml[(MFULL ? QTL : 1) * MLSTR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    // XCD-aware block order (1-D grid).  Workgroups go round-robin over the 8 XCDs, so the q-tiles of ONE (b, h) pair are given
    // ids 8 apart: they then share an L2 and K/V are fetched from memory once instead of once per q-tile (measured 41 MB of HBM
    // traffic per launch for 16 MB of compulsory bytes).  Groups of 8 pairs; padding blocks of the last group exit.
    const int nq = (a.Tq + QTL - 1) / QTL;
    const int grp = block_x / (8 * nq), rem = block_x % (8 * nq);
    const int pair = grp * 8 + rem % 8;
    if (pair >= a.B * a.H) return;                         // uniform, before any barrier
    const int b = pair / a.H, h = pair % a.H;
    const int q0 = (rem / 8) * QTL;
    const int qi = q0 + wave * 16 + l15;
    const int qc = qi < a.Tq ? qi : a.Tq - 1;
    constexpr int dk = 64;
    CFM_ASTAMP(0);
    CFM_ASTAMP(6);

    // ---- every global request of the first super-tile goes out before anything is computed: Q, biases, the positional
    //      row, up to 256 keys of K and V (and P), the mask bytes ------------------------------------------------------
    u32x4 qraw[2], praw[2];
    f32x4 bu[2][2], bv[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int d0 = kk * 32 + g * 8;
        qraw[kk] = *(const u32x4*)((const u16*)a.q + (int64_t)b * a.q_sb + (int64_t)qc * a.q_st + h * dk + d0);
        if constexpr (PMODE != 0) {
            bu[kk][0] = *(const f32x4*)(a.bias_u + h * dk + d0); bu[kk][1] = *(const f32x4*)(a.bias_u + h * dk + d0 + 4);
            bv[kk][0] = *(const f32x4*)(a.bias_v + h * dk + d0); bv[kk][1] = *(const f32x4*)(a.bias_v + h * dk + d0 + 4);
        }
        if constexpr (PMODE == 1) praw[kk] = *(const u32x4*)((const u16*)a.p + (int64_t)b * a.p_sb + h * dk + d0);
    }
    u32x4 kr[NST], vr[NST], pr[PMODE == 2 ? NST : 1];
    // (b, h) bases are wave-uniform (scalar 64-bit arithmetic); what varies per thread is a 32-bit element offset -- the host checks
    // that Tk * stride fits.  The first version did three 64-bit VALU multiply-adds per load: 66 of them per thread.
    using kv_t = std::conditional_t<KVF32, float, u16>;
    const kv_t* const kbase = (const kv_t*)a.k + ((int64_t)b * a.k_sb + (int64_t)h * a.k_sh);
    const kv_t* const vbase = (const kv_t*)a.v + ((int64_t)b * a.v_sb + (int64_t)h * a.v_sh);
    const u16* const pbase = PMODE == 2 ? (const u16*)a.p + ((int64_t)b * a.p_sb + (int64_t)h * dk) : nullptr;
    const unsigned kst = (unsigned)a.k_st, vst = (unsigned)a.v_st, pst = (unsigned)a.p_st;
    auto stage_load = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int id = i * NTH + tid;
            const int key = id >> 3, c = id & 7;
            const int kj = ks + key;
            const bool ok = kj < a.Tk;
            const unsigned kc = ok ? (unsigned)kj : 0u;
            kr[i] = load_row8<HT, KVF32>(kbase, kc * kst + c * 8);
            vr[i] = load_row8<HT, KVF32>(vbase, kc * vst + c * 8);
            if constexpr (PMODE == 2) pr[i] = *(const u32x4*)(pbase + (kc * pst + c * 8));
            if (!ok) {                                     // keys past Tk: clamped address, zeroed value (no branch around the load)
                kr[i] = (u32x4){0u, 0u, 0u, 0u};
                vr[i] = (u32x4){0u, 0u, 0u, 0u};
                if constexpr (PMODE == 2) pr[i] = (u32x4){0u, 0u, 0u, 0u};
            }
        }
    };
    stage_load(0);
    CFM_ASTAMP(1);

    // ---- Q~ fragments: q + u (and q + v) rounded to the MFMA operand type ----------------------------------------
    u32x4 qu[2], qv[2];
    float bd = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const u32x4 raw = qraw[kk];
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
        float f[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = HT::to_f32((u16)(w[i] & 0xffffu));
            f[2 * i + 1] = HT::to_f32((u16)(w[i] >> 16));
        }
        if constexpr (PMODE == 0) {
            qu[kk] = raw;
        } else {
            const f32x4 fa = {f[0], f[1], f[2], f[3]}, fb = {f[4], f[5], f[6], f[7]};
            qu[kk] = pack8<HT>(fa + bu[kk][0], fb + bu[kk][1]);
            qv[kk] = pack8<HT>(fa + bv[kk][0], fb + bv[kk][1]);
            if constexpr (PMODE == 1) {   // bd_i = (q_i + v) . p_b : the lane's 8 d-values of this kk
                const unsigned pw[4] = {praw[kk].x, praw[kk].y, praw[kk].z, praw[kk].w}, qw[4] = {qv[kk].x, qv[kk].y, qv[kk].z, qv[kk].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bd = fmaf(HT::to_f32((u16)(qw[i] & 0xffffu)), HT::to_f32((u16)(pw[i] & 0xffffu)), bd);
                    bd = fmaf(HT::to_f32((u16)(qw[i] >> 16)), HT::to_f32((u16)(pw[i] >> 16)), bd);
                }
            }
        }
    }
    if constexpr (PMODE == 1) {
        bd += __shfl_xor(bd, 16, 64);
        bd += __shfl_xor(bd, 32, 64);
    }

    f32x4 acc_o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int ks = 0; ks < a.Tk; ks += SK) {
        if (ks) {
            __syncthreads();  // everyone is done with the previous super-tile
            stage_load(ks);
        }
        // ---- stage up to 256 keys: the requests above, now the LDS writes ------------------------------------------
        {
            // mask bytes: a validity row (no mask / broadcast mask) or 64 query rows
            if constexpr (!MFULL) {
                const int kj = ks + tid;
                uint8_t mv = kj < a.Tk ? 1 : 0;
                if (mv && a.mask) mv = a.mask[(int64_t)b * a.m_sb + kj] != 0;
                if (tid < SK) Ml[tid] = mv;
            } else {
#pragma unroll 4
                for (int i = 0; i < QTL; ++i) {
                    const int kj = ks + (tid < SK ? tid : SK - 1);
                    const int qr = q0 + i < a.Tq ? q0 + i : a.Tq - 1;
                    if (tid < SK) Ml[i * MLSTR + tid] = kj < a.Tk ? (a.mask[(int64_t)b * a.m_sb + (int64_t)qr * a.m_sq + kj] != 0) : 0;
                }
            }
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int id = i * NTH + tid;
                const int key = id >> 3, c = id & 7;
                Kl[k_swz(key, c)] = kr[i];
                *(u32x4*)(Vl + key * V2STR + c * 8) = vr[i];
                if constexpr (PMODE == 2) Pl[k_swz(key, c)] = pr[i];
            }
        }
        __syncthreads();
        CFM_ASTAMP(2);

        // ---- S^T for all four 64-key tiles of the super-tile, ONE softmax pass over its 256 keys, then O^T += V^T . P^T.
        //      (Per 64-key tile the chain K read -> MFMA -> max -> shuffles -> exp -> pack -> MFMA is serial: 2.3 k cycles a
        //      tile; in bulk the 32 + 32 MFMAs and the 64 exponentials each run back to back.)  Keys past Tk are zero rows
        //      with mask byte 0, so short super-tiles go through the same code.
        f32x4 s[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int idx = k_swz(f * 16 + l15, kk * 4 + g);
                s[f] = HT::mfma(Kl[idx], qu[kk], s[f]);
                if constexpr (PMODE == 2) s[f] = HT::mfma(Pl[idx], qv[kk], s[f]);
            }
        }
        float tmax = -INFINITY;
        const uint8_t* mrow = Ml + (MFULL ? (wave * 16 + l15) * MLSTR : 0) + 4 * g;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const unsigned mb = *(const unsigned*)(mrow + f * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = ((mb >> (8 * r)) & 0xffu) != 0;
                const float x = ok ? (s[f][r] + bd) * a.scale : -INFINITY;
                s[f][r] = x;
                tmax = fmaxf(tmax, x);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        float alpha = 1.f;
        if (m_new != -INFINITY) alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (m_new == -INFINITY) ? 0.f : __expf(s[f][r] - m_new);
                s[f][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_o[i] *= alpha;
        if (a.drop.thresh) {
            const unsigned rowbase = (unsigned)(((int64_t)b * a.H + h) * a.Tq + qc) * (unsigned)a.Tk;
#pragma unroll
            for (int f = 0; f < 16; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][r] = cfm_drop(a.drop, rowbase + (unsigned)(ks + f * 16 + g * 4 + r), s[f][r]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const u32x4 ph = pack8<HT>(s[2 * k2], s[2 * k2 + 1]);
            const int key0 = k2 * 32 + 4 * g;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the 4x16 block; it receives column
                // (lane & 15) of the 4 rows, i.e. V[key0 + 0..3][fd*16 + l15]
                const u16* pa = Vl + (key0 + (l15 >> 2)) * V2STR + fd * 16 + (l15 & 3) * 4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 16 * V2STR));
                const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
                const u32x4 vf = {lo2.x, lo2.y, hi2.x, hi2.y};
                acc_o[fd] = HT::mfma(vf, ph, acc_o[fd]);
            }
        }
    }

    CFM_ASTAMP(3);
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = l_tot > 0.f ? __builtin_amdgcn_rcpf(l_tot) : 0.f;
    if (a.lse && qi < a.Tq && g == 0) a.lse[((int64_t)b * a.H + h) * a.Tq + qi] = l_tot > 0.f ? m_run + logf(l_tot) : -INFINITY;
    if (qi < a.Tq) {
        const int64_t ob = ((int64_t)b * a.Tq + qi) * ((int64_t)a.H * dk) + (int64_t)h * dk;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            const int d = fd * 16 + g * 4;
            const f32x4 o = acc_o[fd] * inv;
            if (a.out_dtype == CFM_F32)
                *(f32x4*)((float*)a.out + ob + d) = o;
            else if (a.out_dtype == CFM_BF16)
                *(u32x2*)((u16*)a.out + ob + d) = (u32x2){pack2<BF16>(o.x, o.y), pack2<BF16>(o.z, o.w)};
            else
                *(u32x2*)((u16*)a.out + ob + d) = (u32x2){pack2<F16>(o.x, o.y), pack2<F16>(o.z, o.w)};
        }
    }
    CFM_ASTAMP(4);
    CFM_ASTAMP(7);
}

template <typename HT, int PMODE, bool MFULL, bool KVF32>
__global__ __launch_bounds__(256, 2) void cfm_attn2_kernel(const AttnArgs a) {   // 2 wavefronts per SIMD: two workgroups share a CU
    attn2_body<HT, PMODE, MFULL, KVF32>(a, (int)blockIdx.x);
}

// The same body with 8 wavefronts = 128 queries per workgroup (one workgroup per CU, still 2 wavefronts per SIMD): half the workgroups to dispatch and
// the keys / values of a (b, h) pair staged twice instead of four times at T = 249 (launch_attn2 picks it when it still fills the chip)
template <typename HT, int PMODE>
__global__ __launch_bounds__(512, 1) void cfm_attn2_q128_kernel(const AttnArgs a) {
    attn2_body<HT, PMODE, false, false, 8>(a, (int)blockIdx.x);
}

// Several attention problems in ONE launch (cfm_attention_group): the micro-batches of a training window have different lengths T_g, so each
// is its own (B_g, H, T_g) problem with its own mask and strides -- but none of them fills the chip (8 x 4 x 4 tiles at a micro-batch of 8
// utterances), and as separate launches they ran one after the other.  Workgroup b belongs to the problem with first[i] <= b < first[i+1].
constexpr int ATTN_GROUP_MAX = 8;
struct AttnGroupArgs {
    AttnArgs a[ATTN_GROUP_MAX];
    int first[ATTN_GROUP_MAX + 1];
    int n;
};

template <typename HT, bool MFULL>
__global__ __launch_bounds__(256, 2) void cfm_attn2_group_kernel(const AttnGroupArgs G) {
    const int b = (int)blockIdx.x;
    int idx = 0;
#pragma unroll
    for (int i = 1; i < ATTN_GROUP_MAX; ++i)
        if (i < G.n && b >= G.first[i]) idx = i;            // uniform
    attn2_body<HT, 0, MFULL, false>(G.a[idx], b - G.first[idx]);
}

template <typename HT>
int launch_attn2(const AttnArgs& a, hipStream_t s, const char* name) {
    const int nq = (a.Tq + QT - 1) / QT, pairs = a.B * a.H;
    const dim3 grid((unsigned)(((pairs + 7) / 8) * 8 * nq)), block(256);      // groups of 8 (b,h) pairs x nq q-tiles (see the kernel)
    const double flops = 4.0 * a.B * a.H * (double)a.Tq * a.Tk * a.dk;
    const double bytes = 2.0 * a.B * a.H * ((double)a.Tq * 2 + (double)a.Tk * 2) * a.dk;
    CfmProfScope prof(name, s, flops, bytes);
    const int pmode = a.p ? (a.p_st == 0 ? 1 : 2) : 0;
    const bool mfull = a.mask && a.m_sq != 0;
    const bool kvf32 = a.kv_dtype == CFM_F32;
    // 128 queries per workgroup when that still gives every CU a workgroup (config 2 / config 4: 128 (b, h) pairs x 2 tiles = 256): half the workgroups to
    // dispatch, keys / values staged twice instead of four times per pair -- measured 1.217 -> 1.201 ms per config-2 step (A/B on one box, CFM_ATTN_Q128=0 for the
    // 64-query form); per query the same arithmetic in the same order: bit-identical
    static const bool q128 = getenv("CFM_ATTN_Q128") == nullptr || atoi(getenv("CFM_ATTN_Q128")) != 0;
    if (q128 && !mfull && !kvf32 && pmode != 2 && a.Tq > 64 && ((pairs + 7) / 8) * 8 * ((a.Tq + 127) / 128) >= 192) {
        const int nq8 = (a.Tq + 127) / 128;
        const dim3 grid8((unsigned)(((pairs + 7) / 8) * 8 * nq8)), block8(512);
        if (pmode == 0) CFM_LAUNCH((cfm_attn2_q128_kernel<HT, 0>), grid8, block8, 0, s, a);
        else CFM_LAUNCH((cfm_attn2_q128_kernel<HT, 1>), grid8, block8, 0, s, a);
        return cfm_launch_status(name);
    }
#define CFM_A2(PM)                                                                                       \
    do {                                                                                                 \
        if (mfull && kvf32) CFM_LAUNCH((cfm_attn2_kernel<HT, PM, true, true>), grid, block, 0, s, a);      \
        else if (mfull) CFM_LAUNCH((cfm_attn2_kernel<HT, PM, true, false>), grid, block, 0, s, a);         \
        else if (kvf32) CFM_LAUNCH((cfm_attn2_kernel<HT, PM, false, true>), grid, block, 0, s, a);         \
        else CFM_LAUNCH((cfm_attn2_kernel<HT, PM, false, false>), grid, block, 0, s, a);                   \
    } while (0)
    if (pmode == 0) CFM_A2(0);
    else if (pmode == 1) CFM_A2(1);
    else CFM_A2(2);
#undef CFM_A2
    return cfm_launch_status(name);
}

// new_cache[b,h,t,:] = [K_t | V_t]  (f32), rows t < Tc from the old cache, the rest from the new k/v.
__global__ void cfm_kv_pack_kernel(const float* old_cache, int Tc, const void* k, const void* v, int dt, int64_t k_sb,
                                   int64_t k_st, int64_t v_sb, int64_t v_st, float* out, int B, int H, int Tn, int dk) {
    const int Tk = Tc + Tn;
    const int64_t n = (int64_t)B * H * Tk * 2 * dk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % (2 * dk));
        int64_t r = i / (2 * dk);
        const int t = (int)(r % Tk);
        r /= Tk;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        float val;
        if (t < Tc) {
            val = old_cache[(((int64_t)b * H + h) * Tc + t) * (2 * dk) + e];
        } else if (e < dk) {
            val = load_as_f32(k, (int64_t)b * k_sb + (int64_t)(t - Tc) * k_st + h * dk + e, dt);
        } else {
            val = load_as_f32(v, (int64_t)b * v_sb + (int64_t)(t - Tc) * v_st + h * dk + (e - dk), dt);
        }
        out[i] = val;
    }
}

template <typename HT, bool SPLIT>
int launch_attn(const AttnArgs& a, bool has_pos, hipStream_t s, const char* name) {
    const dim3 grid((a.Tq + QT - 1) / QT, a.H, a.B), block(256);
    const double flops = 4.0 * a.B * a.H * (double)a.Tq * a.Tk * a.dk;
    const double bytes = 2.0 * a.B * a.H * ((double)a.Tq * 2 + (double)a.Tk * 2) * a.dk;
    CfmProfScope prof(name, s, flops, bytes);
    if (has_pos)
        CFM_LAUNCH((cfm_attn_kernel<HT, true, SPLIT>), grid, block, 0, s, a);
    else
        CFM_LAUNCH((cfm_attn_kernel<HT, false, SPLIT>), grid, block, 0, s, a);
    return cfm_launch_status(name);
}

}  // namespace

static int attn_args(const cfm_attn_desc* d, AttnArgs& a, bool& fast) {
    CFM_CHECK_ARG(d && d->q && d->k && d->v && d->out, "cfm_attention: null pointer");
    CFM_CHECK_ARG(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "cfm_attention: empty problem");
    CFM_CHECK_ARG(d->dk > 0 && d->dk <= DKP && d->dk % 4 == 0, "cfm_attention: need dk %% 4 == 0 and dk <= 64 (dk=%d)", d->dk);
    CFM_CHECK_ARG(d->B <= 65535 && d->H <= 65535, "cfm_attention: B,H must fit a grid dimension");
    CFM_CHECK_ARG(d->mma_dtype == CFM_BF16 || d->mma_dtype == CFM_F16, "cfm_attention: mma_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(!d->split || d->mma_dtype == CFM_BF16, "cfm_attention: split mode uses bf16 planes");
    CFM_CHECK_ARG(!d->p || (d->bias_u && d->bias_v), "cfm_attention: positional term needs bias_u and bias_v");
    a.q = d->q; a.k = d->k; a.v = d->v; a.p = d->p; a.bias_u = d->p ? d->bias_u : nullptr; a.bias_v = d->p ? d->bias_v : nullptr;
    a.mask = d->mask; a.out = d->out;
    a.q_sb = d->q_sb; a.q_st = d->q_st; a.k_sb = d->k_sb; a.k_st = d->k_st; a.k_sh = d->k_sh;
    a.v_sb = d->v_sb; a.v_st = d->v_st; a.v_sh = d->v_sh; a.p_sb = d->p_sb; a.p_st = d->p_st; a.m_sb = d->m_sb; a.m_sq = d->m_sq;
    a.B = d->B; a.H = d->H; a.Tq = d->Tq; a.Tk = d->Tk; a.dk = d->dk;
    a.q_dtype = d->q_dtype; a.kv_dtype = d->kv_dtype; a.p_dtype = d->p_dtype; a.out_dtype = d->out_dtype; a.scale = d->scale; a.lse = d->lse;
    CFM_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f, "cfm_attention: dropout probability must be in [0, 1)");
    CFM_CHECK_ARG(d->drop_p == 0.f || (int64_t)d->B * d->H * d->Tq * d->Tk < ((int64_t)1 << 32), "cfm_attention: dropout needs fewer than 2^32 score elements");
    a.drop = cfm_make_drop(d->drop_p, d->drop_seed);
    const bool pos = d->p != nullptr;
    // v2 fast path: d_k = 64, 16-bit q (and p) of the MFMA type, K/V either that type or f32 (streaming cache), 16-byte
    // aligned rows; everything else (f32-accurate split mode, d_k = 36 ...) runs the general v1 kernel.
    const bool al8 = (d->q_sb % 8 == 0) && (d->q_st % 8 == 0) && (d->k_sb % 4 == 0) && (d->k_st % 8 == 0) && (d->k_sh % 8 == 0) &&
                     (d->v_sb % 4 == 0) && (d->v_st % 8 == 0) && (d->v_sh % 8 == 0) && (!pos || ((d->p_sb % 8 == 0) && (d->p_st % 8 == 0)));
    fast = !d->split && d->dk == 64 && d->q_dtype == d->mma_dtype && (!pos || d->p_dtype == d->mma_dtype) &&
           (d->kv_dtype == d->mma_dtype || d->kv_dtype == CFM_F32) && al8 && !getenv("CFM_ATTN_V1") &&
           (int64_t)d->Tk * d->k_st < ((int64_t)1 << 31) && (int64_t)d->Tk * d->v_st < ((int64_t)1 << 31) && (int64_t)d->Tk * d->p_st < ((int64_t)1 << 31);
    return CFM_OK;
}

extern "C" int cfm_attention(const cfm_attn_desc* d, cfm_stream_t stream) {
    AttnArgs a;
    bool fast = false;
    if (int rc = attn_args(d, a, fast)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool pos = d->p != nullptr;
    if (fast) {
        if (d->mma_dtype == CFM_BF16) return launch_attn2<BF16>(a, s, pos ? "attn2_rel_bf16" : "attn2_bf16");
        return launch_attn2<F16>(a, s, pos ? "attn2_rel_f16" : "attn2_f16");
    }
    if (d->split) return launch_attn<BF16, true>(a, pos, s, pos ? "attn_rel_bf16x3" : "attn_bf16x3");
    if (d->mma_dtype == CFM_BF16) return launch_attn<BF16, false>(a, pos, s, pos ? "attn_rel_bf16" : "attn_bf16");
    return launch_attn<F16, false>(a, pos, s, pos ? "attn_rel_f16" : "attn_f16");
}

// n attention problems (the micro-batches of a training window: different B, T, mask) in one launch when every one takes the d_k = 64 fast
// path without a positional term, with 16-bit K/V and the same kind of mask; otherwise one launch each, in order.  Same results either way.
extern "C" int cfm_attention_group(const cfm_attn_desc* descs, int32_t n, cfm_stream_t stream) {
    CFM_CHECK_ARG(descs && n > 0, "cfm_attention_group: no problems");
    hipStream_t s = (hipStream_t)stream;
    AttnGroupArgs G;
    bool groupable = n >= 2 && n <= ATTN_GROUP_MAX;
    bool mfull0 = false;
    int first = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n && groupable; ++i) {
        bool fast = false;
        if (int rc = attn_args(&descs[i], G.a[i], fast)) return rc;
        const AttnArgs& a = G.a[i];
        const bool mfull = a.mask && a.m_sq != 0;
        if (i == 0) mfull0 = mfull;
        groupable = fast && !a.p && a.kv_dtype != CFM_F32 && mfull == mfull0 && descs[i].mma_dtype == descs[0].mma_dtype;
        G.first[i] = first;
        const int nq = (a.Tq + QT - 1) / QT, pairs = a.B * a.H;
        first += ((pairs + 7) / 8) * 8 * nq;
        flops += 4.0 * a.B * a.H * (double)a.Tq * a.Tk * a.dk;
        bytes += 2.0 * a.B * a.H * ((double)a.Tq * 2 + (double)a.Tk * 2) * a.dk;
    }
    if (!groupable) {
        for (int i = 0; i < n; ++i)
            if (int rc = cfm_attention(&descs[i], stream)) return rc;
        return CFM_OK;
    }
    G.n = n;
    for (int i = n; i <= ATTN_GROUP_MAX; ++i) G.first[i] = first;
    for (int i = n; i < ATTN_GROUP_MAX; ++i) G.a[i] = G.a[0];
    const bool bf = descs[0].mma_dtype == CFM_BF16;
    CfmProfScope prof(bf ? "attn2_group_bf16" : "attn2_group_f16", s, flops, bytes);
    const dim3 grid((unsigned)first), block(256);
    if (bf && mfull0) CFM_LAUNCH((cfm_attn2_group_kernel<BF16, true>), grid, block, 0, s, G);
    else if (bf) CFM_LAUNCH((cfm_attn2_group_kernel<BF16, false>), grid, block, 0, s, G);
    else if (mfull0) CFM_LAUNCH((cfm_attn2_group_kernel<F16, true>), grid, block, 0, s, G);
    else CFM_LAUNCH((cfm_attn2_group_kernel<F16, false>), grid, block, 0, s, G);
    return cfm_launch_status("cfm_attention_group");
}

extern "C" int cfm_kv_cache_pack(const float* old_cache, int32_t Tc, const void* k, const void* v, int32_t kv_dtype,
                                 int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float* new_cache, int32_t B,
                                 int32_t H, int32_t Tn, int32_t dk, cfm_stream_t stream) {
    CFM_CHECK_ARG(k && v && new_cache, "cfm_kv_cache_pack: null pointer");
    CFM_CHECK_ARG(Tc == 0 || old_cache, "cfm_kv_cache_pack: cache_T > 0 needs old_cache");
    CFM_CHECK_ARG(B > 0 && H > 0 && Tn > 0 && dk > 0 && Tc >= 0, "cfm_kv_cache_pack: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * H * (Tc + Tn) * 2 * dk;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    CfmProfScope prof("kv_cache_pack", s, 0.0, (double)n * 8);
    CFM_LAUNCH(cfm_kv_pack_kernel, dim3(blocks), dim3(256), 0, s, old_cache, Tc, k, v, kv_dtype, k_sb, k_st, v_sb, v_st,
                       new_cache, B, H, Tn, dk);
    return cfm_launch_status("cfm_kv_cache_pack");
}
