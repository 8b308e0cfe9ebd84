// attn_common.h -- tile constants and small staging helpers shared by attention.hip (forward) and attention_bwd.hip (backward).
#pragma once
#include "cfm_common.h"

namespace {

constexpr int QT = 64;    // queries per workgroup
constexpr int KT = 64;    // keys per tile
constexpr int DKP = 64;   // padded head dim
constexpr int VSTR = 68;  // V^T row stride in 16-bit elements (136 B: conflict-free ds_read_b64)

// 8 consecutive elements starting at element offset `off`, as f32; elements >= nvalid read as 0.
__device__ __forceinline__ void load8f(const void* base, int dt, int64_t off, int nvalid, float (&o)[8]) {
    if (nvalid >= 8 && dt == CFM_F32 && (off & 3) == 0) {
        const f32x4 a = *(const f32x4*)((const float*)base + off);
        const f32x4 b = *(const f32x4*)((const float*)base + off + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
        return;
    }
    if (nvalid >= 8 && dt != CFM_F32 && (off & 7) == 0) {
        const u32x4 r = *(const u32x4*)((const u16*)base + off);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u16 lo = (u16)(w[i] & 0xffffu), hi = (u16)(w[i] >> 16);
            o[2 * i] = dt == CFM_BF16 ? BF16::to_f32(lo) : F16::to_f32(lo);
            o[2 * i + 1] = dt == CFM_BF16 ? BF16::to_f32(hi) : F16::to_f32(hi);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = i < nvalid ? load_as_f32(base, off + i, dt) : 0.f;
}

template <typename HT, bool SPLIT>
__device__ __forceinline__ void pack_planes(const float (&f)[8], u32x4& hi, u32x4& lo) {
    const f32x4 a = {f[0], f[1], f[2], f[3]}, b = {f[4], f[5], f[6], f[7]};
    if constexpr (SPLIT) {
        split8(a, b, hi, lo);
    } else {
        hi = pack8<HT>(a, b);
        lo = hi;
    }
}

__device__ __forceinline__ int k_swz(int row, int c) { return row * 8 + (c ^ ((row >> 1) & 7)); }

}  // namespace
