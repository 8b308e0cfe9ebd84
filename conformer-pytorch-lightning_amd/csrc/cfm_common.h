// cfm_common.h -- shared device/host helpers for libconformer_gfx950 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/cfm.h"

// ---------------------------------------------------------------------------------------------
// host side: error text + launch check + profiling hooks (abi.cpp)
// ---------------------------------------------------------------------------------------------
int cfm_fail(int code, const char* fmt, ...);

struct CfmProfScope {  // one kernel launch; when profiling is on, CFM_LAUNCH hands its two events to the launch itself
    CfmProfScope(const char* name, hipStream_t s, double flops, double bytes);
    ~CfmProfScope();
    void* rec;
    hipStream_t stream;
    hipEvent_t ev_start, ev_stop;
};

// Launch inside a function that declared `CfmProfScope prof(...)`.  With profiling on, the start/stop events are attached to the
// dispatch itself (hipExtLaunchKernelGGL): they carry the kernel's own begin/end timestamps -- what rocprofv3 reports as its
// duration -- instead of bracketing it with separately recorded events (which adds ~2 us of event processing per launch).
#define CFM_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                     \
    do {                                                                                                                     \
        if (prof.rec) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, prof.ev_start, prof.ev_stop, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                               \
    } while (0)

#define CFM_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return cfm_fail(CFM_ERR_ARG, __VA_ARGS__);    \
    } while (0)

static inline int cfm_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfm_fail(CFM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return CFM_OK;
}

static inline int cfm_elt_size(int dt) { return dt == CFM_F32 ? 4 : 2; }
static inline bool cfm_is16(int dt) { return dt == CFM_BF16 || dt == CFM_F16; }

// ---------------------------------------------------------------------------------------------
// device side: 16-bit operand types.  A wavefront is 64 lanes on gfx950; every constant below
// assumes that.
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

struct BF16 {
    static constexpr int kId = CFM_BF16;
    typedef bf16x8_t frag;
    static __device__ __forceinline__ float to_f32(u16 v) { return __uint_as_float(((unsigned)v) << 16); }
    static __device__ __forceinline__ u16 from_f32(float f) {
        __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        return __builtin_bit_cast(u16, b);
    }
    static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c,
                                                        0, 0, 0);
    }
};

struct F16 {
    static constexpr int kId = CFM_F16;
    typedef f16x8_t frag;
    static __device__ __forceinline__ float to_f32(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ u16 from_f32(float f) {
        _Float16 h = (_Float16)f;
        return __builtin_bit_cast(u16, h);
    }
    static __device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c,
                                                       0, 0, 0);
    }
};

template <typename HT>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)HT::from_f32(lo) | ((unsigned)HT::from_f32(hi) << 16);
}

// 8 consecutive f32 -> one 16-byte fragment of 16-bit values
template <typename HT>
__device__ __forceinline__ u32x4 pack8(const f32x4& a, const f32x4& b) {
    u32x4 r;
    r.x = pack2<HT>(a.x, a.y);
    r.y = pack2<HT>(a.z, a.w);
    r.z = pack2<HT>(b.x, b.y);
    r.w = pack2<HT>(b.z, b.w);
    return r;
}

// hi/lo bf16 split of 8 f32 values:  x ~= hi + lo  with ~16 mantissa bits
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u16 hb = BF16::from_f32(v[i]);
        float r = v[i] - BF16::to_f32(hb);
        h[i] = hb;
        l[i] = BF16::from_f32(r);
    }
    hi = (u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    lo = (u32x4){l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
}

// v_exp_f32 + v_rcp_f32 (1 ulp): the IEEE divide sequence costs ~10 VALU ops per element in a GEMM epilogue
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }

// ---------------------------------------------------------------------------------------------
// dropout: a counter-based generator -- keep(seed, element index) is a pure function, so the backward regenerates the forward's
// mask instead of reading it from memory (nothing of the size of an activation is stored for dropout).  murmur3's 32-bit finaliser
// over (index * golden ratio) ^ seed: 2 multiplies, 3 shifts, 4 xors per element.  No bit-parity with torch's Philox stream (SURVEY 2.2:
// parity is defined at p = 0); statistics and forward/backward consistency are tested.
// ---------------------------------------------------------------------------------------------
struct CfmDrop {
    unsigned seed, thresh;   // keep iff hash >= thresh;  thresh = p * 2^32 (0: dropout off)
    float scale;             // 1 / (1 - p)
};
__device__ __forceinline__ unsigned cfm_hash32(unsigned seed, unsigned idx) {
    unsigned h = (idx * 0x9E3779B1u) ^ seed;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float cfm_drop(const CfmDrop& d, unsigned idx, float v) {
    return cfm_hash32(d.seed, idx) >= d.thresh ? v * d.scale : 0.f;
}
static inline CfmDrop cfm_make_drop(float p, unsigned seed) {
    CfmDrop d;
    d.seed = seed;
    d.thresh = p <= 0.f ? 0u : (p >= 1.f ? 0xFFFFFFFFu : (unsigned)((double)p * 4294967296.0));
    d.scale = p <= 0.f ? 1.f : (p >= 1.f ? 0.f : 1.f / (1.f - p));
    return d;
}

// generic scalar load/store by runtime dtype (slow paths, edges)
__device__ __forceinline__ float load_as_f32(const void* p, int64_t i, int dt) {
    if (dt == CFM_F32) return ((const float*)p)[i];
    u16 v = ((const u16*)p)[i];
    return dt == CFM_BF16 ? BF16::to_f32(v) : F16::to_f32(v);
}
__device__ __forceinline__ void store_from_f32(void* p, int64_t i, int dt, float v) {
    if (dt == CFM_F32)
        ((float*)p)[i] = v;
    else
        ((u16*)p)[i] = dt == CFM_BF16 ? BF16::from_f32(v) : F16::from_f32(v);
}

// Sum over the 64 lanes of a wavefront, result in every lane (EXEC must be all ones).  Four DPP adds fold each row of 16
// lanes (quad swaps, then half-row and row mirrors), four v_readlane pick up the row totals: ~40 clk, against six dependent
// ds_bpermute round trips (~60 clk each) for the shuffle version -- LayerNorm calls this twice per row.
template <int CTRL>
__device__ __forceinline__ float dpp_mov_(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov_<0xB1>(v);                                // quad_perm [1,0,3,2]
    v += dpp_mov_<0x4E>(v);                                // quad_perm [2,3,0,1]
    v += dpp_mov_<0x141>(v);                               // row_half_mirror
    v += dpp_mov_<0x140>(v);                               // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
