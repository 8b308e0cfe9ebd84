// norm.hip -- row LayerNorm for the conformer block, one 64-lane wavefront per row.
//
// HBM-bound (2*M*D*elt algorithmic bytes): each row is read once with 16-byte loads, held in
// registers, reduced with wavefront shuffles (no LDS), and written once.  Two norms can be chained
// in registers (norm_final of block i feeding norm_ff_macaron of block i+1, encoder_layer.py:70,57),
// and a row mask can zero padded frames on the way out (the masked_fill of convolution.py:36-37).
#include "cfm_common.h"

namespace {

__device__ __forceinline__ void store4(void* base, int dt, int64_t off, const f32x4& v) {
    if (dt == CFM_F32) {
        *(f32x4*)((float*)base + off) = v;
    } else if (dt == CFM_BF16) {
        *(u32x2*)((u16*)base + off) = (u32x2){pack2<BF16>(v.x, v.y), pack2<BF16>(v.z, v.w)};
    } else {
        *(u32x2*)((u16*)base + off) = (u32x2){pack2<F16>(v.x, v.y), pack2<F16>(v.z, v.w)};
    }
}

template <int ITERS>
__device__ __forceinline__ void norm_inplace(f32x4 (&v)[ITERS], int lane, int D, const float* g, const float* b, float eps) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < D) s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < D) {
            const f32x4 d = v[it] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < D) {
            const f32x4 gg = *(const f32x4*)(g + c);
            const f32x4 bb = *(const f32x4*)(b + c);
            v[it] = (v[it] - mean) * rstd * gg + bb;
        }
    }
}

template <int ITERS>
__global__ __launch_bounds__(256) void cfm_layernorm_kernel(const float* __restrict__ x, const float* g1, const float* b1,
                                                            void* out1, int dt1, const float* g2, const float* b2,
                                                            void* out2, int dt2, const uint8_t* mask, float eps,
                                                            int64_t M, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (lane + 64 * it) * 4;
        v[it] = c < D ? *(const f32x4*)(x + row * D + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    norm_inplace<ITERS>(v, lane, D, g1, b1, eps);
    if (out1) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < D) store4(out1, dt1, row * D + c, v[it]);
        }
    }
    if (out2) {
        if (g2) norm_inplace<ITERS>(v, lane, D, g2, b2, eps);
        const bool keep = mask ? mask[row] != 0 : true;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (lane + 64 * it) * 4;
            if (c < D) store4(out2, dt2, row * D + c, keep ? v[it] : (f32x4){0.f, 0.f, 0.f, 0.f});
        }
    }
}

}  // namespace

extern "C" int cfm_layernorm(const float* x, const float* g1, const float* b1, void* out1, int out1_dtype, const float* g2,
                             const float* b2, void* out2, int out2_dtype, const uint8_t* row_mask, float eps, int64_t M,
                             int32_t D, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && g1 && b1, "cfm_layernorm: null input");
    CFM_CHECK_ARG(out1 || out2, "cfm_layernorm: no output requested");
    CFM_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "cfm_layernorm: need D %% 4 == 0 and D <= 2048 (M=%lld D=%d)",
                  (long long)M, D);
    CFM_CHECK_ARG(!g2 || (b2 && out2), "cfm_layernorm: second norm needs b2 and out2");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    const double bytes = (double)M * D * (4.0 + (out1 ? cfm_elt_size(out1_dtype) : 0) + (out2 ? cfm_elt_size(out2_dtype) : 0));
    CfmProfScope prof("layernorm", s, 0.0, bytes);
#define CFM_LN_LAUNCH(IT)                                                                                                \
    CFM_LAUNCH((cfm_layernorm_kernel<IT>), grid, block, 0, s, x, g1, b1, out1, out1_dtype, g2, b2, out2, out2_dtype, \
                       row_mask, eps, M, D)
    if (D <= 256)
        CFM_LN_LAUNCH(1);
    else if (D <= 512)
        CFM_LN_LAUNCH(2);
    else if (D <= 1024)
        CFM_LN_LAUNCH(4);
    else
        CFM_LN_LAUNCH(8);
#undef CFM_LN_LAUNCH
    return cfm_launch_status("cfm_layernorm");
}
