// joint.hip -- the transducer joint's broadcast-add + tanh (gfx950).
//
// replaces, in TransducerJoint.forward (reference src/joint.py:31-37):
//     out = enc_out.unsqueeze(2) + pred_out.unsqueeze(1);  out = tanh(out)          [B, T, U, J]
// which is the activation operand of the vocabulary projection ffn_out (cfm_gemm, M = B*T*U rows).  Written ONCE as a 16-bit
// (f32 in the accurate mode) row-major [B*T*U, J] operand: 167 MB at BASELINE config 4 (B 16, T' 249, U 41, J 512) next to the
// 3.3 GB of f32 logits the GEMM writes.  Fusing the tanh into the GEMM's A-tile loader instead would evaluate it once per N tile --
// 40 times at V = 5002, ~14 k VALU/transcendental cycles per wavefront against 4 k cycles of MFMA per 128 x 128 x 512 tile -- and the
// XCD-aware tile order of cfm_gemm already makes the 40 N tiles of one M tile read these rows from one L2.
//
// HBM-bound: per output row J*2 bytes written; the projected encoder / predictor rows (B*T*J and B*U*J f32, a few MB) stay in L2.
#include "cfm_common.h"

namespace {

// tanh(x) = 1 - 2 / (1 + e^{2x}) on v_exp_f32 / v_rcp_f32: absolute error ~1e-7, exact limits at +-inf
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <typename OT>
__global__ __launch_bounds__(256) void cfm_joint_act_kernel(const float* __restrict__ enc, int64_t ld_e, const float* __restrict__ pred,
                                                            int64_t ld_p, void* __restrict__ out, int T, int U, int J, int64_t chunks) {
    const int cpr = J >> 3;                                // 8-column chunks per row
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < chunks; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cpr;                       // output row (b, t, u)
        const int c = (int)(idx - m * cpr) * 8;
        const int64_t bt = m / U;                          // encoder row b*T + t
        const int u = (int)(m - bt * U);
        const int64_t b = bt / T;
        const float* e = enc + bt * ld_e + c;
        const float* p = pred + (b * U + u) * ld_p + c;
        const f32x4 e0 = *(const f32x4*)e, e1 = *(const f32x4*)(e + 4);
        const f32x4 p0 = *(const f32x4*)p, p1 = *(const f32x4*)(p + 4);
        f32x4 v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v0[r] = tanh_fast(e0[r] + p0[r]);
            v1[r] = tanh_fast(e1[r] + p1[r]);
        }
        if constexpr (std::is_same<OT, float>::value) {
            float* o = (float*)out + m * J + c;
            *(f32x4*)o = v0;
            *(f32x4*)(o + 4) = v1;
        } else {
            *(u32x4*)((u16*)out + m * J + c) = pack8<OT>(v0, v1);
        }
    }
}

}  // namespace

extern "C" int cfm_joint_act(const float* enc, int64_t ld_e, const float* pred, int64_t ld_p, void* out, int32_t out_dtype, int32_t B,
                             int32_t T, int32_t U, int32_t J, cfm_stream_t stream) {
    CFM_CHECK_ARG(enc && pred && out, "cfm_joint_act: null pointer");
    CFM_CHECK_ARG(B > 0 && T > 0 && U > 0 && J > 0 && J % 8 == 0, "cfm_joint_act: bad shape B=%d T=%d U=%d J=%d (J %% 8 == 0)", B, T, U, J);
    CFM_CHECK_ARG(ld_e >= J && ld_p >= J && ld_e % 4 == 0 && ld_p % 4 == 0, "cfm_joint_act: row strides must be >= J and multiples of 4");
    CFM_CHECK_ARG(out_dtype >= CFM_F32 && out_dtype <= CFM_F16, "cfm_joint_act: bad out_dtype");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * T * U, chunks = rows * (J / 8);
    const int64_t want = (chunks + 255) / 256;
    const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);      // grid-stride beyond 16 workgroups per CU
    CfmProfScope prof("joint_act", s, 0.0, (double)rows * J * cfm_elt_size(out_dtype));
    if (out_dtype == CFM_F32) CFM_LAUNCH(cfm_joint_act_kernel<float>, dim3(grid), dim3(256), 0, s, enc, ld_e, pred, ld_p, out, T, U, J, chunks);
    else if (out_dtype == CFM_BF16) CFM_LAUNCH(cfm_joint_act_kernel<BF16>, dim3(grid), dim3(256), 0, s, enc, ld_e, pred, ld_p, out, T, U, J, chunks);
    else CFM_LAUNCH(cfm_joint_act_kernel<F16>, dim3(grid), dim3(256), 0, s, enc, ld_e, pred, ld_p, out, T, U, J, chunks);
    return cfm_launch_status("cfm_joint_act");
}
