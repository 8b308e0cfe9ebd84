// stream.hip -- per-stream state for the batched streaming step (beyond the reference, whose forward_chunk
// serves one stream at a time, src/encoder.py:78-123, and rebuilds its attention cache with cat + slice every step, :117).
//
// B independent streams advance together, each at ITS OWN position: stream b has consumed offset[b] encoder frames of its utterance
// (int32 on the device, so a captured HIP graph replays without host-side scalars).  Per layer the keys / values of the last
// `need` frames and of the current chunk live in a ring buffer  kv_ring f32 [B, H, ring_T, 2*dk],  ring_T = need + chunk: frame f of a
// stream sits in slot f mod ring_T, a step writes only its `chunk` new frames, and nothing is ever moved.  Attention runs over all
// ring_T slots with a per-stream slot mask (the softmax is a sum over a SET of keys: their order in memory is irrelevant up to the
// rounding of the running sums) and per-(stream, slot) positional rows, so item b of the result equals the reference's batch-1
// forward_chunk on that stream with that stream's cache and offset (to the precision mode's tolerance).
//
//   cfm_stream_prep     offsets -> slot mask [B, ring_T], positional rows pe[frame in slot] [B, ring_T, D], optional absolute rows [B, D]
//   cfm_kv_ring_write   this step's K / V rows (from the fused QKV projection) -> their slots of the ring
//   cfm_stream_advance  offset[b] += chunk for active streams
//   cfm_dwconv_causal_bn_silu / cfm_conv_cache_update   the OPT-IN causal depthwise convolution with a (ktaps-1)-frame left context per
//                       stream (the reference has no causal mode and ignores its cnn_cache, convolution.py:34-39: off = parity)
#include "cfm_common.h"

namespace {

// the frame held by slot s when the newest frame of the window is `last`: the largest f <= last with f = s (mod ring_T)
__device__ __forceinline__ int slot_frame(int last, int s, int ring_T) {
    int r = (last - s) % ring_T;
    if (r < 0) r += ring_T;
    return last - r;
}

__global__ void cfm_stream_prep_kernel(const int* __restrict__ offsets, int B, int T, int need, int ring_T, const float* __restrict__ pe, int max_len,
                                       int D, uint8_t* __restrict__ slot_mask, float* __restrict__ pos_rows, float* __restrict__ abs_rows) {
    const int b = blockIdx.y, s = blockIdx.x;
    const int off = offsets[b];
    const int cached = off < need ? off : need;
    if (s < ring_T) {
        const int f = slot_frame(off + T - 1, s, ring_T);
        const bool valid = f >= off - cached && f >= 0;
        if (threadIdx.x == 0) slot_mask[(int64_t)b * ring_T + s] = valid ? 1 : 0;
        // f < max_len is the caller's job (encoder.StreamingBatch extends the table before a stream reaches its end); the clamp only keeps a
        // misuse of the C entry point inside the allocation
        const int row = valid ? (f < max_len ? f : max_len - 1) : 0;
        for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4)
            *(f32x4*)(pos_rows + ((int64_t)b * ring_T + s) * D + c) = *(const f32x4*)(pe + (int64_t)row * D + c);
    } else if (abs_rows) {                                  // one extra block per stream: the absolute-encoding row pe[offset] (attention.py:119-120)
        const int row = off < max_len ? off : max_len - 1;
        for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) *(f32x4*)(abs_rows + (int64_t)b * D + c) = *(const f32x4*)(pe + (int64_t)row * D + c);
    }
}

// ring[b, h, (offset[b] + t) mod ring_T, 0:dk] = K_t, [dk:2dk] = V_t for the T new frames
__global__ void cfm_kv_ring_write_kernel(const void* __restrict__ k, const void* __restrict__ v, int dt, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st,
                                         float* __restrict__ ring, const int* __restrict__ offsets, int B, int H, int T, int dk, int ring_T) {
    const int64_t n = (int64_t)B * T * H * 2 * dk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % (2 * dk));
        int64_t r = i / (2 * dk);
        const int h = (int)(r % H);
        r /= H;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        const int slot = (offsets[b] + t) % ring_T;
        const float val = e < dk ? load_as_f32(k, (int64_t)b * k_sb + (int64_t)t * k_st + h * dk + e, dt)
                                 : load_as_f32(v, (int64_t)b * v_sb + (int64_t)t * v_st + h * dk + (e - dk), dt);
        ring[(((int64_t)b * H + h) * ring_T + slot) * (2 * dk) + e] = val;
    }
}

__global__ void cfm_stream_advance_kernel(int* offsets, const uint8_t* __restrict__ active, int B, int T) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (!active || active[b])) offsets[b] += T;
}

// y[b,t,d] = silu( (sum_k w[d,k] xx[b, t+k, d] + bias[d]) * scale[d] + shift[d] ),  xx = [cache (K-1 frames) | x (T frames)]
__global__ void cfm_dwconv_causal_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ cache, const float* __restrict__ w,
                                         const float* __restrict__ bias, const float* __restrict__ sc, const float* __restrict__ sh, void* __restrict__ y, int y_dt,
                                         int B, int T, int D, int K) {
    const int64_t n = (int64_t)B * T * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int t = (int)((i / D) % T);
        const int b = (int)(i / ((int64_t)D * T));
        float a = 0.f;
        for (int k = 0; k < K; ++k) {
            const int tt = t + k - (K - 1);                 // position in x; negative: the left context
            float xv;
            if (tt >= 0) xv = load_as_f32(x, ((int64_t)b * T + tt) * D + d, x_dt);
            else xv = cache ? cache[((int64_t)b * (K - 1) + (K - 1 + tt)) * D + d] : 0.f;
            a = fmaf(w[(int64_t)d * K + k], xv, a);
        }
        store_from_f32(y, i, y_dt, siluf_((a + bias[d]) * sc[d] + sh[d]));
    }
}

// cache <- the last K-1 frames of [cache | x], in place: a thread owns one (stream, channel) column and walks its rows upwards, so a row
// is overwritten only after every later read of it (new row j reads old row j + T > j)
__global__ void cfm_conv_cache_update_kernel(const void* __restrict__ x, int x_dt, float* __restrict__ cache, int B, int T, int D, int K) {
    const int C = K - 1;
    const int64_t n = (int64_t)B * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int b = (int)(i / D);
        for (int j = 0; j < C; ++j) {
            const int tt = j + T - C;                       // position in x of new cache row j
            cache[((int64_t)b * C + j) * D + d] = tt >= 0 ? load_as_f32(x, ((int64_t)b * T + tt) * D + d, x_dt) : cache[((int64_t)b * C + (C + tt)) * D + d];
        }
    }
}

inline int blocks_for(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int cfm_stream_prep(const int32_t* offsets, int32_t B, int32_t T, int32_t need, int32_t ring_T, const float* pe, int32_t max_len, int32_t D,
                               uint8_t* slot_mask, float* pos_rows, float* abs_rows, cfm_stream_t stream) {
    CFM_CHECK_ARG(offsets && pe && slot_mask && pos_rows, "cfm_stream_prep: null pointer");
    CFM_CHECK_ARG(B > 0 && B <= 65535 && T > 0 && need >= 0 && ring_T >= need + T && D > 0 && D % 4 == 0 && max_len > 0,
                  "cfm_stream_prep: bad shape B=%d T=%d need=%d ring_T=%d D=%d (ring_T >= need + T)", B, T, need, ring_T, D);
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("stream_prep", s, 0.0, (double)B * ring_T * D * 8);
    CFM_LAUNCH(cfm_stream_prep_kernel, dim3((unsigned)(ring_T + (abs_rows ? 1 : 0)), (unsigned)B), dim3(64), 0, s, offsets, B, T, need, ring_T, pe, max_len, D,
               slot_mask, pos_rows, abs_rows);
    return cfm_launch_status("cfm_stream_prep");
}

extern "C" int cfm_kv_ring_write(const void* k, const void* v, int32_t kv_dtype, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float* ring,
                                 const int32_t* offsets, int32_t B, int32_t H, int32_t T, int32_t dk, int32_t ring_T, cfm_stream_t stream) {
    CFM_CHECK_ARG(k && v && ring && offsets, "cfm_kv_ring_write: null pointer");
    CFM_CHECK_ARG(B > 0 && H > 0 && T > 0 && dk > 0 && ring_T >= T, "cfm_kv_ring_write: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * T * H * 2 * dk;
    CfmProfScope prof("kv_ring_write", s, 0.0, (double)n * (4 + cfm_elt_size(kv_dtype)));
    CFM_LAUNCH(cfm_kv_ring_write_kernel, dim3((unsigned)blocks_for(n)), dim3(256), 0, s, k, v, kv_dtype, k_sb, k_st, v_sb, v_st, ring, offsets, B, H, T, dk, ring_T);
    return cfm_launch_status("cfm_kv_ring_write");
}

extern "C" int cfm_stream_advance(int32_t* offsets, const uint8_t* active, int32_t B, int32_t T, cfm_stream_t stream) {
    CFM_CHECK_ARG(offsets && B > 0 && T > 0, "cfm_stream_advance: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("stream_advance", s, 0.0, (double)B * 8);
    CFM_LAUNCH(cfm_stream_advance_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, offsets, active, B, T);
    return cfm_launch_status("cfm_stream_advance");
}

extern "C" int cfm_dwconv_causal_bn_silu(const void* x, int32_t x_dtype, const float* cache, const float* w, const float* dw_bias, const float* bn_scale,
                                         const float* bn_shift, void* y, int32_t y_dtype, int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && w && dw_bias && bn_scale && bn_shift && y, "cfm_dwconv_causal_bn_silu: null pointer");
    CFM_CHECK_ARG(B > 0 && T > 0 && D > 0 && ktaps > 0, "cfm_dwconv_causal_bn_silu: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * T * D;
    CfmProfScope prof("dwconv_causal", s, 2.0 * n * ktaps, (double)n * (cfm_elt_size(x_dtype) + cfm_elt_size(y_dtype)));
    CFM_LAUNCH(cfm_dwconv_causal_kernel, dim3((unsigned)blocks_for(n)), dim3(256), 0, s, x, x_dtype, cache, w, dw_bias, bn_scale, bn_shift, y, y_dtype, B, T, D, ktaps);
    return cfm_launch_status("cfm_dwconv_causal_bn_silu");
}

extern "C" int cfm_conv_cache_update(const void* x, int32_t x_dtype, float* cache, int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream) {
    CFM_CHECK_ARG(x && cache, "cfm_conv_cache_update: null pointer");
    CFM_CHECK_ARG(B > 0 && T > 0 && D > 0 && ktaps > 1, "cfm_conv_cache_update: bad shape");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("conv_cache_update", s, 0.0, (double)B * (ktaps - 1) * D * 8);
    CFM_LAUNCH(cfm_conv_cache_update_kernel, dim3((unsigned)blocks_for((int64_t)B * D)), dim3(256), 0, s, x, x_dtype, cache, B, T, D, ktaps);
    return cfm_launch_status("cfm_conv_cache_update");
}
