// masks.hip -- integer / bool mask builders, bit-exact with the reference's torch code.
//
//  cfm_valid_mask : out[b,t] = (first + stride*t) < len[b]
//       first=0,stride=1  ==  ~make_pad_mask(len, T)                       (utils.py:84-93, encoder.py:62)
//       first=6,stride=4  ==  (~make_pad_mask)[:, :, 2::2][:, :, 2::2]     (convolution.py:76) from lengths
//  cfm_chunk_mask : the T'-iteration python row loop of utils.py:96-111 as one launch
//  cfm_attn_mask  : valid[b,j] & chunk[i,j]                                 (utils.py:150-152)
// Pure byte traffic (HBM-bound, tiny): one coalesced byte per lane.
#include "cfm_common.h"

namespace {

__global__ void cfm_valid_mask_kernel(const void* len, int is64, uint8_t* out, int B, int T, int first, int stride) {
    const int64_t n = (int64_t)B * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / T), t = (int)(i % T);
        const int64_t L = is64 ? ((const int64_t*)len)[b] : (int64_t)((const int32_t*)len)[b];
        out[i] = ((int64_t)first + (int64_t)stride * t) < L ? 1 : 0;
    }
}

__global__ void cfm_chunk_mask_kernel(uint8_t* out, int size, int chunk, int left) {
    const int64_t n = (int64_t)size * size;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / size), j = (int)(idx % size);
        const int blk = i / chunk;
        int start = 0;
        if (left >= 0) {
            start = (blk - left) * chunk;
            if (start < 0) start = 0;
        }
        int64_t end = (int64_t)(blk + 1) * chunk;
        if (end > size) end = size;
        out[idx] = (j >= start && j < end) ? 1 : 0;
    }
}

__global__ void cfm_attn_mask_kernel(const uint8_t* valid, const uint8_t* chunk, uint8_t* out, int B, int T) {
    const int64_t tt = (int64_t)T * T, n = (int64_t)B * tt;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / tt);
        const int64_t ij = idx % tt;
        const int j = (int)(ij % T);
        out[idx] = (valid[(int64_t)b * T + j] != 0 && chunk[ij] != 0) ? 1 : 0;
    }
}

inline unsigned blocks_for(int64_t n) {
    int64_t nb = (n + 255) / 256;
    return (unsigned)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}

}  // namespace

extern "C" int cfm_valid_mask(const void* lengths, int len_is_i64, uint8_t* out, int32_t B, int32_t T, int32_t first,
                              int32_t stride, cfm_stream_t stream) {
    CFM_CHECK_ARG(lengths && out && B > 0 && T > 0 && stride > 0 && first >= 0, "cfm_valid_mask: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("valid_mask", s, 0.0, (double)B * T);
    CFM_LAUNCH(cfm_valid_mask_kernel, dim3(blocks_for((int64_t)B * T)), dim3(256), 0, s, lengths, len_is_i64, out, B, T,
                       first, stride);
    return cfm_launch_status("cfm_valid_mask");
}

extern "C" int cfm_chunk_mask(uint8_t* out, int32_t size, int32_t chunk, int32_t left, cfm_stream_t stream) {
    CFM_CHECK_ARG(out && size > 0 && chunk > 0, "cfm_chunk_mask: bad arguments (size=%d chunk=%d)", size, chunk);
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("chunk_mask", s, 0.0, (double)size * size);
    CFM_LAUNCH(cfm_chunk_mask_kernel, dim3(blocks_for((int64_t)size * size)), dim3(256), 0, s, out, size, chunk, left);
    return cfm_launch_status("cfm_chunk_mask");
}

extern "C" int cfm_attn_mask(const uint8_t* valid, const uint8_t* chunk, uint8_t* out, int32_t B, int32_t T,
                             cfm_stream_t stream) {
    CFM_CHECK_ARG(valid && chunk && out && B > 0 && T > 0, "cfm_attn_mask: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    CfmProfScope prof("attn_mask", s, 0.0, (double)B * T * T + (double)T * T + (double)B * T);
    CFM_LAUNCH(cfm_attn_mask_kernel, dim3(blocks_for((int64_t)B * T * T)), dim3(256), 0, s, valid, chunk, out, B, T);
    return cfm_launch_status("cfm_attn_mask");
}
