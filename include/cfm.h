/* cfm.h -- C ABI of libconformer_gfx950.so: the MI355X (gfx950 / CDNA4) conformer-encoder hot path.
 *
 * This is the drop-in boundary described in SURVEY.md section 8(b).  The reference
 * (Lingeng56/conformer-pytorch-lightning) has no native layer at all: its hot path is eager
 * torch.nn ops issued from Python.  Each entry point below therefore names the reference Python
 * call site it replaces (file:line under the reference's src/), and INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only (no torch types).  Every pointer is a DEVICE pointer on the GPU
 *    that owns `stream`, unless a comment says "host".
 *  - every launch function returns CFM_OK (0) or a negative cfm_status; it never throws, aborts,
 *    allocates device memory or synchronises.  Text of the last error of the calling thread:
 *    cfm_last_error().
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *  - activations are row-major [rows, cols]; "16-bit" means bf16 or fp16 as selected by cfm_dtype.
 *  - all entry points are re-entrant; the only global state is the optional profiling table
 *    (cfm_prof_*), guarded by a mutex.
 */
#ifndef CFM_H_
#define CFM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFM_VERSION 302 /* 0.3.2: cfm_rowchain_desc.cin_* (the conv-in chain as the input stage of the next launch). 0.3.1: row chains at D = 512, cfm_rowchain_desc.psum_out / psum_in (feed-forward split over workgroup pairs), cfm_conv12_relu at C = 512. 0.3.0: row groups in the train entry points (cfm_train_group, cfm_layer_train_io.n_groups), cfm_gemm_tn_group + deferred weight gradients, cfm_encoder_train_forward / _backward (the whole stack from one host call). 0.2.3: cfm_ffn_split, cfm_layer_scratch.psum (the feed-forward split over FF for few rows). 0.2.2: cfm_ctc_nll_train / cfm_ctc_grad take a beta buffer (both recursions in one launch); GEMM tile ids 9-11 (K groups). 0.2.1: fused front-end (cfm_conv12_relu); attention stage of the conv-in chain (cfm_rowchain_desc.att_*, cfm_layer_scratch.vt). 0.2.0: training entry points */

typedef void* cfm_stream_t;

typedef enum { CFM_F32 = 0, CFM_BF16 = 1, CFM_F16 = 2 } cfm_dtype;

typedef enum {
    CFM_OK = 0,
    CFM_ERR_ARG = -1,         /* bad shape / alignment / null pointer / unsupported combination */
    CFM_ERR_LAUNCH = -2,      /* hipGetLastError() after a launch */
    CFM_ERR_UNSUPPORTED = -3  /* valid request this build has no kernel for */
} cfm_status;

typedef enum {
    CFM_ACT_NONE = 0, CFM_ACT_SILU = 1, CFM_ACT_RELU = 2, CFM_ACT_GLU = 3,
    /* backward epilogues (training): v = alpha * (acc + bias) * f'(aux[m,n]) with aux the forward pre-activation (SiLU) or output (ReLU) */
    CFM_ACT_DSILU = 4, CFM_ACT_DRELU = 5
} cfm_act;

int cfm_version(void);
const char* cfm_last_error(void);
/* 1 when the library was built for gfx950 and a gfx950 device is visible (host query, no launch). */
int cfm_device_ok(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:   C = epilogue( A[M,K] . W[N,K]^T )
 *
 * replaces nn.Linear / nn.Conv1d(k=1) / nn.Conv2d(3,stride 2) call sites:
 *   feedforward.py:17-20 (w_1 + SiLU, w_2), attention.py:62-64,78,99 (linear_q/k/v/pos/out),
 *   convolution.py:41-42 (pointwise_conv1 + GLU), :46-48 (pointwise_conv2 + mask),
 *   convolution.py:62-63 (Conv2d(D,D,3,2)+ReLU as implicit GEMM), :74 (out Linear).
 *
 *  A        a_dtype 16-bit or f32 (f32 is rounded to `w_dtype` while staging; or split, below)
 *  W        [N,K] row-major (the nn.Linear weight layout), 16-bit `w_dtype` (bf16|fp16)
 *  W_lo     optional second plane: when non-NULL the product is evaluated as
 *           A_hi.W_hi + A_lo.W_hi + A_hi.W_lo with A split on the fly (A must be f32, w_dtype bf16):
 *           ~16 mantissa bits, the "f32-accurate" mode.
 *  epilogue v = acc + bias[n];  v = act(v)   (GLU: columns are interleaved in blocks of 16 as
 *           [a0..a15 | g0..g15 | a16.. ] and the output has N/2 columns = a * sigmoid(g));
 *           if row_mask && !row_mask[m]: v = 0   (or acc = 0 before the bias, see mask_mode);
 *           if residual: v = residual[m,n] + alpha * v;     store as c_dtype.
 *  conv     when conv_C > 0, A is a channels-last image [B, T1, F1, C] and row m = (b, t2, f2) of the
 *           3x3 stride-2 convolution output [B, T2, F2, N]; K = 9*C ordered (kt, kf, c).
 *
 * constraints: K % 8 == 0, lda % 8 == 0 (elements), ldc % 2 == 0, N % 2 == 0 (residual: both % 4; GLU: N % 32 == 0).
 * N or ldc not a multiple of 4 (a vocabulary of 5002 columns) is written with column-pair stores.
 */
#define CFM_TILE_AUTO_TRAIN (-1) /* cfm_gemm_desc.tile */
typedef struct {
    const void* A;
    const void* W;
    const void* W_lo;
    const float* bias;
    const float* residual;
    const uint8_t* row_mask;
    void* C;
    int64_t lda, ldc, ldr;
    int32_t M, N, K;
    int32_t a_dtype, w_dtype, c_dtype;
    int32_t act;
    float alpha;
    int32_t conv_C, conv_T1, conv_F1, conv_T2, conv_F2; /* conv_C == 0: plain GEMM */
    int32_t tile;                                       /* 0 = auto (every choice keeps the K order: results do not depend on M, a batch shard reproduces the
                                                           batch bit for bit); -1 = CFM_TILE_AUTO_TRAIN: auto, K-group tiles allowed; 1: 128x128, 2: 64x128, 3: 64x64, 4: 128x64, 5: 32x64, 6: 32x128,
                                                           7: 128x128 persistent workgroups (16-bit plain products, K %% 64 == 0),
                                                           8: 256x256 with LDS-DMA staging (csrc/gemm256.hip: 16-bit operands, K %% 64 == 0, bias / SiLU /
                                                              ReLU epilogues; auto when a cost model says its whole rounds beat the 128x128 tiles),
                                                           9 / 11 / 10: 32x64 k2 / 64x64 k2 / 64x64 k4 -- two or four K groups of four wavefronts per tile that meet
                                                              through LDS (16-bit operands, any epilogue; 11 is chosen under tile = -1 for K >= 1024 on <= 512 tiles
                                                              of 64x64: the long-K products of a training micro-batch) */
    int32_t mask_mode;                                  /* 0: row_mask zeroes the OUTPUT row (after act, before residual);
                                                           1: row_mask zeroes the INPUT row (acc = 0, bias/act still apply) */
    /* training: */
    void* C_pre;        /* optional second output [M,N] (row stride ld_pre, pre_dtype): acc + bias BEFORE the activation -- what the
                           backward of SiLU / GLU needs (feedforward.py:18, convolution.py:42).  GLU: all N interleaved columns. */
    int64_t ld_pre;
    int32_t pre_dtype;
    int32_t aux_dtype;  /* CFM_ACT_DSILU / CFM_ACT_DRELU: dtype and row stride of aux [M,N] */
    const void* aux;
    int64_t ld_aux;
    /* dropout on the OUTPUT element (m,n), after the activation and before row_mask / residual (train mode): kept with probability 1-p and
     * scaled by 1/(1-p); the mask is a pure function of (seed, m*N_out + n) (csrc/cfm_common.h cfm_hash32), so the backward passes the same
     * (p, seed) instead of reading a stored mask.  CFM_ACT_DSILU applies it to acc (the gradient arriving at the dropped activation).
     * drop2: a second, independent mask on the same element (the plain MHSA's extra dropout after linear_out, attention.py:177). */
    float drop_p, drop2_p;
    uint32_t drop_seed, drop2_seed;
} cfm_gemm_desc;

int cfm_gemm(const cfm_gemm_desc* d, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient ("TN" product, csrc/gemm_tn.hip):   C[N,K] (+)= alpha * A[M,N]^T . B[M,K]      (f32 output, MFMA)
 * the d loss / d weight of every nn.Linear / Conv1d(k=1) / Conv2d(3,2) on the path: A = gradient of the layer's output rows,
 * B = the rows the layer consumed (what autograd computes for the `weight` of feedforward.py:17-20, attention.py:62-64,99,
 * convolution.py:41,46,62,74, decoder.py:19).  Both operands are read with the contraction index (the row m) as the SLOW index and
 * transposed on the way out of LDS (ds_read_b64_tr_b16), so no transposed copy of an activation is ever written.
 *  A, B      16-bit (mma_dtype) or f32 (rounded while staging; split = 1: both f32, hi/lo bf16 planes, 3 MFMAs: "f32-accurate")
 *  colsum    optional f32 [N]: (+)= alpha * sum_m A[m,n]     (the gradient of the layer's bias)
 *  row_mask  optional uint8 [M]: rows with 0 contribute nothing (padded frames of convolution.py:47-48)
 *  conv      conv_C > 0: B is a channels-last image [Bt, T1, F1, C] and row m = (b, t2, f2) of its 3x3 stride-2 im2col matrix,
 *            K = 9*C ordered (kt, kf, c) -- the weight gradient of the front-end's second convolution
 *  The M rows are split over `splits` workgroups per output tile (0 = auto) that add their partial products with f32 atomics;
 *  accumulate = 0 zero-fills C / colsum first (same stream).  splits = 1 is bitwise reproducible.
 * constraints: N % 8 == 0, K % 8 == 0, lda/ldb % 8 == 0 (16-bit) or % 4 (f32), ldc % 4 == 0.
 */
typedef struct {
    const void* A;
    const void* B;
    float* C;
    float* colsum;
    const uint8_t* row_mask;
    int64_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t a_dtype, b_dtype, mma_dtype;
    int32_t split, accumulate, splits;
    float alpha;
    int32_t conv_C, conv_T1, conv_F1, conv_T2, conv_F2;
    /* optional scatter of the output (needs accumulate = 1: the caller has zero-filled the destination): row n of C starts at C + row_off[n]
     * (elements) instead of C + n*ldc, column sum n goes to colsum[colsum_off[n]] -- lets one fused product (q|k|v, the interleaved
     * pointwise-conv-1 pack) write each reference parameter's gradient where it lives in a flat gradient buffer.  int64 [N], device. */
    const int64_t* row_off;
    const int64_t* colsum_off;
    int32_t tile; /* 0 = chosen by the library; 64 or 128: output tile edge (experiments, scripts/bench_gemm_tn_splits.py) */
    /* optional SECOND destination of column sum n: colsum[colsum_off2[n]] (entries < 0: none).  pos_bias_u's gradient is linear_q.bias'
     * gradient (attention.py:81: (q + u) . k^T), so the fused q|k|v product writes both.  int64 [N], device; needs accumulate = 1. */
    const int64_t* colsum_off2;
} cfm_gemm_tn_desc;

int cfm_gemm_tn(const cfm_gemm_tn_desc* d, cfm_stream_t stream);
/* n products in ONE launch (the weight gradients of a conformer block: feedforward.py:17-20 x 2, attention.py:62-64,99, convolution.py:41,46
 * under autograd): each is too small to fill the chip and none feeds the chain of input gradients, so a block's backward defers them to
 * its end.  Grouped when all are 16-bit products of one type without row mask / f32 split / implicit convolution and n <= 12; otherwise
 * one launch each, in order.  Same results as n calls of cfm_gemm_tn up to the order of the atomic sums. */
int cfm_gemm_tn_group(const cfm_gemm_tn_desc* descs, int32_t n, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused feed-forward block (one launch; the [M,FF] hidden activation never reaches memory):
 *     y  = (add_x ? x : 0) + alpha * ( W2 . act( W1 . LN(x; ln_g, ln_b) + b1 ) + b2 )
 *     y1 = ln1_g ? LN(y; ln1_g, ln1_b) : y        -> out_f32 (if non-NULL)
 *     y2 = ln2_g ? LN(y1; ln2_g, ln2_b) : y1      -> out16   (if non-NULL, 16-bit)
 * replaces norm + feed_forward(_macaron) + the residual add of encoder_layer.py:56-58 / :67-69 (and :70 through ln1,
 * the next sub-block's norm through ln2).  ln_g == NULL: no input LayerNorm (feedforward.py:16-21 on its own).
 *  x     f32 [M,D];  D in {144, 256};  FF % 32 == 0, FF <= 2048
 *  w1f   W1 [FF,D] packed fragment-major (see cfm/packing.py pack_ffn_fragments): 16-bit
 *        w1f[((ffb*KS1 + kk)*64 + lane)*8 + j] = W1[ffb*16 + (lane&15)][kk*32 + 8*(lane>>4) + j]   (0 for k >= D)
 *  w2f   W2 [D,FF] packed fragment-major with the k order of each 32-block permuted to accumulator order:
 *        w2f[((fs*(D/16) + nf)*64 + lane)*8 + j] = W2[nf*16 + (lane&15)][fs*32 + (j<4 ? 0 : 16) + 4*(lane>>4) + (j&3)]
 */
typedef struct {
    const float* x;
    const float *ln_g, *ln_b;
    const void *w1f, *w2f;
    const float *b1, *b2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float* out_f32;
    void* out16;
    int64_t M;
    int32_t D, FF;
    int32_t w_dtype, out16_dtype;
    int32_t act;   /* CFM_ACT_SILU | CFM_ACT_RELU */
    int32_t add_x;
    float alpha, eps;
} cfm_ffn_desc;

int cfm_ffn_fused(const cfm_ffn_desc* d, cfm_stream_t stream);

/* The same launch in TRAIN mode (one feed-forward sub-block of encoder_layer.py:56-58 / :67-69 under module.train()):
 *     y = x + alpha * drop_o( W2 . drop_h( silu( W1 . LN(x) + b1 ) ) + b2 )
 * keeping what the backward needs: xn_out = LN(x) (16-bit [M,D], the operand of dW1), z_out = the pre-activation (16-bit [M,FF], silu'),
 * h_out = the hidden activation after its dropout (16-bit [M,FF], the operand of dW2).  Dropout masks are the counter-based ones of the
 * unfused products (element row*FF + column for the hidden, row*D + column for the output: the backward regenerates them).
 * w1f / w2f: the fragment-major packs of cfm_ffn_fused (cfm_pack_ffn_fragments builds them on the device).  D = 256, FF % 128 == 0, FF <= 2048. */
typedef struct {
    const float* x;
    const float *ln_g, *ln_b;
    const void *w1f, *w2f;
    const float *b1, *b2;
    float* y;
    void *xn_out, *z_out, *h_out;
    int64_t M;
    int32_t D, FF, w_dtype;
    float alpha, eps;
    float p_hidden, p_out;
    uint32_t seed_hidden, seed_out;
} cfm_ffn_train_desc;
int cfm_ffn_train_supported(int32_t D, int32_t FF);
int cfm_ffn_train_forward(const cfm_ffn_train_desc* d, cfm_stream_t stream);
/* w1f / w2f of n_jobs feed-forwards in one launch; job = 4 x int64 on the device: W1 f32 [FF,D], W2 f32 [D,FF], w1f, w2f (16-bit, FF*D elements each). */
int cfm_pack_ffn_fragments(const int64_t* jobs_dev, int32_t n_jobs, int32_t D, int32_t FF, int32_t w_dtype, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row-local chain on 32-row tiles, one launch (csrc/rowchain.hip):
 *     x  = head_a ? head_res + mask_out( head_a . Wh^T + head_b ) : x          (head_mask zeroes the product's row)
 *     xn = LN(x; ln_g, ln_b), rows with ln_mask == 0 zeroed
 *     y  = w1f ? x + alpha * FFN(xn) : x                                         (FFN as in cfm_ffn_fused, SiLU)
 *     y1 = ln1_g ? LN(y) : y -> out_f32        (head without FFN: out_f32 receives x, the new residual stream)
 *     y2 = ln2_g ? LN(y1) : xn -> out16 (optional) and the tail's input
 *     t  = tail_w ? y2 . Wt^T + tail_b, tail_glu: value*sigmoid(gate) on 16-column fragment pairs -> tail_out (16 bit)
 * The three chains of a conformer block (encoder_layer.py:56-70):
 *     macaron : x, LN_ffm, FFN_m, ln2 = LN_mha, tail = fused QKV                       (no head)
 *     conv-in : head = out-proj on the attention context + residual, LN_conv + pad mask, tail = pointwise-conv-1 + GLU
 *     final   : head = pointwise-conv-2 (+ pad mask) + residual, LN_ff, FFN, ln1 = LN_final
 * All matrices are 16-bit FRAGMENT-MAJOR: w[((nfrag*KS + kk)*64 + lane)*8 + j] = W[nfrag*16 + (lane&15)][kk*32 + 8*(lane>>4) + j]
 * for every matrix here, W2 included (the feed-forward keeps its hidden activation as a 32 x FF tile in LDS between the two
 * products, so W2 is read in natural k order).  Instances: D in {144, 256} with the FF / tail sizes of those configs.
 */
typedef struct {
    const float* x;
    const void* head_a;
    const void* head_w;
    const float* head_b;
    const float* head_res;
    const uint8_t* head_mask;
    const float *ln_g, *ln_b;
    const uint8_t* ln_mask;
    const void *w1f, *w2n; /* fragment-major W1 [FF,D] and W2 [D,FF] (w2n: NATURAL k order, not cfm_ffn_fused's permuted w2f) */
    const float *b1, *b2;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float* out_f32;
    void* out16;
    const void* tail_w;
    const float* tail_b;
    void* tail_out;
    int64_t M;
    int32_t D, FF, tail_N, tail_glu;
    int32_t w_dtype;
    float alpha, eps;
    float* out2_f32; /* optional f32 [M,D]: y2 = LN2(y1) as well (ln2_g required) -- e.g. the encoder's after_norm (encoder.py:74) applied to
                        the last block's output in the same launch */
    /* optional depthwise input stage of the "final" chain (head + feed-forward, no tail): head_a is the GLU output and
     *   a = SiLU( (DepthwiseConv15(head_a) + dw_b) * dw_scale + dw_shift )      rows [B, dw_T] flattened, zero padding at
     * utterance edges -- exactly cfm_dwconv_bn_silu (convolution.py:43-45) without its launch and its round trip. */
    const float *dw_w, *dw_b, *dw_scale, *dw_shift;
    int32_t dw_T, dw_K;
    /* optional, macaron chain with the fused-QKV tail at D = 256 (4 heads x 64): the value columns [2D, 3D) of row (b, t) are written
     * transposed per head, tail_vt[((b*H + h)*64 + d) * vt_ld + t] (16 bit), INSTEAD of into tail_out -- the layout the attention stage
     * below reads as MFMA fragments.  vt_T frames per utterance (M % vt_T == 0), vt_ld >= vt_T elements per row. */
    void* tail_vt;
    int32_t vt_T, vt_ld;
    /* optional attention input stage of the conv-in chain (head_a NULL, att_qkv set): the head input is the self-attention context of the
     * tile's 32 frames, computed in the same launch (attention.py:81-96 batch path: no cache, key-validity mask, one positional row per
     * item) -- replaces a cfm_attention launch and the [M, D] context round trip.  Tiles do not cross utterances: B * ceil(T/32)
     * workgroups.  att_qkv [B*T, 3D] rows q | k | -, att_vt the transposed values above (key columns >= T must hold finite numbers:
     * zero-fill the buffer once), att_p one projected positional row per item (stride att_p_sb elements, 0 = shared) or NULL for plain
     * MHSA, att_mask key validity bytes [B, >= T] (stride att_m_sb) or NULL.  D = 256, att_H = 4, att_T <= 256. */
    const void *att_qkv, *att_vt, *att_p;
    const float *att_bias_u, *att_bias_v;
    const uint8_t* att_mask;
    int64_t att_p_sb, att_m_sb;
    int32_t att_T, att_H, att_vt_ld;
    float att_scale;
    /* optional SECOND feed-forward segment on the same rows, kept in registers (final chain of block i + macaron chain of block i+1):
     *     y1 = LN1(y)                       (as above; out_f32 may be NULL then: nothing else reads the block's output)
     *     z  = y1 + s2_alpha * FFN2( LN(y1; s2_ln_g, s2_ln_b) )   -> s2_out_f32
     *     y2 = LN(z; ln2_g, ln2_b) -> out16 / the tail's input;  the tail follows z.
     * Needs w1f (a first feed-forward), ln2_g and the same FF for both. */
    const float *s2_ln_g, *s2_ln_b;
    const void *s2_w1f, *s2_w2n;
    const float *s2_b1, *s2_b2;
    float* s2_out_f32;
    float s2_alpha;
    /* D = 512 (config 4: 3 984 rows are 125 row tiles for 256 CUs): the feed-forward of a chain split over PAIRS of workgroups, one half of FF each.
     *   psum_out  (with w1f, no tail, no ln1 / ln2): the launch runs [head ->] LN -> this workgroup's half of the feed-forward and leaves the partial
     *             sums psum_out[half][M][D] (f32, without b2 / alpha / residual); a head chain also writes its rows x to out_f32, which must NOT alias
     *             head_res then (the pair's other workgroup still reads it).  2 x ceil(M/32) workgroups, halves on different XCDs.
     *   psum_in   (a chain without head and feed-forward): the rows are x + psum_alpha * (psum_in[0] + psum_in[1] + psum_b2) -> out_f32, then LN -> the
     *             tail; without a tail the normalised rows go to out2_f32. */
    float* psum_out;
    const float* psum_in;
    const float* psum_b2;
    float psum_alpha;
    /* The conv-in chain of the SAME block as the input stage of the depthwise stage (with dw_w and the second segment, D = 256): out-projection of the
     * attention context cin_a [M,D] (16 bit) + residual cin_res -> cin_out (this tile's rows; == head_res, != cin_res), LN(cin_ln_*; rows with cin_mask == 0
     * zeroed), pointwise-conv-1 + GLU (cin_tail_w: GLU-interleaved, fragment-major) -- computed per tile for its 32 + 14 halo rows, the GLU rows go
     * straight to the depthwise stage (head_a is not read).  One launch per block less than conv-in chain + this chain. */
    const void *cin_a, *cin_w;
    const float *cin_b, *cin_res;
    float* cin_out;
    const float *cin_ln_g, *cin_ln_b;
    const uint8_t* cin_mask;
    const void* cin_tail_w;
    const float* cin_tail_b;
    int32_t tail_pair; /* D = 512, a chain with a tail and no feed-forward: the tail's columns split over workgroup pairs (rows and LayerNorm computed by
                          both, written by the first); out_f32 must NOT alias head_res then */
} cfm_rowchain_desc;

int cfm_rowchain(const cfm_rowchain_desc* d, cfm_stream_t stream);
/* 1 when cfm_rowchain has instances for all three chains of a block with these sizes (host query) */
int cfm_rowchain_supported(int32_t D, int32_t FF);
/* 1 when the depthwise input stage (dw_w ...) exists at this width; otherwise cfm_dwconv_bn_silu runs in front of the final chain (D = 512) */
int cfm_rowchain_dw_supported(int32_t D);
/* 1 when the pair-split feed-forward (psum_out / psum_in) has instances at this size */
int cfm_rowchain_pair_supported(int32_t D, int32_t FF);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (eps inside sqrt, biased variance), optionally two chained norms in one pass:
 *   y1 = LN(x; g1, b1);  if out1: out1 = y1 (out1_dtype)
 *   if g2:  y2 = LN(y1; g2, b2) else y2 = y1;  if out2: out2 = row_mask[m] ? y2 : 0  (out2_dtype)
 * replaces nn.LayerNorm at encoder_layer.py:57,60,64,68,70 and encoder.py:74, and the
 * masked_fill of convolution.py:36-37 (row_mask).   x is f32 [M,D]; D % 4 == 0, D <= 2048.
 */
int cfm_layernorm(const float* x, const float* g1, const float* b1, void* out1, int out1_dtype,
                  const float* g2, const float* b2, void* out2, int out2_dtype,
                  const uint8_t* row_mask, float eps, int64_t M, int32_t D, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention:  out[b,i,h,:] = softmax_j( scale * ((q_i+u_h).k_j + (q_i+v_h).p_{b,j}) ) . v_j
 * with masked scores = -inf and fully masked rows giving 0 (attention.py:81-96 / :162-172).
 *   q,k,v   element (b,t,h,d) at  base + b*sb + t*st + h*dk + d   (dtype 16-bit or f32)
 *   p       projected positions, element (b,j,h,d) at p + b*p_sb + j*p_st + h*dk + d;
 *           p_st == 0 broadcasts one row over all keys (the reference's batch path);
 *           p == NULL: plain MHSA (no u/v either).
 *   mask    uint8/bool, element (b,i,j) at mask + b*m_sb + i*m_sq + j ; m_sq == 0 broadcasts over
 *           queries; NULL = no mask.
 *   out     [B,Tq,H*dk] row-major, out_dtype.          dk <= 64.
 */
typedef struct {
    const void* q;
    const void* k;
    const void* v;
    const void* p;
    const float* bias_u; /* [H,dk] f32 */
    const float* bias_v; /* [H,dk] f32 */
    const uint8_t* mask;
    void* out;
    int64_t q_sb, q_st, k_sb, k_st, k_sh, v_sb, v_st, v_sh, p_sb, p_st, m_sb, m_sq;
    int32_t B, H, Tq, Tk, dk;
    int32_t q_dtype, kv_dtype, p_dtype, out_dtype, mma_dtype; /* mma_dtype: bf16|fp16 operand type */
    int32_t split;                                            /* 1: hi/lo bf16 split (f32-accurate) */
    float scale;
    float* lse;        /* optional (training): f32 [B,H,Tq], log-sum-exp of each row's scaled masked scores (-inf: fully masked) */
    float drop_p;      /* dropout on the attention probabilities (attention.py:93), element ((b*H+h)*Tq+i)*Tk+j; the softmax normaliser is */
    uint32_t drop_seed;/* that of the undropped row, as torch's softmax -> dropout gives */
} cfm_attn_desc;

int cfm_attention(const cfm_attn_desc* d, cfm_stream_t stream);
/* n attention problems in ONE launch: the micro-batches of a training window (cfm_train_group) differ in B, T and mask, and none fills the
 * chip alone.  Grouped when every problem takes the d_k = 64 fast path without a positional term (16-bit K/V, same mask kind), n <= 8;
 * otherwise one launch each, in order.  Results are those of n calls of cfm_attention. */
int cfm_attention_group(const cfm_attn_desc* descs, int32_t n, cfm_stream_t stream);

/* new_cache[b,h,t,0:dk] = K_t, [dk:2dk] = V_t with rows t < Tc copied from old_cache and rows t >= Tc
 * taken from (k,v) (same addressing as cfm_attn_desc).  f32 output [B,H,Tc+Tn,2dk].
 * replaces torch.split/cat of attention.py:70-76. */
int cfm_kv_cache_pack(const float* old_cache, int32_t Tc, const void* k, const void* v, int32_t kv_dtype,
                      int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float* new_cache,
                      int32_t B, int32_t H, int32_t Tn, int32_t dk, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Depthwise conv (k taps, zero padding at the tensor edges only) + per-channel affine (folded
 * BatchNorm, eval) + SiLU on a channels-last [B,T,D] activation.
 * replaces convolution.py:43-45.   w [D,Ktaps] f32, dw_bias/bn_scale/bn_shift [D] f32.
 *   y = silu( (sum_k w[d,k] x[b,t+k-(Ktaps-1)/2,d] + dw_bias[d]) * bn_scale[d] + bn_shift[d] )
 */
int cfm_dwconv_bn_silu(const void* x, int x_dtype, const float* w, const float* dw_bias,
                       const float* bn_scale, const float* bn_shift, void* y, int y_dtype, int32_t B,
                       int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Front-end first conv: x [B,T,F] f32 -> relu(conv3x3 stride 2) as channels-last [B,T1,F1,C].
 * replaces convolution.py:60-61.  w [9,C] f32 (tap-major), bias [C].  C % 8 == 0.
 * Optional global CMVN folded into the tap loads (cmvn.py:22-33, encoder.py:59-60): every input sample is read as
 * (x[b,t,f] - cmvn_mean[f]) * cmvn_istd[f]  (cmvn_istd NULL: mean only; both NULL: plain) -- the same two f32 operations the
 * reference applies before the convolution, so results are bit-identical to normalising first.
 */
int cfm_conv1_relu(const float* x, const float* w, const float* bias, void* y, int y_dtype, int32_t B,
                   int32_t T, int32_t F, int32_t C, const float* cmvn_mean, const float* cmvn_istd, cfm_stream_t stream);

/* The same convolution on the matrix pipe (csrc/convmod.hip cfm_conv1_mma_kernel): the 9 taps as a K = 32 MFMA contraction.  Inputs
 * (after the optional CMVN, applied in f32) and weights are ROUNDED to the 16-bit type y_dtype (bf16 / f16), products accumulate in
 * f32 on top of the bias; cfm_conv1_relu multiplies in f32.  For the 16-bit precision modes, whose conv1 output is 16-bit anyway.
 * C % 16 == 0, C <= 256.  Same arguments otherwise. */
int cfm_conv1_relu_mma(const float* x, const float* w, const float* bias, void* y, int y_dtype, int32_t B, int32_t T, int32_t F,
                       int32_t C, const float* cmvn_mean, const float* cmvn_istd, cfm_stream_t stream);

/* Both front-end convolutions in one kernel (csrc/frontend.hip): relu(conv3x3 s2 (relu(conv3x3 s2 (x)))) -> y [B,T2,F2,C] channels-last in
 * the 16-bit type y_dtype.  replaces convolution.py:60-63 in eval mode.  The first convolution is recomputed inside the second one's
 * A-operand producer with the operands of cfm_conv1_relu_mma, so its [B,T1,F1,C] output never exists in memory; the result is
 * bit-identical to cfm_conv1_relu_mma followed by cfm_gemm(conv_*, ReLU) on the same packed weights.
 * w1 [9,C] f32 tap-major, b1 [C], w2 [C, 9*C] 16-bit (y_dtype) in K order (kt, kf, ci), b2 [C] f32.  C in {64,128,192,256}.  CMVN as cfm_conv1_relu. */
int cfm_conv12_supported(int32_t C, int32_t y_dtype);
int cfm_conv12_relu(const float* x, const float* w1, const float* b1, const void* w2, const float* b2, void* y, int32_t y_dtype, int32_t B,
                    int32_t T, int32_t F, int32_t C, const float* cmvn_mean, const float* cmvn_istd, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Masks -- integer/bool, bit-exact with the reference.
 *  cfm_valid_mask     out[b,t] = (t*stride + first) < len[b]     (uint8 0/1), t in [0,T)
 *                     first=0,stride=1: ~make_pad_mask (utils.py:84-93, encoder.py:62);
 *                     first=6,stride=4: the subsampled mask of convolution.py:76 built straight from lengths.
 *  cfm_chunk_mask     out[i,j] = start_i <= j < end_i            (utils.py:96-111)
 *  cfm_attn_mask      out[b,i,j] = valid[b,j] & chunk[i,j]       (utils.py:150-152)
 */
int cfm_valid_mask(const void* lengths, int len_is_i64, uint8_t* out, int32_t B, int32_t T, int32_t first,
                   int32_t stride, cfm_stream_t stream);
int cfm_chunk_mask(uint8_t* out, int32_t size, int32_t chunk, int32_t left, cfm_stream_t stream);
int cfm_attn_mask(const uint8_t* valid, const uint8_t* chunk, uint8_t* out, int32_t B, int32_t T,
                  cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-stream state of the batched streaming step (csrc/stream.hip) -- beyond the reference, whose forward_chunk serves one stream at a
 * time (encoder.py:78-123) and rebuilds its cache with cat + slice (:117).  offsets int32 [B] on the device: encoder frames each stream has
 * consumed.  need = chunk * left_chunks cached frames, ring_T >= need + T slots.
 *  cfm_stream_prep     slot_mask u8 [B,ring_T] = slot holds a frame of [offset-min(offset,need), offset+T); pos_rows f32 [B,ring_T,D] = pe[that
 *                      frame] (pe f32 [max_len,D], the sinusoid table of attention.py:12-16); abs_rows f32 [B,D] = pe[offset] (optional: the
 *                      absolute encoding's row, attention.py:119-120)
 *  cfm_kv_ring_write   K / V rows of the T new frames (addressed like cfm_attn_desc k / v, head h at h*dk) -> ring slots (offset+t) mod ring_T
 *  cfm_stream_advance  offset[b] += T for streams with active[b] != 0 (active NULL: all)
 *  cfm_dwconv_causal_bn_silu  y = SiLU(BN_eval(causal depthwise conv over [cache | x])), cache f32 [B,ktaps-1,D] or NULL (zeros)
 *  cfm_conv_cache_update      cache <- last ktaps-1 frames of [cache | x], in place
 */
int cfm_stream_prep(const int32_t* offsets, int32_t B, int32_t T, int32_t need, int32_t ring_T, const float* pe, int32_t max_len, int32_t D,
                    uint8_t* slot_mask, float* pos_rows, float* abs_rows, cfm_stream_t stream);
int cfm_kv_ring_write(const void* k, const void* v, int32_t kv_dtype, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, float* ring,
                      const int32_t* offsets, int32_t B, int32_t H, int32_t T, int32_t dk, int32_t ring_T, cfm_stream_t stream);
int cfm_stream_advance(int32_t* offsets, const uint8_t* active, int32_t B, int32_t T, cfm_stream_t stream);
int cfm_dwconv_causal_bn_silu(const void* x, int32_t x_dtype, const float* cache, const float* w, const float* dw_bias, const float* bn_scale,
                              const float* bn_shift, void* y, int32_t y_dtype, int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream);
int cfm_conv_cache_update(const void* x, int32_t x_dtype, float* cache, int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream);

/* element-wise dtype conversion:  dst = cast(src) */
int cfm_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, cfm_stream_t stream);
/* x[r,:] += add[r / group, :]   (f32, in place; the absolute positional encoding of attention.py:119-120,
 * where one table row is added to every frame of a batch item: group = T') */
int cfm_add_rows(float* x, const float* add, int64_t rows, int32_t D, int32_t group, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Composite: one conformer block / the whole encoder stack, enqueued from C++ so that a forward is
 * ~1 host call instead of ~170 (encoder_layer.py:49-71 and the loop at encoder.py:72-74).
 * Weights are the PACKED device copies the Python side prepares once (see cfm/packing.py).
 */
typedef struct {
    /* layer norms, f32 [D] */
    const float *ln_ffm_g, *ln_ffm_b, *ln_mha_g, *ln_mha_b, *ln_conv_g, *ln_conv_b, *ln_ff_g, *ln_ff_b,
        *ln_final_g, *ln_final_b;
    /* macaron FFN, FFN: W 16-bit [FF,D] / [D,FF] (+ lo planes in split mode), bias f32 */
    const void *ffm_w1, *ffm_w1_lo, *ffm_w2, *ffm_w2_lo;
    const float *ffm_b1, *ffm_b2;
    const void *ff_w1, *ff_w1_lo, *ff_w2, *ff_w2_lo;
    const float *ff_b1, *ff_b2;
    /* optional fragment-major packs for cfm_ffn_fused / cfm_rowchain (NULL: the separate-GEMM path is used) */
    const void *ffm_w1f, *ffm_w2f, *ff_w1f, *ff_w2f;
    const void *ffm_w2n, *ff_w2n; /* W2 fragment-major in natural k order: what cfm_rowchain reads (w2f: cfm_ffn_fused) */
    const void *qkv_wf, *out_wf, *pw1_wf, *pw2_wf;
    /* attention: fused qkv [3D,D], pos [D,D] (NULL for plain MHSA), out [D,D] */
    const void *qkv_w, *qkv_w_lo, *pos_w, *pos_w_lo, *out_w, *out_w_lo;
    const float *qkv_b, *out_b, *bias_u, *bias_v;
    /* conv module: pw1 [2D,D] GLU-interleaved, dw [D,K], folded BN, pw2 [D,D] */
    const void *pw1_w, *pw1_w_lo, *pw2_w, *pw2_w_lo;
    const float *pw1_b, *pw2_b, *dw_w, *dw_b, *bn_scale, *bn_shift;
} cfm_layer_weights;

typedef struct {
    void *xn, *hid, *qkv, *pos, *ctx, *glu, *dw; /* activation-dtype scratch: [M,D],[M,FF],[M,3D],[R,D],[M,D],[M,D],[M,D] */
    void* vt;      /* optional [B, D, vt_ld] activation-dtype, ZERO-FILLED ONCE by the caller: transposed values for the attention stage of the
                      conv-in chain (cfm_rowchain_desc.att_*); NULL: attention runs as its own launch */
    int32_t vt_ld; /* elements per row of vt: >= 256, multiple of 4 */
    float* psum;   /* optional f32 [psum_splits, M, D]: partial slabs of the split feed-forward (cfm_ffn_split).  Given with psum_splits >= FF/256,
                      blocks of at most CFM_FFSPLIT_MAX_ROWS rows at D = 256 run their two feed-forwards split over FF/256 workgroups per 32-row
                      tile instead of inside the row chains (few rows: a streaming step); null = never.  At D = 512 (cfm_rowchain_pair_supported) with
                      psum_splits >= 3 and at most CFM_PAIR_MAX_ROWS rows, both feed-forwards run split over workgroup PAIRS (two slabs + the parked rows of
                      the final chain): 125 row tiles alone leave half of the CUs idle */
    int32_t psum_splits;
} cfm_layer_scratch;
#define CFM_PAIR_MAX_ROWS 4096    /* 128 row tiles x 2 halves = one workgroup per CU */
#define CFM_FFSPLIT_MAX_ROWS 1536 /* measured crossover against the row chains: 996 rows -21 %, 1992 rows +4 % (scripts/bench_small_batch.py) */

typedef struct {
    int32_t B, T, D, H, FF, ktaps;
    int32_t act_dtype; /* CFM_BF16 | CFM_F16 (16-bit modes) | CFM_F32 (split mode) */
    int32_t w_dtype;   /* CFM_BF16 | CFM_F16 */
    const uint8_t* attn_mask;
    int64_t am_sb, am_sq;     /* see cfm_attn_desc.mask */
    const uint8_t* pad_valid; /* [B*T] or NULL */
    const float* pos_embed;   /* f32 [R,D] rows, R = B*P */
    int32_t pos_rows;         /* R (0: plain MHSA) */
    const void* pos_proj;     /* optional: linear_pos(pos_embed) already projected (act dtype), row stride pos_proj_ld;
                                 lets the driver project the positions of ALL blocks with one GEMM */
    int64_t pos_proj_ld;
    const float* attn_cache;  /* f32 [B,H,Tc,2dk] or NULL */
    int32_t cache_T;
    float* new_cache;         /* f32 [B,H,Tc+T,2dk] or NULL (not materialised) */
    const float *after_g, *after_b; /* optional, chain path only: ALSO write LN(x_out; after_g, after_b) to after_out (f32 [B*T,D]) -- the
                                       encoder's after_norm fused into the last block's final chain */
    float* after_out;
    /* Batched streaming with PER-STREAM state (csrc/stream.hip; beyond the reference): when kv_ring is set, this layer's keys /
     * values live in a ring buffer f32 [B,H,ring_T,2dk] (frame f of a stream in slot f mod ring_T); the step's T new frames of stream b are
     * frames stream_offset[b] .. +T-1 and are written to their slots, attention runs over all ring_T slots with attn_mask = the (B,1,ring_T)
     * slot mask and pos_rows = B*ring_T positional rows (both from cfm_stream_prep).  attn_cache / new_cache are unused then. */
    float* kv_ring;
    const int32_t* stream_offset;
    int32_t ring_T;
    /* OPT-IN causal depthwise convolution (not in the reference, which has no causal mode and ignores cnn_cache: convolution.py:34-39):
     * taps reach back ktaps-1 frames instead of (ktaps-1)/2 each way; conv_cache f32 [B,ktaps-1,D] (optional) is the left context of this
     * chunk -- the last GLU outputs of the previous one -- and is updated in place.  0 = the reference's symmetric convolution. */
    int32_t causal_conv;
    float* conv_cache;
    int32_t pos_shared;       /* 1: the pos_rows == Tk positional rows are the SAME for every batch item (batched streaming step: all
                                 streams at one offset).  Beyond the reference, whose forward_chunk only works at batch 1
                                 (attention.py:78-88); per item it equals that batch-1 call. */
    /* Chaining consecutive blocks (chain path only): the final chain of this block and the macaron chain of the NEXT block are both
     * row-local, so one launch can run them back to back on rows that stay in registers (cfm_rowchain_desc.s2_*): one launch, one f32
     * read and one f32 write of the residual stream less per block.
     *   next_w != NULL : after norm_final, run the next block's macaron feed-forward + norm_mha + fused QKV projection as well; the
     *                    next block's post-macaron residual goes to next_x_out (f32 [B*T,D]) and its q|k|v rows to s->qkv.  x_out then
     *                    does NOT receive this block's output (it is left holding the residual stream before the feed-forward).
     *   macaron_done   : this block's macaron chain already ran in the previous block's call: x_out holds its residual, s->qkv its
     *                    projections; x_in is not read. */
    const cfm_layer_weights* next_w;
    float* next_x_out;
    int32_t macaron_done;
} cfm_layer_io;

/* x_in f32 [B*T,D] (not modified) -> x_out f32 [B*T,D] = norm_final(block(x_in)).
 * If next_g != NULL additionally writes LN(x_out; next_g,next_b) to s->xn for the following block. */
int cfm_encoder_layer_forward(const cfm_layer_weights* w, const cfm_layer_scratch* s, const cfm_layer_io* io,
                              const float* x_in, float* x_out, int xn_ready, const float* next_g,
                              const float* next_b, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CTC negative log-likelihood per utterance on top of the vocabulary projection (csrc/ctc.hip):
 * replaces  probs = logits.transpose(0,1).log_softmax(2); nn.CTCLoss(reduction='sum')(probs, labels, enc_lens, label_lens)
 * of CTCDecoder.forward (reference src/decoder.py:20-21; blank = 0, no zero_infinity).  logits f32 [B,T,ld>=V] (from cfm_gemm with
 * ctc_lo.weight / bias), labels int32 [B,Umax] (padding ignored beyond label_lens[b]), work f32 scratch [B,T,2*Umax+2], nll f32 [B];
 * the caller sums nll and divides by Umax as decoder.py:22 does.  Umax <= 255.  An impossible alignment gives +inf, like nn.CTCLoss.
 */
int cfm_ctc_nll(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens,
                const int32_t* labels, int32_t Umax, const int32_t* label_lens, float* work, float* nll, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Transducer joint, activation operand (csrc/joint.hip):  replaces
 *     out = enc_out.unsqueeze(2) + pred_out.unsqueeze(1);  out = tanh(out)        (reference src/joint.py:31-37)
 * enc f32 [B*T, ld_e >= J] = enc_ffn(encoder_out), pred f32 [B*U, ld_p >= J] = pred_ffn(predictor_out) (both from cfm_gemm);
 * out [B*T*U, J] row-major in out_dtype (f32 / bf16 / f16), row (b*T + t)*U + u = tanh(enc[b*T+t] + pred[b*U+u]).
 * The vocabulary projection ffn_out is then cfm_gemm over these rows (N = vocab_size may be any multiple of 2).  J % 8 == 0.
 */
int cfm_joint_act(const float* enc, int64_t ld_e, const float* pred, int64_t ld_p, void* out, int32_t out_dtype, int32_t B,
                  int32_t T, int32_t U, int32_t J, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training (BASELINE config 3: encoder + CTC loss + backward).  Input gradients of the dense layers are cfm_gemm on transposed
 * weight packs (with the CFM_ACT_DSILU / CFM_ACT_DRELU epilogues), weight / bias gradients are cfm_gemm_tn; the rest is below.
 * Every reduction over rows is two-stage through a caller-provided f32 workspace (size from the matching *_ws function, in floats)
 * and bitwise reproducible.
 *
 * cfm_layernorm_bwd     x f32 [M,D] (the norm's INPUT; statistics are recomputed), dy [M,D] (dy_dtype), gamma [D];
 *                       dx = (dres ? dres : 0) + dLN(dy)   (dx may alias dres: the residual stream's gradient is updated in place);
 *                       rows with row_mask == 0 have dy = 0 (the masked norm_conv of convolution.py:36-37);  dgamma, dbeta f32 [D].
 *                       backward of nn.LayerNorm at encoder_layer.py:57,60,64,68,70 and encoder.py:74.   D % 4 == 0, D <= 1024.
 */
int64_t cfm_layernorm_bwd_ws(int64_t M, int32_t D);
/* The same backward with two launches folded in (what a conformer block's backward does right before and after it):
 *   accumulate != 0 : dgamma / dbeta are ADDED to with f32 atomics in the same launch (no workspace pass; the caller zero-fills them or
 *                     carries a running sum; summation order varies from run to run) -- 0: two-stage through ws, overwritten, reproducible;
 *   dx2 != NULL     : also writes dropout(alpha2 * dx) in dx2_dtype -- the next residual branch's gradient as a GEMM operand, exactly
 *                     cfm_dropout_rows(dx, .., alpha2, p1, seed1, p2, seed2) without its launch. */
typedef struct {
    const float* x;
    const void* dy;
    const float* gamma;
    const uint8_t* row_mask;
    const float* dres;
    float *dx, *dgamma, *dbeta, *ws;
    void* dx2;
    int64_t M;
    int32_t D, dy_dtype, dx2_dtype, accumulate;
    float eps, alpha2, p1, p2;
    uint32_t seed1, seed2;
    const uint8_t* dx2_row_mask; /* optional: rows of dx2 with mask == 0 are written as zeros (cfm_dropout_rows' row_mask) */
    /* optional CHAINED second norm (needs accumulate = 1): dx = dLN( dres + dLN(dy; x, gamma) ; chain_x, chain_gamma ) -- the backward of two
     * LayerNorms applied one after the other to the same rows (block l's norm_final feeding block l+1's norm_ff_macaron, encoder_layer.py:70,57)
     * in one launch; the intermediate gradient is never stored, dx / dx2 come from the second stage, chain_dgamma / chain_dbeta get its sums. */
    const float *chain_x, *chain_gamma;
    float *chain_dgamma, *chain_dbeta;
} cfm_ln_bwd_desc;
int cfm_layernorm_bwd_fused(const cfm_ln_bwd_desc* d, cfm_stream_t stream);
int cfm_layernorm_bwd(const float* x, const void* dy, int32_t dy_dtype, const float* gamma, const uint8_t* row_mask, const float* dres,
                      float* dx, float* dgamma, float* dbeta, float* ws, float eps, int64_t M, int32_t D, cfm_stream_t stream);

/* GLU backward (convolution.py:42) on the column-interleaved layout the pointwise-conv-1 GEMM writes through cfm_gemm_desc.C_pre:
 * u [M,2D] (blocks of 16 value columns followed by their 16 gate columns), dg [M,D] -> du [M,2D] same layout.  D % 16 == 0. */
int cfm_glu_bwd(const void* u, int32_t u_dtype, const void* dg, int32_t dg_dtype, void* du, int32_t du_dtype, int64_t M, int32_t D,
                cfm_stream_t stream);

/* One micro-batch of an accumulation window (train.sh:36 accum_grad; src/executor.py:151): B utterances of T frames whose rows start at row0
 * of the window's row matrices.  Groups are contiguous: row0 of group i+1 = row0 of group i + B*T. */
#define CFM_TRAIN_MAX_GROUPS 8
typedef struct {
    int32_t B, T;
    int64_t row0;
    const uint8_t* attn_mask; /* this micro-batch's attention mask, strides as cfm_layer_train_io.am_sb / am_sq */
    int64_t am_sb, am_sq;
} cfm_train_group;

/* Depthwise conv (15 taps) -> BatchNorm1d in TRAINING mode -> SiLU  (convolution.py:43-45 under module.train()):
 *   c = dw(g) + bias            -> c_out f32 [B,T,D]  (kept for the backward)
 *   batch mean / biased variance over ALL B*T rows of each channel, padded frames included
 *   stats f32 [4][D] = mean, rstd, scale = gamma*rstd, shift = beta - mean*scale;   s_out = SiLU(c*scale + shift)  (s_dtype)
 *   running_mean / running_var (optional) are updated in place with `momentum` and the UNBIASED variance, as torch does.
 * cfm_dwconv_bn_train_bwd: ds = d loss / d s_out -> dg_out = d loss / d g, and the gradients of the taps dw_w [D,15], the conv bias
 * dw_b [D], the BatchNorm gain / bias dgamma, dbeta [D].  dy_ws: f32 [B*T,D] scratch.  ws: cfm_dwconv_bn_ws(B,T,D) floats (both). */
int64_t cfm_dwconv_bn_ws(int32_t B, int32_t T, int32_t D);
int cfm_dwconv_bn_train(const void* g, int32_t g_dtype, const float* w, const float* dw_bias, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* c_out, float* stats, void* s_out,
                        int32_t s_dtype, float* ws, int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream);
int cfm_dwconv_bn_train_bwd(const void* ds, int32_t ds_dtype, const float* c, const float* stats, const void* g, int32_t g_dtype, const float* w,
                            void* dg_out, int32_t dg_dtype, float* dw_w, float* dw_b, float* dgamma, float* dbeta, float* dy_ws, float* ws,
                            int32_t B, int32_t T, int32_t D, int32_t ktaps, cfm_stream_t stream);
/* accumulate != 0: the four parameter gradients (dw_w, dw_b, dgamma, dbeta) are added to what the buffers hold (one writer per element:
 * reproducible) instead of overwritten -- a gradient buffer shared by the micro-batches of an optimizer step. */
int cfm_dwconv_bn_train_bwd_acc(const void* ds, int32_t ds_dtype, const float* c, const float* stats, const void* g, int32_t g_dtype, const float* w,
                                void* dg_out, int32_t dg_dtype, float* dw_w, float* dw_b, float* dgamma, float* dbeta, float* dy_ws, float* ws,
                                int32_t B, int32_t T, int32_t D, int32_t ktaps, int32_t accumulate, cfm_stream_t stream);

/* Front-end backward (convolution.py:60-63).  cfm_col2im_relu_bwd: dcol [B*T2*F2, 9*C] = dh2 . W2 (cfm_gemm on the transposed conv2 pack,
 * K order (kt,kf,c)) -> dh1 [B,T1,F1,C] = ReLU'(h1) * (transposed im2col of dcol).  cfm_conv1_wgrad: dh1 and the fbank input x [B,T,F]
 * (global CMVN folded as in cfm_conv1_relu) -> dw [9,C] tap-major, db [C]. */
/* The same two entry points over the micro-batches of a training window (cfm_train_group; csrc/train_layer.cpp): g / c / s / ds / dg / dy_ws are the
 * window's [M, D] row matrices, stats is [n_groups][4*D]; every stage is ONE launch for all micro-batches, each with its own batch statistics,
 * the running statistics updated micro-batch after micro-batch.  ws: the sum of cfm_dwconv_bn_ws over the groups. */
int cfm_dwconv_bn_train_groups(const void* g, int32_t g_dtype, const float* w, const float* dw_bias, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* c_out, float* stats, void* s_out,
                               int32_t s_dtype, float* ws, const cfm_train_group* groups, int32_t n_groups, int32_t D, int32_t ktaps, cfm_stream_t stream);
int cfm_dwconv_bn_train_bwd_groups(const void* ds, int32_t ds_dtype, const float* c, const float* stats, const void* g, int32_t g_dtype, const float* w,
                                   void* dg_out, int32_t dg_dtype, float* dw_w, float* dw_b, float* dgamma, float* dbeta, float* dy_ws, float* ws,
                                   const cfm_train_group* groups, int32_t n_groups, int32_t D, int32_t ktaps, int32_t accumulate, const void* glu_u,
                                   void* glu_du, cfm_stream_t stream);
/* glu_u / glu_du (both or neither; dtype = g_dtype = dg_dtype): the GLU backward of convolution.py:42 in the depthwise launch -- u [M,2D] the
 * pre-GLU columns (cfm_gemm_desc.C_pre layout), du [M,2D] their gradient; dg_out is then not written (may be NULL).  Same values as
 * cfm_glu_bwd on the rounded dg. */
int cfm_col2im_relu_bwd(const void* dcol, int32_t dcol_dtype, const void* h1, int32_t h1_dtype, void* dh1, int32_t dh1_dtype, int32_t B, int32_t T1,
                        int32_t F1, int32_t C, cfm_stream_t stream);
int64_t cfm_conv1_wgrad_ws(int32_t B, int32_t T, int32_t C);
int cfm_conv1_wgrad(const void* dh1, int32_t dh1_dtype, const float* x, const float* cmvn_mean, const float* cmvn_istd, float* dw, float* db, float* ws,
                    int32_t B, int32_t T, int32_t F, int32_t C, cfm_stream_t stream);

/* Attention backward (csrc/attention_bwd.hip; attention.py:81-96 under autograd).  cfm_attention with `lse` set also writes the row
 * log-sum-exp of the scaled, masked scores, lse f32 [B,H,Tq] (-inf for a fully masked row).  Given dout [B,Tq,H*dk] the backward
 * recomputes the probabilities tile by tile (nothing of size Tq x Tk reaches memory) and writes grad_q/k/v with the strides of q, k, v.
 * Batch path only (p broadcast over keys or absent: the positional term is then constant along a softmax row and has no gradient;
 * d pos_bias_u = the column sums of dq).  Fully masked rows contribute nothing (the reference's masked_fill(0) makes them constant). */
typedef struct {
    const void *q, *k, *v;          /* element (b,t,h,d) at base + b*sb + t*st + h*dk + d; q already includes pos_bias_u (train mode adds it
                                       through the projection's bias) */
    const uint8_t* mask;
    const void* out;                /* forward output [B,Tq,H*dk] */
    const void* dout;               /* [B,Tq,H*dk] */
    const float* lse;               /* [B,H,Tq] */
    void *grad_q, *grad_k, *grad_v; /* same strides as q, k, v */
    float* delta;                   /* f32 scratch [B,H,Tq] */
    int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, m_sb, m_sq;
    int32_t B, H, Tq, Tk, dk;
    int32_t io_dtype;               /* dtype of q,k,v,out,dq,dk,dv (16-bit or f32) */
    int32_t dout_dtype;
    int32_t mma_dtype, split;
    float scale;
    float drop_p;                   /* the forward's probability dropout (same p, same seed) */
    uint32_t drop_seed;
} cfm_attn_bwd_desc;
int cfm_attention_bwd(const cfm_attn_bwd_desc* d, cfm_stream_t stream);
/* n backward problems in three launches (delta, dq, dkv over all of them) when every one takes the d_k = 64 / 16-bit kernels; else one by one. */
int cfm_attention_bwd_group(const cfm_attn_bwd_desc* descs, int32_t n, cfm_stream_t stream);
/* d_k = 64 with 16-bit, 16-byte-aligned q / k / v / dO rows and no mask or a key-validity mask (m_sq == 0) takes fast kernels (tiles staged as
 * they lie in memory, transposed operands read with ds_read_b64_tr_b16, next tile prefetched) -- bit-identical to the general ones, which
 * this switch forces (tests). */
void cfm_attention_bwd_force_general(int32_t on);
/* Chained blocks (cfm_layer_io.next_w): run the conv-in chain as the input stage of the block's last launch (two launches per block; the default, also
 * switched off by CFM_CIN_MERGE=0 in the environment at first use) or as a launch of its own (three).  The results are bit-identical; bench.py times the
 * narrower launch beside the merged one with it.  Returns the previous setting. */
int32_t cfm_set_cin_merge(int32_t on);

/* Feed-forward for FEW rows, the hidden dimension split across workgroups (csrc/ffnsplit.hip; feedforward.py:17-20 inside encoder_layer.py:55-58,
 * 67-70 when B*T' is a few hundred rows: a streaming step, BASELINE config 5).  One launch = the row-reduce input stage + (by mode) nothing, a
 * projection, or a feed-forward that leaves PARTIAL sums:
 *   rows      x = psum ? x + psum_alpha * (sum_{g < psum_splits} psum[g] + psum_b2) : x;   if ln1: x = LN1(x);   rows_out = x (optional)
 *   mode 0    rows2_out = LN2(x) (optional)
 *   mode 1    out16[:, n] = act(LN(x) . w1[n,:] + b1[n]),  n < N1                       grid = (ceil(M/32), N1/256)
 *   mode 2    psum_out[g] = act(LN(x) . w1[256 g .. 256 g + 255,:]^T + b1) . w2[:, 256 g ..]^T,  g < N1/256 (N1 = FF), to be reduced -- with b2, alpha,
 *             the residual and the next norm(s) -- by the rows stage of the next cfm_ffn_split call (fixed order g = 0,1,..: reproducible)
 * w1: fragment-major [N1/16][D/32][64][8] (packing.pack_frag_major / pack_ffn_fragments' w1f), w2: fragment-major [D/16][FF/32][64][8] (w2n);
 * 16-bit `w_dtype`; f32 rows [M,D]; psum / psum_out f32 [splits][M][D]; rows_out may alias x only when one slice runs (mode 0, or N1 == 256).  D = 256. */
typedef struct {
    const float* x;
    const float* psum;
    const float* psum_b2;
    int32_t psum_splits;
    float psum_alpha;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float *rows_out, *rows2_out;
    const float *ln_g, *ln_b;
    const void* w1;
    const float* b1;
    int32_t N1, act;
    const void* w2;
    float* psum_out;
    void* out16;
    int64_t ldo;
    int32_t M, D, mode, w_dtype;
    float eps;
    /* mode 1 as the fused q|k|v projection of a batched streaming step (N1 = 3 D): when kv_ring is given, the key and value columns are ALSO written,
     * as the 16-bit values widened to f32, into the per-stream ring f32 [B, ring_H, ring_T, 2 dk] at slot (ring_offsets[b] + t) mod ring_T, row
     * m = b * ring_Tq + t -- what cfm_kv_ring_write would copy from out16, without its launch */
    float* kv_ring;
    const int32_t* ring_offsets;
    int32_t ring_T, ring_H, ring_Tq;
} cfm_ffn_split_desc;
int cfm_ffn_split(const cfm_ffn_split_desc* d, cfm_stream_t stream);
int cfm_ffn_split_supported(int32_t D, int32_t FF);

/* CTC backward (csrc/ctc.hip): cfm_ctc_nll_train is cfm_ctc_nll that also keeps log alpha (alpha f32 [B,T,2*Umax+2]), the per-frame
 * log-sum-exp (lse f32 [B,T]) and nll_shifted f32 [B] (-log P of the per-frame-shifted recursion: the posteriors' normaliser); cfm_ctc_grad runs the beta recursion and writes d loss / d logits [B,T,ld] = gscale[b] * (softmax - occupancy)
 * for t < enc_lens[b] and 0 elsewhere (pad columns V..ld-1 too) -- what autograd gives for nn.CTCLoss(reduction='sum') on
 * log_softmax(logits) (decoder.py:20-21).  The scale is gscale * (gscale_dev ? *gscale_dev : 1): a host factor (1 / padded label length,
 * decoder.py:22) times an optional DEVICE scalar (the upstream gradient).  alpha_beta is cfm_ctc_nll_train's alpha, overwritten.
 * beta (f32 [B,T,2*Umax+2], may be null in both calls): given to cfm_ctc_nll_train, the backward recursion runs BESIDE the forward one in the same
 * launch (2 B workgroups) and leaves log(beta / y) there; given to cfm_ctc_grad, that array is used instead of running the recursion (alpha is then
 * left untouched).  Both forms produce the same bits.
 * An utterance with no valid alignment (nll = inf) gets a zero gradient (torch's is undefined without zero_infinity).  V <= 8192. */
int cfm_ctc_nll_train(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens, const int32_t* labels, int32_t Umax,
                      const int32_t* label_lens, float* work, float* alpha, float* lse, float* nll, float* nll_shifted, float* beta, cfm_stream_t stream);
/* cfm_ctc_nll_train for the micro-batches of a training window: the per-frame row passes per micro-batch, then BOTH recursions of ALL micro-batches in
 * one launch (each is a serial chain on one CU; 2 * B workgroups per micro-batch).  Fields as cfm_ctc_nll_train's arguments; beta is required. */
typedef struct {
    const float* logits;
    int64_t ld;
    int32_t B, T, Umax;
    const int32_t *enc_lens, *labels, *label_lens;
    float *work, *alpha, *lse, *nll, *nll_shifted, *beta;
} cfm_ctc_group;
int cfm_ctc_nll_train_groups(const cfm_ctc_group* groups, int32_t n, int32_t V, cfm_stream_t stream);
int cfm_ctc_grad(const float* logits, int64_t ld, int32_t B, int32_t T, int32_t V, const int32_t* enc_lens, const int32_t* labels, int32_t Umax,
                 const int32_t* label_lens, const float* work, float* alpha_beta, const float* beta, const float* lse, const float* nll_shifted,
                 float gscale, const float* gscale_dev, float* dlogits, cfm_stream_t stream);

/* Dropout as an elementwise pass:  y[m,n] = keep(seed, m*N + n) ? alpha * x[m,n] / (1-p) : 0, rows with row_mask == 0 zeroed.  The backward of
 * a residual branch x + alpha * dropout(f(..)) needs d f = alpha * mask/(1-p) * dx as a GEMM operand (encoder_layer.py:58,61,64,69); the
 * forward of the same dropout is fused into the producing GEMM's epilogue (cfm_gemm_desc.drop_p), this pass only regenerates its mask.
 * cfm_dropout_mask writes the 0/1 keep mask itself (tests). */
int cfm_dropout_rows(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, const uint8_t* row_mask, float alpha, float p, uint32_t seed,
                     float p2, uint32_t seed2, int64_t M, int32_t N, cfm_stream_t stream);
int cfm_dropout_mask(uint8_t* out, int64_t n, float p, uint32_t seed, cfm_stream_t stream);

/* Weight packs of the training path, any number of matrices in one launch (csrc/train.hip cfm_pack_kernel).  jobs_dev: DEVICE array of
 * n_jobs x 8 int64 -- { rows, N, K, dst, dst_lo, dst_t, dst_t_lo, first_tile }: `rows` a device array of N pointers to f32 source rows of K
 * contiguous values (concatenated or permuted sources are just pointer tables: the fused q|k|v matrix, the GLU-interleaved pointwise
 * conv), dst [N,K] and dst_t [K,N] the 16-bit matrix and its transpose (either may be 0), *_lo the bf16 lo planes when split != 0,
 * first_tile the prefix sum of ceil(N/64) * ceil(K/64) over the preceding jobs; total_tiles the sum over all jobs.  N % 8 == 0, K % 8 == 0.
 * Replaces the per-step torch casts / transposes of cfm/packing.py (what the reference's optimizer step makes necessary: the kernels
 * read 16-bit copies of the f32 master weights). */
int cfm_pack_matrices(const int64_t* jobs_dev, int32_t n_jobs, int64_t total_tiles, int32_t w_dtype, int32_t split, cfm_stream_t stream);
/* out[i] = *a[i] + (b[i] ? *b[i] : 0), i < n: the small f32 vectors of the training packs of a whole block stack in one launch (the fused
 * q|k|v bias with pos_bias_u added to its first third, attention.py:62-64,81; the GLU-interleaved pointwise-conv-1 bias, convolution.py:41-42)
 * as gathers through two device tables of element pointers. */
int cfm_pack_vectors(const float* const* a, const float* const* b, float* out, int64_t n, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * One conformer block in TRAIN mode as two host calls (csrc/train_layer.cpp): encoder_layer.py:49-71 under module.train() and its
 * backward, the same launches in the same order as the op-by-op composition of cfm/autograd.py.  All buffers are the caller's:
 *   weights  the train packs (forward [N,K] and transposed [K,N] 16-bit matrices, + lo planes in the f32-accurate mode; cfm/packing.py)
 *   saved    activations the backward needs, act dtype unless typed: xn1..4 [M,D], z1,z2,h1,h2 [M,FF], qkv [M,3D], ctx [M,D], u [M,2D], glu/s [M,D],
 *            f32 x1..x4 / c [M,D], lse [B,H,T], stats [4,D]
 *   scratch  backward work buffers (dxn f32 [M,D], dz [M,FF], dyb/ds/dglu/dctx [M,D], du [M,2D], dqkv [M,3D], delta f32 [B,H,T],
 *            ln_ws cfm_layernorm_bwd_ws floats, dwbn_ws cfm_dwconv_bn_ws floats, dy_ws f32 [M,D]; dz2 / dyb2..4 with a side stream);
 *            the forward uses dwbn_ws only
 *   grads    where each parameter's gradient goes: plain pointers, and for the two fused products (q|k|v, the interleaved pointwise-conv-1
 *            pack) per-row element offsets relative to `slab` (cfm_gemm_tn_desc.row_off).  Weight / bias gradients are ACCUMULATED: the
 *            caller zero-fills; LayerNorm / BatchNorm / depthwise gradients are overwritten.  pos_bias_u receives a copy of q_bias' gradient.
 * Dropout: probabilities per site and one seed (site s uses seed + 0x9E3779B1*s); 0 disables.  deterministic = 1: weight-gradient products
 * run unsplit (no atomics).  Returns like every entry point; nothing is synchronised. */
typedef struct {
    const float *ln_ffm_g, *ln_ffm_b, *ln_mha_g, *ln_mha_b, *ln_conv_g, *ln_conv_b, *ln_ff_g, *ln_ff_b, *ln_final_g, *ln_final_b;
    const void *ffm_w1, *ffm_w1_lo, *ffm_w2, *ffm_w2_lo, *ffm_w1t, *ffm_w1t_lo, *ffm_w2t, *ffm_w2t_lo;
    const float *ffm_b1, *ffm_b2;
    const void *ff_w1, *ff_w1_lo, *ff_w2, *ff_w2_lo, *ff_w1t, *ff_w1t_lo, *ff_w2t, *ff_w2t_lo;
    const float *ff_b1, *ff_b2;
    const void *qkv_w, *qkv_w_lo, *qkv_t, *qkv_t_lo, *out_w, *out_w_lo, *out_t, *out_t_lo;
    const float *qkv_b, *out_b;
    const void *pw1_w, *pw1_w_lo, *pw1_t, *pw1_t_lo, *pw2_w, *pw2_w_lo, *pw2_t, *pw2_t_lo;
    const float *pw1_b, *pw2_b, *dw_w, *dw_b, *bn_gamma, *bn_beta;
    float *bn_running_mean, *bn_running_var;
    float bn_momentum, bn_eps;
    /* optional: fragment-major packs of the two feed-forwards (cfm_pack_ffn_fragments).  When set (and D = 256, 16-bit mode) each feed-forward's
     * forward is ONE launch (cfm_ffn_train_forward) instead of LayerNorm + two products. */
    const void *ffm_w1f, *ffm_w2f, *ff_w1f, *ff_w2f;
} cfm_layer_train_weights;


typedef struct {
    int32_t B, T, D, H, FF, ktaps, act_dtype, w_dtype;
    const uint8_t* attn_mask;
    int64_t am_sb, am_sq;
    const uint8_t* pad_valid;
    float p_hidden_m, p_hidden, p_branch, p_attn, p_attn_out;
    uint32_t seed;
    int32_t deterministic;
    /* backward only: the gradient slab is a running sum (the optimizer step's flat gradient buffer itself) -- every parameter gradient is ADDED
     * to it and nothing in it is overwritten.  Needs deterministic == 0 (the LayerNorm sums are added with atomics). */
    int32_t grads_accumulate;
    /* backward only, optional: a second HIP stream of the same device.  The weight-gradient products do not feed the chain of input
     * gradients, so they are issued there (each after an event on its operands) and overlap with the chain on the main stream -- at
     * training batch sizes no single kernel fills the chip; the main stream waits for the side stream before the call's work is
     * complete from its point of view (an event wait, not a host synchronisation).  NULL: everything on `stream`. */
    cfm_stream_t side_stream;
    /* ROW GROUPS (optional; n_groups = 0: one micro-batch B x T with attn_mask / am_sb / am_sq above).  The micro-batches of an accumulation
     * window concatenated along the row axis: every row matrix has M = sum B_g*T_g rows, pad_valid has M entries, lse / delta are the groups'
     * [B_g,H,T_g] arrays one after the other, stats is [n_groups][4*D].  Row-local work (dense products, LayerNorm, their backward, the
     * weight gradients) runs once over all rows; attention, the depthwise convolution and BatchNorm run per group, BatchNorm's running
     * statistics updated group after group -- the reference's sequence of forward passes over those micro-batches.  host array. */
    int32_t n_groups;
    const cfm_train_group* groups;
    /* backward: keep the operands of the block's eight weight-gradient products until its last launch and issue them as ONE
     * cfm_gemm_tn_group (needs the dz2 / dyb2..4 scratch buffers and, for pos_bias_u, grads.qkv_bias_off2; 16-bit modes -- ignored in the
     * f32-accurate mode and with a side stream). */
    int32_t defer_wgrad;
} cfm_layer_train_io;

typedef struct {
    void *xn1, *z1, *h1, *xn2, *qkv, *ctx, *xn3, *u, *glu, *s, *xn4, *z2, *h2;
    float *x1, *x2, *x3, *x4, *c, *lse, *stats;
} cfm_layer_train_saved;

typedef struct {
    float* dxn;
    void *dz, *dyb, *ds, *dglu, *du, *dctx, *dqkv;
    float *delta, *ln_ws, *dwbn_ws, *dy_ws;
    void *dz2, *dyb2, *dyb3, *dyb4; /* with a side stream: the operands of overlapped weight-gradient products must outlive the sub-block that
                                       made them: second feed-forward's dz, one branch-gradient buffer per sub-block (act dtype, [M,FF] / [M,D]) */
} cfm_layer_train_scratch;

typedef struct {
    float* slab;
    float *ln_ffm_g, *ln_ffm_b, *ln_mha_g, *ln_mha_b, *ln_conv_g, *ln_conv_b, *ln_ff_g, *ln_ff_b, *ln_final_g, *ln_final_b;
    float *ffm_w1, *ffm_b1, *ffm_w2, *ffm_b2, *ff_w1, *ff_b1, *ff_w2, *ff_b2;
    float *out_w, *out_b, *pw2_w, *pw2_b, *dw_w, *dw_b, *bn_g, *bn_b;
    float* pos_bias_u;
    const float* q_bias;
    const int64_t *qkv_row_off, *qkv_bias_off, *pw1_row_off, *pw1_bias_off;
    const int64_t* qkv_bias_off2; /* optional: int64 [3D], entry n < D = offset of pos_bias_u[n] relative to slab, others -1 (cfm_gemm_tn_desc.colsum_off2) */
} cfm_layer_train_grads;

int cfm_encoder_layer_train_forward(const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                                    const cfm_layer_train_scratch* t, const float* x_in, float* y_out, cfm_stream_t stream);
int cfm_encoder_layer_train_backward(const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                                     const cfm_layer_train_scratch* t, const cfm_layer_train_grads* g, const float* x_in, const float* dy, float* dx,
                                     cfm_stream_t stream);

/* The whole block stack in train mode (src/encoder.py:72-73 `for block in self.encoders` under module.train()) from ONE host call each way.
 *   w, sv, g  arrays of n_layers structs; io, t shared by all blocks (layer l's dropout seed is derived from io->seed and l)
 *   xs        n_layers + 1 f32 [M,D] row matrices: xs[0] the stack's input, xs[l+1] block l's output (kept: block l+1's backward reads xs[l+1])
 * backward: dy = gradient of xs[n_layers]; dbuf0 / dbuf1 two f32 [M,D] work buffers, *dx_out is set to the one holding the gradient of xs[0].
 * `done(l, user)` (optional) is called on the host right after block l's backward launches have been enqueued, last block first -- the
 * data-parallel trainer starts that block's gradient bucket all-reduce from it (DDP's reducer hook, executor.py:137-154).
 * t is an array of n_scratch (1 or 2) scratch sets, block l uses set l & 1.  With io->defer_wgrad AND io->side_stream (two sets needed) each
 * block's grouped weight-gradient launch runs on the side stream beside the next block's chain of input gradients; `done(l)` is then
 * reported one block late, after the main stream has been made to wait for block l's launch, and the call ends with the streams joined. */
typedef void (*cfm_layer_done_fn)(int32_t layer, void* user);
int cfm_encoder_train_forward(int32_t n_layers, const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                              const cfm_layer_train_scratch* t, float* const* xs, cfm_stream_t stream);
int cfm_encoder_train_backward(int32_t n_layers, const cfm_layer_train_weights* w, const cfm_layer_train_io* io, const cfm_layer_train_saved* sv,
                               const cfm_layer_train_scratch* t, int32_t n_scratch, const cfm_layer_train_grads* g, float* const* xs, const float* dy,
                               float* dbuf0, float* dbuf1, cfm_layer_done_fn done, void* user, float** dx_out, cfm_stream_t stream);

/* Optimizer step over flat f32 buffers (module.py:140-143 Adam; executor.py:150 gradient_clip_val):  g' = g * (*grad_scale) + wd * p;
 * m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)   (torch.optim.Adam).
 * grad_scale: optional DEVICE scalar (the clip coefficient), so nothing synchronises between backward and step.
 * cfm_sumsq: out[0] = sum x^2 (two-stage, n_partials <= 4096 workgroups). */
int cfm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, const float* grad_scale, cfm_stream_t stream);
int cfm_sumsq(const float* x, int64_t n, float* partials, int32_t n_partials, float* out, cfm_stream_t stream);
/* cfm_adam_step with the step's scalar glue inside (executor.py:150 gradient_clip_val + Lightning DDP's gradient averaging): the gradient
 * scale is min(1, clip / (sqrt(*sumsq) * inv_world + 1e-6)) * inv_world (clip <= 0: inv_world), computed on the device from cfm_sumsq's result;
 * *norm_out (optional) receives the averaged gradient's norm; zero_grad != 0 zeroes g as it is read (the next step's accumulation buffer). */
int cfm_adam_clip_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t step, const float* sumsq, float clip, float inv_world, int32_t zero_grad, float* norm_out, cfm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Profiling table (aux subsystem: tracing).  When enabled, every kernel launch made through this
 * library carries a start and a stop HIP event attached to the dispatch itself (its own begin/end timestamps, the
 * duration a profiler reports); cfm_prof_collect() synchronises those events and accumulates per-kernel-name
 * totals.  Used by bench.py for the live roofline figure.
 */
void cfm_prof_enable(int on);
void cfm_prof_reset(void);
int cfm_prof_collect(void);                 /* returns number of distinct kernel names */
int cfm_prof_entry(int i, char* name, int name_cap, int64_t* calls, double* total_ms, double* flops,
                   double* bytes);

/* ------------------------------------------------------------------------------------------------
 * One step of the batched RNN-T greedy search (reference src/model.py:215-269: predictor step predictor.py:76-86, joint joint.py:20-38,
 * argmax and the loop's bookkeeping) for B <= 64 streams, six launches, everything float32 (csrc/greedy.hip).  All state lives in caller
 * buffers and is updated in place; calling it repeatedly (or replaying a captured graph of calls) runs the search, n_done counts the
 * streams that have reached their last frame.  Weights are f32 row-major [out, in]:
 *   embed [vocab, E];  lstm_w[l] [4H, in_l + H] = [W_ih | W_hh] with ROWS ORDERED [unit][gate i,f,g,o] (not torch's [gate][unit]),
 *   lstm_b[l] [4H] = b_ih + b_hh in the same order;  proj_w [P, H], pf_w [J, P] (pred_ffn), out_w [Vp, J] (ffn_out, rows >= V zero,
 *   out_b there -inf);  enc_proj [B, T, J] = enc_ffn(encoder output).
 *   state: token / t / count / frame_count / lens int64 [B], hyps int64 [B, hyp_ld] (at most hyp_cap + 1 entries used), h / c and the
 *   candidates h_new / c_new f32 [L, B, H], done uint8 [B];  scratch: pred [B, P], act [B, J], pmax / pidx [Vp/16, B]. */
typedef struct {
    const float* embed;
    const float* lstm_w[4];
    const float* lstm_b[4];
    const float *proj_w, *proj_b, *pf_w, *pf_b, *out_w, *out_b;
    const float* enc_proj;
    int64_t *token, *t, *count, *frame_count, *hyps;
    const int64_t* lens;
    float *h, *c, *h_new, *c_new, *pred, *act, *pmax;
    int32_t* pidx;
    uint8_t* done;
    int32_t* n_done;
    int64_t hyp_cap, hyp_ld;
    int32_t B, T, L, E, H, P, J, Vp, blank, n_steps;
} cfm_greedy_desc;
int cfm_greedy_step(const cfm_greedy_desc* d, cfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CFM_H_ */
