// gemm_tn.hip -- weight gradients on gfx950 MFMA:   C[N,K] (+)= alpha * A[M,N]^T . B[M,K]      (f32 out)
//
// The backward of every dense layer of the conformer block needs  dW = dY^T . X  with dY [M,N] and X [M,K] both stored
// row-major, i.e. with the CONTRACTION index m as the slow index of both operands (feedforward.py:17-20, attention.py:62-64,99,
// convolution.py:41,46,62,74 and decoder.py:19 under autograd).  The forward GEMM (gemm.hip) wants k-contiguous operands; rather
// than writing transposed copies of every activation, this kernel stages [m][col] tiles as they lie in memory (16-byte global
// loads along n / k) and lets the LDS hardware transpose on the way out: ds_read_b64_tr_b16 hands each lane 4 consecutive m of
// one column, two of them make the 8-element MFMA fragment.  Both operands use the same (permuted) m order inside a 32-row
// step -- {4g..4g+3, 16+4g..16+4g+3} for lane group g -- which is all the contraction needs.
//
//   tile   128 x 128 or 64 x 64 outputs per workgroup (template TILE), 256 threads = 2 x 2 wavefronts of TILE/2 x TILE/2
//   M      split over `splits` workgroups per tile (training batches are M = 2-8 k rows against N x K <= 2048 x 256 outputs: a
//          single pass over M would leave most CUs idle); partial products meet in C through f32 atomics (global_atomic_add_f32).
//          Every split costs N x K atomics, and those -- not the MFMAs -- set the time of a small product (measured, config 3:
//          66 us average with 128 x 128 tiles and 8-64 splits, 4 M atomics per feed-forward weight): small outputs therefore take the
//          64 x 64 tile (4 x the tiles, so 4 x fewer splits for the same number of workgroups)
//   bias   the column sums of A (d loss / d bias) ride along in the k-tile-0 workgroups: the A tile is in LDS anyway
//   conv   B rows may be the implicit im2col rows of a channels-last image (3x3 stride 2): the front-end's conv2 weight gradient
//   SPLIT  f32 operands split into bf16 hi/lo planes while staging, 3 MFMAs per fragment pair (the f32-accurate mode)
#include <math.h>
#include <stdlib.h>
#include <string>

#include "cfm_common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct TnArgs {
    const void* A;
    const void* B;
    float* C;
    float* colsum;
    const uint8_t* mask;
    int64_t lda, ldb, ldc;
    int M, N, K;
    float alpha;
    int convC, T1, F1, T2, F2;
    int chunks_per_split, atomic;   // atomic: 1 = f32 atomic adds, 0 = plain store, 2 = plain read-add-store (accumulate with ONE workgroup per tile)
    const int64_t* row_off;
    const int64_t* colsum_off;
    const int64_t* colsum_off2;     // optional second destination of column sum n (< 0: none): pos_bias_u shares linear_q.bias' gradient
};

// LDS row stride in 16-bit elements: TILE + 16 -> 288 B = 72 words = 8 (mod 64) for TILE 128, 160 B = 40 words for TILE 64: the 4 rows of a
// transposed read hit disjoint banks either way
template <typename HT, bool SPLIT, int STR>
__device__ __forceinline__ void stage_store(u16* tile, int plane_elems, int row, int col, const u32x4& raw, const f32x4& f0, const f32x4& f1, bool is_f32) {
    u16* p = tile + row * STR + col;
    if constexpr (SPLIT) {
        u32x4 hi, lo;
        split8(f0, f1, hi, lo);
        *(u32x4*)p = hi;
        *(u32x4*)(p + plane_elems) = lo;
    } else {
        *(u32x4*)p = is_f32 ? pack8<HT>(f0, f1) : raw;
    }
}

// one MFMA operand fragment (16 columns starting at c0, the 32 rows of m-step ms) from a [m][col] tile
template <int STR>
__device__ __forceinline__ u32x4 tr_frag(const u16* tile, int ms, int c0, int g, int l15) {
    const u16* pa = tile + (ms * 32 + 4 * g + (l15 >> 2)) * STR + c0 + (l15 & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 16 * STR));
    const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
    return (u32x4){lo2.x, lo2.y, hi2.x, hi2.y};
}

template <typename HT, bool SPLIT, bool A32, bool B32, bool CONV, int TILE>
__global__ __launch_bounds__(256) void cfm_gemm_tn_kernel(const TnArgs g) {
    constexpr int TN_BN = TILE, TN_BK = TILE, TN_STR = TILE + 16;
    constexpr int FR = TILE / 32;                    // 16-wide fragments per wavefront and operand (4 or 2)
    constexpr int PPR = TILE / 8;                    // 16-byte pieces per tile row
    constexpr int MCH = (SPLIT ? 32 : 64) * (TILE == 64 ? 2 : 1);   // rows of M per staged chunk (per barrier): 64-wide tiles do only 8 MFMAs
                                                                     // per wavefront and 32-row step, so they take twice the rows per barrier
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int PLANE = MCH * TN_STR;              // 16-bit elements per operand plane
    constexpr int PASSES = MCH * PPR / 256;          // 16-byte pieces per thread and operand
    constexpr int RPP = 256 / PPR;                   // tile rows staged per pass
    __shared__ __attribute__((aligned(16))) u16 smem[2 * 2 * NPL * PLANE];   // [buffer][A | B][plane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;         // wavefront's 64 x 64 quadrant: n rows wr, k columns wc
    const int tiles_k = (g.K + TN_BK - 1) / TN_BK;
    const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x % tiles_k;
    const int n0 = tile_n * TN_BN, k0 = tile_k * TN_BK;
    const int chunk_begin = blockIdx.y * g.chunks_per_split;
    const int total_chunks = (g.M + MCH - 1) / MCH;
    int chunk_end = chunk_begin + g.chunks_per_split;
    chunk_end = chunk_end < total_chunks ? chunk_end : total_chunks;
    if (chunk_begin >= chunk_end) return;            // uniform, before any barrier

    // ---- staging: thread -> (row = pass*16 + tid/16, 8 columns at (tid & 15) * 8) of both operand tiles -------------------------
    const int prow = tid / PPR, pcol = (tid % PPR) * 8;
    const bool a_col_ok = n0 + pcol < g.N;           // N % 8 == 0: a piece is entirely inside or outside
    const bool b_col_ok = k0 + pcol < g.K;
    int64_t b_koff = k0 + pcol;                      // element offset of this thread's B columns inside a row
    if constexpr (CONV) {
        const int kk = k0 + pcol;
        const int tap = kk / g.convC, ci = kk - tap * g.convC;
        const int k3 = tap / 3, f3 = tap - 3 * k3;
        b_koff = (int64_t)(k3 * g.F1 + f3) * g.convC + ci;
    }
    u32x4 ra[A32 ? 1 : PASSES], rb[B32 ? 1 : PASSES];
    f32x4 fa[A32 ? PASSES : 1][2], fb[B32 ? PASSES : 1][2];
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};

    auto gload = [&](int chunk) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = chunk * MCH + p * RPP + prow;
            const bool row_ok = m < g.M;
            const bool a_ok = row_ok && a_col_ok && (!g.mask || g.mask[m] != 0);
            const bool b_ok = row_ok && b_col_ok;
            const int mc = row_ok ? m : 0;
            if constexpr (A32) {
                const float* pa = (const float*)g.A + ((int64_t)mc * g.lda + n0 + pcol);
                fa[p][0] = a_ok ? *(const f32x4*)pa : zf;
                fa[p][1] = a_ok ? *(const f32x4*)(pa + 4) : zf;
            } else {
                ra[p] = a_ok ? *(const u32x4*)((const u16*)g.A + ((int64_t)mc * g.lda + n0 + pcol)) : z4;
            }
            int64_t boff;
            if constexpr (CONV) {
                const int per_b = g.T2 * g.F2;
                const int b = mc / per_b, rem = mc - b * per_b;
                const int t2 = rem / g.F2, f2 = rem - t2 * g.F2;
                boff = ((int64_t)(b * g.T1 + 2 * t2) * g.F1 + 2 * f2) * g.convC + b_koff;
            } else {
                boff = (int64_t)mc * g.ldb + b_koff;
            }
            if constexpr (B32) {
                const float* pb = (const float*)g.B + boff;
                fb[p][0] = b_ok ? *(const f32x4*)pb : zf;
                fb[p][1] = b_ok ? *(const f32x4*)(pb + 4) : zf;
            } else {
                rb[p] = b_ok ? *(const u32x4*)((const u16*)g.B + boff) : z4;
            }
        }
    };
    auto lstore = [&](int buf) {
        u16* At = smem + buf * (2 * NPL * PLANE);
        u16* Bt = At + NPL * PLANE;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = p * RPP + prow;
            if constexpr (A32) stage_store<HT, SPLIT, TN_STR>(At, PLANE, row, pcol, z4, fa[p][0], fa[p][1], true);
            else stage_store<HT, SPLIT, TN_STR>(At, PLANE, row, pcol, ra[p], zf, zf, false);
            if constexpr (B32) stage_store<HT, SPLIT, TN_STR>(Bt, PLANE, row, pcol, z4, fb[p][0], fb[p][1], true);
            else stage_store<HT, SPLIT, TN_STR>(Bt, PLANE, row, pcol, rb[p], zf, zf, false);
        }
    };

    f32x4 acc[FR][FR];                               // [n fragment][k fragment]: lane holds C[n = .. + l15][k = .. + 4*g4 + r]
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = zf;
    // bias gradient (k-tile 0 workgroups, their wavefronts of k-half 0): the column sums of A are one more MFMA per A fragment against a
    // fragment of ones -- every output row then holds sum_m A[m][n].  (The first version summed the A tile column by column out of LDS,
    // MCH dependent 2-byte reads per chunk: 2 us per 128-row chunk in exactly the workgroups that finish last.)
    f32x4 acs[FR];
#pragma unroll
    for (int i = 0; i < FR; ++i) acs[i] = zf;
    const bool do_colsum = g.colsum != nullptr && tile_k == 0 && wc == 0;      // wave-uniform
    const unsigned one2 = pack2<HT>(1.0f, 1.0f);
    const u32x4 ones = {one2, one2, one2, one2};

    gload(chunk_begin);
    int buf = 0;
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
        lstore(buf);
        __syncthreads();                             // chunk ch visible; everyone has left chunk ch-1's compute on the OTHER buffer... see below
        if (ch + 1 < chunk_end) gload(ch + 1);
        const u16* At = smem + buf * (2 * NPL * PLANE);
        const u16* Bt = At + NPL * PLANE;
#pragma unroll
        for (int ms = 0; ms < MCH / 32; ++ms) {
            u32x4 af[FR], bf[FR], afl[SPLIT ? FR : 1], bfl[SPLIT ? FR : 1];
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                af[i] = tr_frag<TN_STR>(At, ms, wr * (TILE / 2) + i * 16, g4, l15);
                if constexpr (SPLIT) afl[i] = tr_frag<TN_STR>(At + PLANE, ms, wr * (TILE / 2) + i * 16, g4, l15);
            }
#pragma unroll
            for (int j = 0; j < FR; ++j) {
                bf[j] = tr_frag<TN_STR>(Bt, ms, wc * (TILE / 2) + j * 16, g4, l15);
                if constexpr (SPLIT) bfl[j] = tr_frag<TN_STR>(Bt + PLANE, ms, wc * (TILE / 2) + j * 16, g4, l15);
            }
#pragma unroll
            for (int i = 0; i < FR; ++i)
#pragma unroll
                for (int j = 0; j < FR; ++j) {
                    if constexpr (SPLIT) {
                        acc[i][j] = HT::mfma(bf[j], afl[i], acc[i][j]);
                        acc[i][j] = HT::mfma(bfl[j], af[i], acc[i][j]);
                    }
                    acc[i][j] = HT::mfma(bf[j], af[i], acc[i][j]);
                }
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    if constexpr (SPLIT) acs[i] = HT::mfma(ones, afl[i], acs[i]);
                    acs[i] = HT::mfma(ones, af[i], acs[i]);
                }
            }
        }
        // two buffers, one barrier per chunk: chunk ch+1 is stored into the other buffer, whose last readers (chunk ch-1) all passed
        // the barrier above before anyone gets here
        buf ^= 1;
    }

    // ---- epilogue: C[n][k..k+3] -----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int n = n0 + wr * (TILE / 2) + i * 16 + l15;
        if (n >= g.N) continue;
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const int k = k0 + wc * (TILE / 2) + j * 16 + 4 * g4;
            if (k >= g.K) continue;                  // K % 4 == 0: the 4 columns are valid together
            float* c = g.C + (g.row_off ? g.row_off[n] : (int64_t)n * g.ldc) + k;
            const f32x4 v = acc[i][j] * g.alpha;
            if (g.atomic == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) unsafeAtomicAdd(c + r, v[r]);
            } else if (g.atomic == 2) {
                *(f32x4*)c = *(const f32x4*)c + v;
            } else {
                *(f32x4*)c = v;
            }
        }
    }
    if (do_colsum && g4 == 0) {
#pragma unroll
        for (int i = 0; i < FR; ++i) {
            const int n = n0 + wr * (TILE / 2) + i * 16 + l15;
            if (n < g.N) {
                unsafeAtomicAdd(g.colsum + (g.colsum_off ? g.colsum_off[n] : (int64_t)n), acs[i].x * g.alpha);
                if (g.colsum_off2 && g.colsum_off2[n] >= 0) unsafeAtomicAdd(g.colsum + g.colsum_off2[n], acs[i].x * g.alpha);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The same product for 16-bit operands (every weight gradient of a bf16 / fp16 training step), staged by LDS-DMA with several chunks in
// flight.  The register-staged kernel above keeps ONE chunk ahead, and the compiler drains vmcnt(0) at its loop back-edge, so every chunk
// of a workgroup pays a full memory latency (measured: 2 us per 128-row chunk, 94-104 TFLOP/s at M = 2 380 / 4 000 rows) while a chunk's
// compute is 16 MFMAs per wavefront.  Here a chunk is 64 rows x 64 columns of each operand = 4 global_load_lds requests per thread with no
// register destination; FOUR LDS buffers, chunks c+1 .. c+3 in flight while chunk c is multiplied (48 KB per workgroup, two workgroups
// per CU), a counted s_waitcnt vmcnt(8) and a raw s_barrier per chunk.  The DMA's LDS image is lane-linear (rows of 128 B, no padding), so
// the 16-byte chunk index is XOR-swizzled by (row >> 1) & 7 on the source address and again on the transposed read: rows r and r + 2 of
// a ds_read_b64_tr_b16 group would otherwise share banks.  Rows past M read a 16-byte zero.  64 x 64 tiles only, no row mask.
__device__ __attribute__((aligned(16))) unsigned cfm_tn_zero16[4] = {0u, 0u, 0u, 0u};

template <typename HT, bool CONV, int TILE, int NG>
__device__ __forceinline__ void tn_dma_body(const TnArgs& g, const int block_x, const int block_y) {
    // Measured (scripts/bench_gemm_tn_splits.py, M = 2 380): ONE 64 x 64 workgroup walks its rows at ~0.6 us per 64-row chunk whatever is in
    // flight (8 buffers instead of 4: 27 us instead of 24 for 2 380 rows) -- with one wavefront per SIMD the chunk's own chain of address
    // arithmetic, barrier, 16 transposed reads and 8 dependent MFMAs is exposed; every split added costs ~1.5-5 us of atomics; the
    // 128 x 128 tile is slower at these sizes (36 vs 24 us for a feed-forward weight).  Hence the split rule in cfm_gemm_tn.
    // NG = 2 (the 64 x 64 tile): TWO groups of four wavefronts share the tile -- a chunk is 128 rows, staged by all eight, group 0 multiplies
    // its first 64 rows and group 1 the other 64 into accumulators of their own, which meet through LDS after the last chunk: the same
    // instruction chain per iteration covers twice the rows (two wavefronts per SIMD), without a second split's atomics.
    constexpr int MCH = 64 * NG, NBUF = 4, FR = TILE / 32;
    constexpr int RB = TILE * 2;                       // bytes of a tile row (128 or 256)
    constexpr int PPR = TILE / 8;                      // 16-byte pieces per row
    constexpr int RPQ = 64 / PPR;                      // rows per 1 KB request (8 or 4)
    constexpr int NQ = MCH / RPQ / (4 * NG);           // requests per wavefront, operand and chunk (2 or 4)
    constexpr int OPB = MCH * RB;                      // bytes of one operand chunk (8 or 16 KB)
    constexpr int INFL = (NBUF - 2) * 2 * NQ;          // requests of the NBUF - 2 chunks that may stay in flight behind the one being waited for
    constexpr int WAIT = 0x0F70 | (INFL & 0xF) | ((INFL >> 4) << 14);   // s_waitcnt vmcnt(INFL)
    // The requests are issued from an asm statement (the recipe of /opt/skills/guides/cdna_hip_programming.md: M0 written in the statement that
    // reads it), so they are absent from hipcc's wait-count bookkeeping: issued through __builtin_amdgcn_global_load_lds, hipcc puts an
    // s_waitcnt vmcnt(0) in front of the fragment reads of every chunk (one LDS array or four) and nothing stays in flight.  The waits
    // are the two explicit ones below.
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * 2 * OPB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, l15 = lane & 15;
    const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int tiles_k = (g.K + TILE - 1) / TILE;
    const int tile_n = block_x / tiles_k, tile_k = block_x % tiles_k;
    const int n0 = tile_n * TILE, k0 = tile_k * TILE;
    const int chunk_begin = block_y * g.chunks_per_split;
    const int total_chunks = (g.M + MCH - 1) / MCH;
    int chunk_end = chunk_begin + g.chunks_per_split;
    chunk_end = chunk_end < total_chunks ? chunk_end : total_chunks;
    const int nch = chunk_end - chunk_begin;
    if (nch <= 0) return;                              // uniform, before any barrier

    // ---- staging: request q of wavefront w fills rows 8 (w + 4 q) .. + 7 of an operand chunk; lane L lands at position L & 7 of row
    //      8 (w + 4 q) + (L >> 3) and fetches the 16-byte piece (L & 7) ^ ((row >> 1) & 7) of that row
    // 128-byte rows: rows r and r + 2 share banks -> (r >> 1) & 7; 256-byte rows: every row starts at bank 0 -> r & 7 (the 4 consecutive rows of a
    // transposed read then sit in 4 different 16-byte columns)
    auto swz = [](int row) { return TILE == 64 ? (row >> 1) & 7 : row & 7; };
    int srow[NQ], scol[NQ];
    bool a_ok[NQ], b_ok[NQ];
    int64_t b_koff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        srow[q] = RPQ * (wave + 4 * NG * q) + lane / PPR;
        scol[q] = ((lane % PPR) ^ swz(srow[q])) * 8;                 // (128-wide rows: the XOR stays inside the row's first / second 8 pieces)
        a_ok[q] = n0 + scol[q] < g.N;
        b_ok[q] = k0 + scol[q] < g.K;
        b_koff[q] = k0 + scol[q];
        if constexpr (CONV) {
            const int kk = k0 + scol[q];
            const int tap = kk / g.convC, ci = kk - tap * g.convC;
            const int k3 = tap / 3, f3 = tap - 3 * k3;
            b_koff[q] = (int64_t)(k3 * g.F1 + f3) * g.convC + ci;
        }
    }
    const char* const zsrc = (const char*)cfm_tn_zero16;
    auto glds16 = [&](const char* gsrc, unsigned char* lds_dst) __attribute__((always_inline)) {
        unsigned keep;
        const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_dst;      // wave-uniform LDS byte address
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    auto stage = [&](int buf, int chunk) __attribute__((always_inline)) {
        unsigned char* const At = smem + buf * (2 * OPB);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = chunk * MCH + srow[q];
            const bool row_ok = m < g.M;
            const int mc = row_ok ? m : 0;
            const char* pa = (const char*)((const u16*)g.A + ((int64_t)mc * g.lda + n0 + scol[q]));
            int64_t boff;
            if constexpr (CONV) {
                const int per_b = g.T2 * g.F2;
                const int b = mc / per_b, rem = mc - b * per_b;
                const int t2 = rem / g.F2, f2 = rem - t2 * g.F2;
                boff = ((int64_t)(b * g.T1 + 2 * t2) * g.F1 + 2 * f2) * g.convC + b_koff[q];
            } else {
                boff = (int64_t)mc * g.ldb + b_koff[q];
            }
            const char* pb = (const char*)((const u16*)g.B + boff);
            pa = (row_ok && a_ok[q]) ? pa : zsrc;
            pb = (row_ok && b_ok[q]) ? pb : zsrc;
            glds16(pa, At + (wave + 4 * NG * q) * 1024);
            glds16(pb, At + OPB + (wave + 4 * NG * q) * 1024);
        }
    };
    // transposed fragment: 16 columns from c0, the 32 rows of m-step ms; 8-byte piece (l15 & 3) of rows 4 g4 + (l15 >> 2) and + 16
    auto frag = [&](const unsigned char* tile, int ms, int c0) __attribute__((always_inline)) {
        const int row = ms * 32 + 4 * g4 + (l15 >> 2);
        const int col = c0 + (l15 & 3) * 4;
        const int sw = swz(row);                         // rows row and row + 16 swizzle alike
        const unsigned char* pa = tile + row * RB + (((col >> 3) ^ sw) << 4) + (col & 4) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pa + 16 * RB));
        const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
        return (u32x4){lo2.x, lo2.y, hi2.x, hi2.y};
    };

    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[FR][FR], acs[FR];
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        acs[i] = zf;
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = zf;
    }
    const bool do_colsum = g.colsum != nullptr && tile_k == 0 && wc == 0;
    const unsigned one2 = pack2<HT>(1.0f, 1.0f);
    const u32x4 ones = {one2, one2, one2, one2};

    // prologue: chunks 0 .. 2 in flight (requests past the last chunk read zeros: every chunk slot costs the same 4 requests per thread,
    // which keeps the counted waits below exact)
    auto chunk_id = [&](int c) { return c < nch ? chunk_begin + c : total_chunks; };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) stage(i, chunk_id(i));
    for (int c = 0; c < nch; ++c) {
        __builtin_amdgcn_s_waitcnt(WAIT);                // chunk c has landed, chunks c+1 .. c+NBUF-2 may still be in flight
        __builtin_amdgcn_s_barrier();                    // ... for every wavefront; and everyone is done reading the buffer chunk c+3 goes to
        stage((c + NBUF - 1) % NBUF, chunk_id(c + NBUF - 1));
        const unsigned char* const At = smem + (c % NBUF) * (2 * OPB);
        const unsigned char* const Bt = At + OPB;
#pragma unroll
        for (int msl = 0; msl < 2; ++msl) {
            const int ms = grp * 2 + msl;                // this group's 64 rows of the chunk
            u32x4 af[FR], bf[FR];
#pragma unroll
            for (int i = 0; i < FR; ++i) af[i] = frag(At, ms, wr * (TILE / 2) + i * 16);
#pragma unroll
            for (int j = 0; j < FR; ++j) bf[j] = frag(Bt, ms, wc * (TILE / 2) + j * 16);
#pragma unroll
            for (int i = 0; i < FR; ++i)
#pragma unroll
                for (int j = 0; j < FR; ++j) acc[i][j] = HT::mfma(bf[j], af[i], acc[i][j]);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < FR; ++i) acs[i] = HT::mfma(ones, af[i], acs[i]);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // the zero-source requests of the tail: no DMA may still be writing LDS when the workgroup ends
    if constexpr (NG == 2) {                             // group 1's sums join group 0's (fixed order), group 0 writes the tile
        __syncthreads();
        f32x4* const xch = (f32x4*)smem + ((wave & 3) * (FR * FR + FR)) * 64 + lane;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < FR; ++i) {
#pragma unroll
                for (int j = 0; j < FR; ++j) xch[(i * FR + j) * 64] = acc[i][j];
                xch[(FR * FR + i) * 64] = acs[i];
            }
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < FR; ++i) {
#pragma unroll
            for (int j = 0; j < FR; ++j) acc[i][j] += xch[(i * FR + j) * 64];
            acs[i] += xch[(FR * FR + i) * 64];
        }
    }

#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int n = n0 + wr * (TILE / 2) + i * 16 + l15;
        if (n >= g.N) continue;
#pragma unroll
        for (int j = 0; j < FR; ++j) {
            const int k = k0 + wc * (TILE / 2) + j * 16 + 4 * g4;
            if (k >= g.K) continue;
            float* cp = g.C + (g.row_off ? g.row_off[n] : (int64_t)n * g.ldc) + k;
            const f32x4 v = acc[i][j] * g.alpha;
            if (g.atomic == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) unsafeAtomicAdd(cp + r, v[r]);
            } else if (g.atomic == 2) {
                *(f32x4*)cp = *(const f32x4*)cp + v;
            } else {
                *(f32x4*)cp = v;
            }
        }
    }
    if (do_colsum && g4 == 0) {
#pragma unroll
        for (int i = 0; i < FR; ++i) {
            const int n = n0 + wr * (TILE / 2) + i * 16 + l15;
            if (n < g.N) {
                unsafeAtomicAdd(g.colsum + (g.colsum_off ? g.colsum_off[n] : (int64_t)n), acs[i].x * g.alpha);
                if (g.colsum_off2 && g.colsum_off2[n] >= 0) unsafeAtomicAdd(g.colsum + g.colsum_off2[n], acs[i].x * g.alpha);
            }
        }
    }
}

template <typename HT, bool CONV, int TILE, int NG>
__global__ __launch_bounds__(256 * NG, 1) void cfm_gemm_tn_dma_kernel(const TnArgs g) {
    tn_dma_body<HT, CONV, TILE, NG>(g, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Several products in ONE launch (cfm_gemm_tn_group): the eight weight gradients of a conformer block -- 2 x (2048 x 256, 256 x 2048),
// 768 x 256, 512 x 256, 2 x 256 x 256 -- are each too small to fill 256 CUs (16 .. 128 tiles of 64 x 64) and were 8 launches of 10-17 us per
// block and micro-batch.  They do not feed the chain of input gradients, so the block's backward defers them to its end and issues them as
// one grid: workgroup b belongs to the product p with first[p] <= b < first[p+1], tile (b - first[p]) % tiles[p], M-split (b - first[p]) /
// tiles[p].  The products are ordered by rows per workgroup, longest first, so the tail of the grid is made of the short ones.
constexpr int TN_GROUP_MAX = 12;
struct TnGroupArgs {
    TnArgs p[TN_GROUP_MAX];
    int first[TN_GROUP_MAX + 1];
    int tiles[TN_GROUP_MAX];
    int n;
};

template <typename HT, int TILE, int NG>
__global__ __launch_bounds__(256 * NG, 1) void cfm_gemm_tn_group_kernel(const TnGroupArgs G) {
    const int b = (int)blockIdx.x;
    int idx = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP_MAX; ++i)
        if (i < G.n && b >= G.first[i]) idx = i;           // uniform: b is the workgroup id
    const int rel = b - G.first[idx];
    const int tiles = G.tiles[idx];
    tn_dma_body<HT, false, TILE, NG>(G.p[idx], rel % tiles, rel / tiles);
}

template <typename HT>
int launch_tn_dma(const TnArgs& a, bool conv, int tile, int splits, hipStream_t s, const char* name) {
    const int tiles = ((a.N + tile - 1) / tile) * ((a.K + tile - 1) / tile);
    CfmProfScope prof(name, s, 2.0 * a.M * (double)a.N * a.K, (double)a.M * (a.N + a.K) * 2 + 4.0 * a.N * a.K);
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    if (tile == 128) {
        if (conv) CFM_LAUNCH((cfm_gemm_tn_dma_kernel<HT, true, 128, 1>), grid, dim3(256), 0, s, a);
        else CFM_LAUNCH((cfm_gemm_tn_dma_kernel<HT, false, 128, 1>), grid, dim3(256), 0, s, a);
    } else {
        if (conv) CFM_LAUNCH((cfm_gemm_tn_dma_kernel<HT, true, 64, 2>), grid, dim3(512), 0, s, a);
        else CFM_LAUNCH((cfm_gemm_tn_dma_kernel<HT, false, 64, 2>), grid, dim3(512), 0, s, a);
    }
    return cfm_launch_status(name);
}

template <typename HT, bool SPLIT, bool A32, bool B32>
int launch_tn(const TnArgs& a, bool conv, int tile, int splits, hipStream_t s, const char* name) {
    const int tiles = ((a.N + tile - 1) / tile) * ((a.K + tile - 1) / tile);
    CfmProfScope prof(name, s, 2.0 * a.M * (double)a.N * a.K, (double)a.M * (a.N * (A32 ? 4 : 2) + a.K * (B32 ? 4 : 2)) + 4.0 * a.N * a.K);
    const dim3 grid((unsigned)tiles, (unsigned)splits), block(256);
    if (tile == 128) {
        if (conv) CFM_LAUNCH((cfm_gemm_tn_kernel<HT, SPLIT, A32, B32, true, 128>), grid, block, 0, s, a);
        else CFM_LAUNCH((cfm_gemm_tn_kernel<HT, SPLIT, A32, B32, false, 128>), grid, block, 0, s, a);
    } else {
        if (conv) CFM_LAUNCH((cfm_gemm_tn_kernel<HT, SPLIT, A32, B32, true, 64>), grid, block, 0, s, a);
        else CFM_LAUNCH((cfm_gemm_tn_kernel<HT, SPLIT, A32, B32, false, 64>), grid, block, 0, s, a);
    }
    return cfm_launch_status(name);
}

// Argument checks, tile / split choice and the kernel arguments of one product.  group_tiles > 0: the product is part of a grouped launch with
// that many output tiles in total (the split rule then looks at the group, not at the product alone).
struct TnPlan {
    TnArgs a;
    int tile, splits, tiles;
    bool dma, conv, a32, b32;
};

int plan_tn(const cfm_gemm_tn_desc* d, TnPlan& pl, int group_tiles) {
    CFM_CHECK_ARG(d && d->A && d->B && d->C, "cfm_gemm_tn: null pointer");
    CFM_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "cfm_gemm_tn: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    CFM_CHECK_ARG(d->N % 8 == 0 && d->K % 8 == 0, "cfm_gemm_tn: N=%d and K=%d must be multiples of 8", d->N, d->K);
    CFM_CHECK_ARG(d->mma_dtype == CFM_BF16 || d->mma_dtype == CFM_F16, "cfm_gemm_tn: mma_dtype must be bf16 or fp16");
    CFM_CHECK_ARG(d->a_dtype == CFM_F32 || d->a_dtype == d->mma_dtype, "cfm_gemm_tn: a_dtype must be f32 or the MFMA type");
    CFM_CHECK_ARG(d->b_dtype == CFM_F32 || d->b_dtype == d->mma_dtype, "cfm_gemm_tn: b_dtype must be f32 or the MFMA type");
    CFM_CHECK_ARG(!d->split || (d->a_dtype == CFM_F32 && d->b_dtype == CFM_F32 && d->mma_dtype == CFM_BF16),
                  "cfm_gemm_tn: split mode needs f32 operands and bf16 planes");
    CFM_CHECK_ARG(d->lda % (d->a_dtype == CFM_F32 ? 4 : 8) == 0 && d->lda >= d->N, "cfm_gemm_tn: bad lda=%lld", (long long)d->lda);
    CFM_CHECK_ARG(d->ldc % 4 == 0 && d->ldc >= d->K, "cfm_gemm_tn: bad ldc=%lld", (long long)d->ldc);
    const bool conv = d->conv_C > 0;
    if (conv) {
        CFM_CHECK_ARG(d->conv_C % 8 == 0 && d->K == 9 * d->conv_C, "cfm_gemm_tn: conv needs C %% 8 == 0 and K == 9*C");
        CFM_CHECK_ARG(d->conv_T2 == (d->conv_T1 - 3) / 2 + 1 && d->conv_F2 == (d->conv_F1 - 3) / 2 + 1 && d->conv_T2 > 0 && d->conv_F2 > 0,
                      "cfm_gemm_tn: conv output shape does not match a 3x3 stride-2 convolution");
        CFM_CHECK_ARG(d->M % (d->conv_T2 * d->conv_F2) == 0, "cfm_gemm_tn: conv M must be B*T2*F2");
    } else {
        CFM_CHECK_ARG(d->ldb % (d->b_dtype == CFM_F32 ? 4 : 8) == 0 && d->ldb >= d->K, "cfm_gemm_tn: bad ldb=%lld", (long long)d->ldb);
    }
    const int mch128 = d->split ? 32 : 64;                  // chunk rows of the 128-wide tile; the 64-wide tile stages twice as many per barrier
    const int chunks128 = (d->M + mch128 - 1) / mch128;
    // 64 x 64 tiles while the output is small (fewer than 128 tiles of 128 x 128: every weight of the d = 256 / 512 blocks), 128 x 128 for
    // the big ones (the CTC head's 5008 x 256, the front-end convolutions): then about one workgroup per CU
    const int t128 = ((d->N + 127) / 128) * ((d->K + 127) / 128);
    CFM_CHECK_ARG(d->tile == 0 || d->tile == 64 || d->tile == 128, "cfm_gemm_tn: tile must be 0 (auto), 64 or 128");
    const int tile = group_tiles > 0 ? 64 : (d->tile ? d->tile : ((t128 >= 128 || chunks128 >= 128) ? 128 : 64));   // M >= 8 k rows: enough splits of >= 4 chunks even with few big tiles
    const bool a32 = d->a_dtype == CFM_F32, b32 = d->b_dtype == CFM_F32;
    // 16-bit operands without a row mask: the LDS-DMA kernel (64-row chunks, several in flight)
    const bool dma = !a32 && !b32 && !d->split && !d->row_mask;
    const int mch = dma ? (tile == 64 ? 128 : 64) : (tile == 64 ? 2 * mch128 : mch128);
    const int chunks = (d->M + mch - 1) / mch;
    const int tiles = ((d->N + tile - 1) / tile) * ((d->K + tile - 1) / tile);
    int splits = d->splits;
    if (splits <= 0) {
        if (group_tiles > 0) {
            // a grouped launch fills the chip with its tiles; M is split only while the whole group has fewer workgroups than ~2 per CU
            splits = (512 + group_tiles / 2) / group_tiles;
            // (measured and dropped, config-3 window, ms per optimizer step: 128 x 128 tiles with four wavefronts for the whole group -- 156 workgroups in
            // one round -- 8.44 against 8.33; the same with every product split in two: 9.64, the atomics of a split cost far more than the idle CUs;
            // splitting the SMALL products of a block's group in two so that they fill the last, partly filled round of the
            // grid with half passes -- 59 us by the round count against 70 -- costs more in atomics than it saves: 8.49 against 8.37 ms per step)
        } else if (dma && tile == 128) {
            // one workgroup per CU is resident (128 KB of LDS): the grid runs in rounds of 256, and a partly filled second round costs a whole pass.
            // Fill ONE round: the front-end's second convolution (36 tiles, M = 38 k rows) 159 us with 7 splits against 220 with the 11 of the
            // rule below (396 workgroups = two rounds) and 222 with 4 (scripts/bench_conv2_wgrad.py)
            splits = tiles >= 256 ? 1 : 256 / tiles;
        } else if (dma) {
            // measured optimum at M = 2 380 / 1 300 rows (profiles/r02_gemm_tn_splits.txt): 4 splits for 16 tiles, 3 for 32, 2-3 for 48, 1-2 for 128;
            // a split's cost (atomics) is fixed and its gain shrinks with the rows it removes, so the optimum grows like sqrt(M)
            splits = (int)(16.0 / sqrt((double)tiles) * sqrt((double)d->M / 2400.0) + 0.5);
        } else {
            splits = (256 + tiles - 1) / tiles;
        }
        const int max_s = (chunks * mch + 255) / 256;       // at least 256 rows per split
        splits = splits > max_s ? max_s : splits;
    }
    splits = splits < 1 ? 1 : (splits > chunks ? chunks : splits);
    TnArgs& a = pl.a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.colsum = d->colsum; a.mask = d->row_mask; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.M = d->M; a.N = d->N; a.K = d->K; a.alpha = d->alpha;
    a.convC = d->conv_C; a.T1 = d->conv_T1; a.F1 = d->conv_F1; a.T2 = d->conv_T2; a.F2 = d->conv_F2;
    CFM_CHECK_ARG((!d->row_off && !d->colsum_off && !d->colsum_off2) || d->accumulate, "cfm_gemm_tn: row_off / colsum_off need accumulate = 1 (zero-filled by the caller)");
    CFM_CHECK_ARG(!d->colsum_off2 || d->colsum, "cfm_gemm_tn: colsum_off2 needs colsum");
    a.row_off = d->row_off; a.colsum_off = d->colsum_off; a.colsum_off2 = d->colsum_off2;
    a.chunks_per_split = (chunks + splits - 1) / splits;
    splits = (chunks + a.chunks_per_split - 1) / a.chunks_per_split;   // no empty split
    a.atomic = splits > 1 ? 1 : (d->accumulate ? 2 : 0);
    pl.tile = tile; pl.splits = splits; pl.tiles = tiles; pl.dma = dma; pl.conv = conv; pl.a32 = a32; pl.b32 = b32;
    return CFM_OK;
}

int zero_fill_tn(const cfm_gemm_tn_desc* d, const TnPlan& pl, hipStream_t s) {
    if (d->accumulate) return CFM_OK;
    if (pl.a.atomic && hipMemset2DAsync(d->C, (size_t)d->ldc * 4, 0, (size_t)d->K * 4, (size_t)d->N, s) != hipSuccess)
        return cfm_fail(CFM_ERR_LAUNCH, "cfm_gemm_tn: memset of C failed");
    if (d->colsum && hipMemsetAsync(d->colsum, 0, (size_t)d->N * 4, s) != hipSuccess)
        return cfm_fail(CFM_ERR_LAUNCH, "cfm_gemm_tn: memset of colsum failed");
    return CFM_OK;
}

}  // namespace

extern "C" int cfm_gemm_tn(const cfm_gemm_tn_desc* d, cfm_stream_t stream) {
    TnPlan pl;
    if (int rc = plan_tn(d, pl, 0)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = zero_fill_tn(d, pl, s)) return rc;
    const TnArgs& a = pl.a;
    const bool conv = pl.conv, a32 = pl.a32, b32 = pl.b32;
    const int tile = pl.tile, splits = pl.splits;
    if (pl.dma) return d->mma_dtype == CFM_BF16 ? launch_tn_dma<BF16>(a, conv, tile, splits, s, "gemm_tn_dma_bf16") : launch_tn_dma<F16>(a, conv, tile, splits, s, "gemm_tn_dma_f16");
    if (d->split) return launch_tn<BF16, true, true, true>(a, conv, tile, splits, s, "gemm_tn_bf16x3");
#define CFM_TN(HT, NAME)                                                                              \
    do {                                                                                              \
        if (a32 && b32) return launch_tn<HT, false, true, true>(a, conv, tile, splits, s, NAME);      \
        if (a32) return launch_tn<HT, false, true, false>(a, conv, tile, splits, s, NAME);            \
        if (b32) return launch_tn<HT, false, false, true>(a, conv, tile, splits, s, NAME);            \
        return launch_tn<HT, false, false, false>(a, conv, tile, splits, s, NAME);                    \
    } while (0)
    if (d->mma_dtype == CFM_BF16) CFM_TN(BF16, "gemm_tn_bf16");
    CFM_TN(F16, "gemm_tn_f16");
#undef CFM_TN
}

// n products in one launch when every one of them can take the LDS-DMA 64 x 64 kernel (16-bit operands of one type, no row mask, no f32-accurate
// split, no implicit convolution) and n <= TN_GROUP_MAX; otherwise one launch each, in order -- same results either way (the M splits may
// differ: atomics then sum in another order).
extern "C" int cfm_gemm_tn_group(const cfm_gemm_tn_desc* descs, int32_t n, cfm_stream_t stream) {
    CFM_CHECK_ARG(descs && n > 0, "cfm_gemm_tn_group: no products");
    hipStream_t s = (hipStream_t)stream;
    bool groupable = n >= 2 && n <= TN_GROUP_MAX;
    int group_tiles = 0;
    for (int i = 0; i < n && groupable; ++i) {
        const cfm_gemm_tn_desc& d = descs[i];
        groupable = d.a_dtype != CFM_F32 && d.b_dtype != CFM_F32 && d.a_dtype == descs[0].a_dtype && d.b_dtype == d.a_dtype && d.mma_dtype == d.a_dtype &&
                    !d.split && !d.row_mask && d.conv_C == 0 && (d.tile == 0 || d.tile == 64);
        group_tiles += ((d.N + 63) / 64) * ((d.K + 63) / 64);
    }
    if (!groupable) {
        for (int i = 0; i < n; ++i)
            if (int rc = cfm_gemm_tn(&descs[i], stream)) return rc;
        return CFM_OK;
    }
    TnPlan pl[TN_GROUP_MAX];
    int order[TN_GROUP_MAX];
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        if (int rc = plan_tn(&descs[i], pl[i], group_tiles)) return rc;
        if (int rc = zero_fill_tn(&descs[i], pl[i], s)) return rc;
        order[i] = i;
        flops += 2.0 * descs[i].M * (double)descs[i].N * descs[i].K;
        bytes += (double)descs[i].M * (descs[i].N + descs[i].K) * 2 + 4.0 * descs[i].N * descs[i].K;
    }
    for (int i = 1; i < n; ++i)                             // longest workgroups first (insertion sort, stable)
        for (int j = i; j > 0 && pl[order[j]].a.chunks_per_split > pl[order[j - 1]].a.chunks_per_split; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    TnGroupArgs G;
    G.n = n;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        const TnPlan& p = pl[order[i]];
        G.p[i] = p.a;
        G.first[i] = first;
        G.tiles[i] = p.tiles;
        first += p.tiles * p.splits;
    }
    for (int i = n; i < TN_GROUP_MAX; ++i) { G.p[i] = G.p[0]; G.first[i] = first; G.tiles[i] = 1; }
    G.first[TN_GROUP_MAX] = first;
    const bool bf = descs[0].mma_dtype == CFM_BF16;
    CfmProfScope prof(bf ? "gemm_tn_group_bf16" : "gemm_tn_group_f16", s, flops, bytes);
    if (bf) CFM_LAUNCH((cfm_gemm_tn_group_kernel<BF16, 64, 2>), dim3((unsigned)first), dim3(512), 0, s, G);
    else CFM_LAUNCH((cfm_gemm_tn_group_kernel<F16, 64, 2>), dim3((unsigned)first), dim3(512), 0, s, G);
    return cfm_launch_status("cfm_gemm_tn_group");
}
