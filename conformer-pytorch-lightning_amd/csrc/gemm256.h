// gemm256.h -- internal interface between gemm.hip (cfm_gemm's dispatch) and gemm256.hip (the 256 x 256 LDS-DMA kernel).
#pragma once
#include "cfm_common.h"

struct Gemm256Args {
    const u16* A;          // [M, lda] 16-bit activations, or (convC > 0) a channels-last image [B, T1, F1, convC]
    const u16* W;          // [N, K] 16-bit weights
    const float* bias;     // f32 [N] or null
    void* C;               // [M, ldc] in c_dtype
    int64_t lda, ldc;
    int M, N, K;
    int c_dtype, act;      // act: none / SiLU / ReLU
    int convC, T1, F1, T2, F2;
};

bool cfm_gemm256_eligible(const Gemm256Args& a, bool w_bf16);
int cfm_gemm256_launch(const Gemm256Args& a, bool w_bf16, hipStream_t s);
