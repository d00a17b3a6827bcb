/*
 * jmac_hip_testing.h -- entry points that exist in libjmac_hip_testing.so ONLY (jmac_amd/csrc/Makefile): second
 * implementations and measured-but-rejected experiments the parity tests exercise.  Nothing on the product path (libjmac_hip.so,
 * include/jmac_hip.h) declares, exports or calls them.
 *
 *  - the ATOMIC aggregation backward (mode 0 of jmac_rel_attn_aggregate_bwd_f32): compiled in with -DJMAC_TEST_ATOMIC_BWD;
 *  - jmac_gemm_nt_x3_f32: the split-bf16 GEMM of round 2 (as fast as the tuned library kernel, but the bf16 MFMA's
 *    accumulation bias broke gradient parity: not on the path).
 */
#ifndef JMAC_HIP_TESTING_H_
#define JMAC_HIP_TESTING_H_

#include "jmac_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fp32 GEMM on the bf16 matrix cores for the N-row dense products of the encoder and of the factorised layer
 * (replaces torch.mm at src/jmac_model.py:177-203 and the hoisted X [Wt|Wb|Wg] projection / its adjoint):
 *   C[M,N] = A[M,K] B[N,K]^T   ("NT": both operands k-contiguous; a weight W [K,N] is passed as its transpose),
 * fp32 in, fp32 out.  Each operand element is split into three bf16 terms (24 mantissa bits) while it is staged and the
 * six significant term pairs are accumulated in fp32: fp32-GEMM-level error at 6/16 of the fp32 MFMA's issue time.
 * K % 4 == 0, lda / ldb % 4 == 0 (16-byte rows), any M, N. */
int jmac_gemm_nt_x3_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N,
                        int64_t K, float* C, int64_t ldc, jmac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
