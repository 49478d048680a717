/*
 * speechclip_hip.h -- C ABI of libspeechclip_hip.so: the MI355X (gfx950) kernels behind the
 * SpeechCLIP forward / contrastive hot path.
 *
 * The reference (atosystem/SpeechCLIP) has NO native boundary: every op on this path is executed
 * inside fairseq / openai-clip / torch.nn.  Each entry point below therefore
 * cites the reference call site whose arithmetic it replaces; INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - every pointer is a DEVICE pointer unless named host_*; the caller owns all memory
 *     (including workspaces) and has already selected the device.
 *   - `stream` is a hipStream_t passed as void*; kernels are stateless, re-entrant and launch only
 *     on that stream (safe under one-process-per-GPU and one-thread-per-GPU callers).
 *   - return 0 on success, negative on error; sc_last_error() returns a thread-local message.
 *   - bf16 tensors are raw uint16 storage ("bf16"); "f32" is IEEE float.  Rows are contiguous;
 *     leading dimensions (ld*) are in ELEMENTS.
 */
#ifndef SPEECHCLIP_HIP_H
#define SPEECHCLIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int sc_abi_version(void);
const char* sc_last_error(void);

/* ---- GEMM family (MFMA bf16, fp32 accumulate) ------------------------------------------------
 * C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] ) (+ residual[M,N])
 * Replaces every nn.Linear / Conv1d-as-GEMM / patch-projection on the path:
 *   fairseq q/k/v/out_proj, fc1, fc2, post_extract_proj  (speech_encoder_plus.py:52,:84-85 call sites),
 *   ConvFeatureExtractionModel layers 1-6 as implicit GEMM over a channels-last activation with
 *   overlapping rows (lda < K)                           (speech_encoder_plus.py:75),
 *   CLIP c_fc / c_proj / in_proj / out_proj / conv1      (clip_official.py:209),
 *   parallel/cascaded branch linears                    (kwClip.py:1097-1104, :877-883).
 * flags: see SC_GEMM_* below.  K must be a multiple of 64; A rows may overlap (lda < K) and the
 * caller guarantees A is readable for rows [0, M) x [0, K).
 * Batched form: grid over `batch`; batch z uses A + z*strideA, W + (z % w_mod)*strideW,
 * C + z*strideC (bias/residual follow W / C respectively: bias + (z % w_mod)*N).
 */
#define SC_ACT_NONE 0
#define SC_ACT_GELU 1       /* erf GELU (fairseq "gelu", torch F.gelu) */
#define SC_ACT_QUICKGELU 2  /* x*sigmoid(1.702x) (openai CLIP) */
#define SC_GEMM_ACT_MASK 0x3
#define SC_GEMM_OUT_F32 0x10      /* C (and residual, if given) are f32 instead of bf16 */
#define SC_GEMM_RES_AFTER_ACT 0x0 /* residual is always added after the activation */

int sc_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                 const float* bias, const void* residual, int64_t ldr,
                 int64_t M, int N, int K, int flags, void* stream);

int sc_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw,
                         int64_t strideW, int w_mod, void* C, int64_t ldc, int64_t strideC,
                         const float* bias, int64_t M, int N, int K, int batch, int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHCLIP_HIP_H */
