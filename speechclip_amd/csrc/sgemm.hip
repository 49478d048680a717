// fp32 SIMT GEMM for the trainable tail: every dense product of the parallel branch's forward-for-training
// and backward has only B (= pairs per GPU, 256) rows on one side -- 6.5 GFLOP per step in total -- so these run in full fp32 on the
// vector ALUs straight from the fp32 master weights (no bf16 weight copies to refresh after each optimizer step) and the MFMA kernel
// stays reserved for the frame-level work.
//   C[M,N] = alpha * op(A)[M,K] . op(B)[K,N] + beta * C (+ bias[n])      row-major, any sizes / leading dimensions
//   transa = 0: A stored [M,K];  1: A stored [K,M].   transb = 0: B stored [K,N];  1: B stored [N,K] (nn.Linear weight: y = x W^T)
// 64 x 64 tile, 16-deep LDS stages, 256 threads x (4 x 4) outputs.  split_k > 1: blockIdx.z owns a K range and accumulates with
// atomics into C, which the caller has pre-scaled (used for the weight gradients dW = dY^T X where M, N are small and K = rows is long).
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {
constexpr int TS = 64, KC = 16, PAD = 4;

// 4 consecutive floats starting at p[0] (elements >= n_valid read as 0); one 16-byte load when `vec` (aligned operand) and all 4 are valid
__device__ __forceinline__ void load4(const float* __restrict__ p, int n_valid, bool vec, float (&v)[4]) {
    if (vec && n_valid >= 4) {
        const f32x4_t t = *(const f32x4_t*)p;
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = i < n_valid ? p[i] : 0.f;
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, int64_t lda, int64_t sA,
                                                    const float* __restrict__ B, int64_t ldb, int64_t sB, float beta, float* __restrict__ C, int64_t ldc,
                                                    int64_t sC, const float* __restrict__ bias, int64_t sBias, int k_per_split, int nsplit, int vec_a, int vec_b) {
    __shared__ float sAs[KC][TS + PAD], sBs[KC][TS + PAD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    const int bz = blockIdx.z / nsplit, ks = blockIdx.z - bz * nsplit;
    A += bz * sA; B += bz * sB; C += bz * sC;
    if (bias) bias += bz * sBias;
    const int kbeg = ks * k_per_split, kend = min(K, kbeg + k_per_split);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += KC) {
        float va[4], vb[4];
        if (!TA) {   // stored [M,K]: a thread reads 4 consecutive k of one row
            const int r = tid >> 2, kq = (tid & 3) * 4;
            load4(A + (int64_t)(m0 + r) * lda + k0 + kq, m0 + r < M ? kend - (k0 + kq) : 0, vec_a, va);
        } else {     // stored [K,M]: a thread reads 4 consecutive m of one k
            const int kk = tid >> 4, mq = (tid & 15) * 4;
            load4(A + (int64_t)(k0 + kk) * lda + m0 + mq, k0 + kk < kend ? M - (m0 + mq) : 0, vec_a, va);
        }
        if (TB) {    // stored [N,K]
            const int r = tid >> 2, kq = (tid & 3) * 4;
            load4(B + (int64_t)(n0 + r) * ldb + k0 + kq, n0 + r < N ? kend - (k0 + kq) : 0, vec_b, vb);
        } else {     // stored [K,N]
            const int kk = tid >> 4, nq = (tid & 15) * 4;
            load4(B + (int64_t)(k0 + kk) * ldb + n0 + nq, k0 + kk < kend ? N - (n0 + nq) : 0, vec_b, vb);
        }
        __syncthreads();
        if (!TA) { const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) sAs[kq + i][r] = va[i];
        } else { const int kk = tid >> 4, mq = (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) sAs[kk][mq + i] = va[i];
        }
        if (TB) { const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) sBs[kq + i][r] = vb[i];
        } else { const int kk = tid >> 4, nq = (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) sBs[kk][nq + i] = vb[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const f32x4_t a4 = *(const f32x4_t*)&sAs[kk][ty * 4];
            const f32x4_t b4 = *(const f32x4_t*)&sBs[kk][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
    }
    const bool split = nsplit > 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float* c = C + (int64_t)m * ldc + n;
            float v = alpha * acc[i][j];
            if (split) {
                atomicAdd(c, v);
            } else {
                if (bias) v += bias[n];
                if (beta != 0.f) v += beta * *c;
                *c = v;
            }
        }
    }
}

__global__ void sgemm_prescale_kernel(float* C, int64_t ldc, int64_t sC, int M, int N, float beta, const float* bias, int64_t sBias) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y, bz = blockIdx.z;
    if (n < N) {
        float* c = C + bz * sC + (int64_t)m * ldc + n;
        float v = beta != 0.f ? beta * *c : 0.f;
        if (bias) v += bias[bz * sBias + n];
        *c = v;
    }
}
}  // namespace

// batch > 1: operand b of the batch starts at A + b*strideA etc. (the per-head products of the attention block in ONE launch)
extern "C" int sc_sgemm_batched(int transa, int transb, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t strideA,
                                const float* B, int64_t ldb, int64_t strideB, float beta, float* C, int64_t ldc, int64_t strideC, const float* bias,
                                int64_t strideBias, int batch, void* stream) {
    SC_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "sc_sgemm: empty problem M=%d N=%d K=%d batch=%d", M, N, K, batch);
    SC_CHECK_ARG(A && B && C, "sc_sgemm: null operand");
    SC_CHECK_ARG(lda >= (transa ? M : K) && ldb >= (transb ? K : N) && ldc >= N, "sc_sgemm: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((M + TS - 1) / TS) * ((N + TS - 1) / TS) * batch;
    int split = 1;
    if (tiles < 128 && K >= 2048) {               // few output tiles, long reduction: split K so the chip is not idle
        split = min(64, max(1, 512 / tiles));
        while (split > 1 && K / split < 256) --split;
    }
    int k_per_split = ((K + split - 1) / split + KC - 1) / KC * KC;
    split = (K + k_per_split - 1) / k_per_split;
    SC_CHECK_ARG((int64_t)batch * split <= 65535, "sc_sgemm: batch x split-K = %lld exceeds grid.z", (long long)batch * split);
    if (split > 1) {
        hipLaunchKernelGGL(sgemm_prescale_kernel, dim3((N + 255) / 256, M, batch), dim3(256), 0, s, C, ldc, strideC, M, N, beta, bias, strideBias);
        SC_CHECK_LAUNCH();
    }
    const int vec_a = ((uintptr_t)A % 16 == 0) && lda % 4 == 0 && strideA % 4 == 0;
    const int vec_b = ((uintptr_t)B % 16 == 0) && ldb % 4 == 0 && strideB % 4 == 0;
    dim3 grid((N + TS - 1) / TS, (M + TS - 1) / TS, split * batch);
#define SC_SGEMM_LAUNCH(TA, TB) hipLaunchKernelGGL((sgemm_kernel<TA, TB>), grid, dim3(256), 0, s, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, bias, strideBias, k_per_split, split, vec_a, vec_b)
    if (transa) { if (transb) SC_SGEMM_LAUNCH(true, true); else SC_SGEMM_LAUNCH(true, false); }
    else        { if (transb) SC_SGEMM_LAUNCH(false, true); else SC_SGEMM_LAUNCH(false, false); }
#undef SC_SGEMM_LAUNCH
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_sgemm(int transa, int transb, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B, int64_t ldb,
                        float beta, float* C, int64_t ldc, const float* bias, void* stream) {
    return sc_sgemm_batched(transa, transb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, bias, 0, 1, stream);
}
