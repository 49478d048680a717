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

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, int64_t lda,
                                                    const float* __restrict__ B, int64_t ldb, float beta, float* __restrict__ C, int64_t ldc,
                                                    const float* __restrict__ bias, int k_per_split) {
    __shared__ float sA[KC][TS + PAD], sB[KC][TS + PAD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += KC) {
        float va[4], vb[4];
        // A tile -> sA[k][m]
        if (!TA) {   // stored [M,K]: a thread reads 4 consecutive k of one row
            const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) va[i] = (m0 + r < M && k0 + kq + i < kend) ? A[(int64_t)(m0 + r) * lda + k0 + kq + i] : 0.f;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) sA[kq + i][r] = va[i];
        } else {     // stored [K,M]: a thread reads 4 consecutive m of one k
            const int kk = tid >> 4, mq = (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) va[i] = (k0 + kk < kend && m0 + mq + i < M) ? A[(int64_t)(k0 + kk) * lda + m0 + mq + i] : 0.f;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) sA[kk][mq + i] = va[i];
        }
        if (TB) {    // stored [N,K]
            const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[i] = (n0 + r < N && k0 + kq + i < kend) ? B[(int64_t)(n0 + r) * ldb + k0 + kq + i] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sB[kq + i][r] = vb[i];
        } else {     // stored [K,N]
            const int kk = tid >> 4, nq = (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[i] = (k0 + kk < kend && n0 + nq + i < N) ? B[(int64_t)(k0 + kk) * ldb + n0 + nq + i] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) sB[kk][nq + i] = vb[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const f32x4_t a4 = *(const f32x4_t*)&sA[kk][ty * 4];
            const f32x4_t b4 = *(const f32x4_t*)&sB[kk][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
    }
    const bool split = gridDim.z > 1;
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

__global__ void sgemm_prescale_kernel(float* C, int64_t ldc, int M, int N, float beta, const float* bias) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n < N) {
        float* c = C + (int64_t)m * ldc + n;
        float v = beta != 0.f ? beta * *c : 0.f;
        if (bias) v += bias[n];
        *c = v;
    }
}
}  // namespace

extern "C" int sc_sgemm(int transa, int transb, int M, int N, int K, float alpha, const float* A, int64_t lda, const float* B, int64_t ldb,
                        float beta, float* C, int64_t ldc, const float* bias, void* stream) {
    SC_CHECK_ARG(M > 0 && N > 0 && K > 0, "sc_sgemm: empty problem M=%d N=%d K=%d", M, N, K);
    SC_CHECK_ARG(A && B && C, "sc_sgemm: null operand");
    SC_CHECK_ARG(lda >= (transa ? M : K) && ldb >= (transb ? K : N) && ldc >= N, "sc_sgemm: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((M + TS - 1) / TS) * ((N + TS - 1) / TS);
    int split = 1;
    if (tiles < 128 && K >= 2048) {               // few output tiles, long reduction: split K so the chip is not idle
        split = min(64, max(1, 512 / tiles));
        while (split > 1 && K / split < 256) --split;
    }
    int k_per_split = ((K + split - 1) / split + KC - 1) / KC * KC;
    split = (K + k_per_split - 1) / k_per_split;
    if (split > 1) {
        hipLaunchKernelGGL(sgemm_prescale_kernel, dim3((N + 255) / 256, M), dim3(256), 0, s, C, ldc, M, N, beta, bias);
        SC_CHECK_LAUNCH();
    }
    dim3 grid((N + TS - 1) / TS, (M + TS - 1) / TS, split);
#define SC_SGEMM_LAUNCH(TA, TB) hipLaunchKernelGGL((sgemm_kernel<TA, TB>), grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, k_per_split)
    if (transa) { if (transb) SC_SGEMM_LAUNCH(true, true); else SC_SGEMM_LAUNCH(true, false); }
    else        { if (transb) SC_SGEMM_LAUNCH(false, true); else SC_SGEMM_LAUNCH(false, false); }
#undef SC_SGEMM_LAUNCH
    SC_CHECK_LAUNCH();
    return 0;
}
