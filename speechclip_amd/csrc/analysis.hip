// Kernels of the cascaded branch's ANALYSIS surface, off the hot path:
//   sc_attention_probs_fwd -- the per-head attention probabilities torch.nn.MultiheadAttention returns with need_weights=True,
//                             average_attn_weights=False, as MultiheadAttentionAndNorm.extract_attention_map asks for them
//                             (avssl/module/kw_modules/TransformerModels.py:130-135), consumed by KW_CascadedBranch.getAttentionMap
//                             (avssl/model/kwClip.py:918-1001: the keyword rows of every head against every valid key);
//   sc_topk_rows_f32       -- torch.topk(x, K) along the last axis: the K nearest sub-words of every keyword, getAttentionMap (:990) and
//                             validation_epoch_end's de-tokenisation (:357-375).
// One wave per query row / one block per score row; simple on purpose (a validation epoch calls them a handful of times).
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

constexpr int PROB_ROWS_PER_BLOCK = 4;
constexpr int MAX_HD = 1024;

// probs[b, h, i, :] = softmax_j(scale * q_i . k_j) over the keys that are not padding; exactly 0 at padded keys (torch: exp(-inf)).
__global__ __launch_bounds__(256) void attention_probs_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, float* __restrict__ probs,
                                                              const uint8_t* __restrict__ pad, int B, int H, int L, int hd, int n_rows,
                                                              int64_t ld_qkv, float scale_log2e) {
    extern __shared__ float sq[];                        // [PROB_ROWS_PER_BLOCK][hd] queries (pre-scaled), fp32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qblocks = (n_rows + PROB_ROWS_PER_BLOCK - 1) / PROB_ROWS_PER_BLOCK;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
    const int b = bh / H, h = bh - b * H;
    const int i = qb * PROB_ROWS_PER_BLOCK + wave;
    if (i >= n_rows) return;                             // (no block-level barrier below)
    float* myq = sq + wave * hd;
    const bf16_t* qrow = q + ((int64_t)b * L + i) * ld_qkv + h * hd;
    for (int d = lane; d < hd; d += 64) myq[d] = bf2f(qrow[d]) * scale_log2e;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* prow_out = probs + (((int64_t)b * H + h) * n_rows + i) * L;
    const uint8_t* prow = pad ? pad + (int64_t)b * L : nullptr;
    float run_max = -INFINITY;
    for (int j0 = 0; j0 < L; j0 += 64) {                 // pass 1: log2-domain scores into the output row, running maximum
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j < L && !(prow && prow[j])) {
            const bf16_t* krow = k + ((int64_t)b * L + j) * ld_qkv + h * hd;
            float dot = 0.f;
            int d = 0;
            for (; d + 8 <= hd; d += 8) {
                const uint4 kk = *(const uint4*)(krow + d);
                dot = fmaf(lo2f(kk.x), myq[d], dot);     dot = fmaf(hi2f(kk.x), myq[d + 1], dot);
                dot = fmaf(lo2f(kk.y), myq[d + 2], dot); dot = fmaf(hi2f(kk.y), myq[d + 3], dot);
                dot = fmaf(lo2f(kk.z), myq[d + 4], dot); dot = fmaf(hi2f(kk.z), myq[d + 5], dot);
                dot = fmaf(lo2f(kk.w), myq[d + 6], dot); dot = fmaf(hi2f(kk.w), myq[d + 7], dot);
            }
            for (; d < hd; ++d) dot = fmaf(bf2f(krow[d]), myq[d], dot);
            s = dot;
        }
        if (j < L) prow_out[j] = s;
        run_max = fmaxf(run_max, s);
    }
    run_max = wave_max(run_max);
    float sum = 0.f;
    if (run_max != -INFINITY)
        for (int j = lane; j < L; j += 64) {             // pass 2: every lane re-reads what IT wrote (same j): no fence needed
            const float p = __builtin_amdgcn_exp2f(prow_out[j] - run_max);
            prow_out[j] = p;
            sum += p;
        }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;      // a row with every key masked: zeros (torch gives NaN; the reference never builds one)
    for (int j = lane; j < L; j += 64) prow_out[j] = run_max == -INFINITY ? 0.f : prow_out[j] * inv;
}

// K rounds of a block-wide arg-max; round r picks the largest element that comes strictly after round r-1's pick in (value descending,
// index ascending) order, so nothing has to be marked and ties resolve to the lowest index.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, int64_t ld, int V, int K, float* __restrict__ vals,
                                                        int32_t* __restrict__ idx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* row = x + (int64_t)blockIdx.x * ld;
    float pv = INFINITY;
    int pi = -1;
    for (int r = 0; r < K; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = threadIdx.x; j < V; j += 256) {
            const float v = row[j];
            const bool eligible = v < pv || (v == pv && j > pi);
            if (eligible && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        bv = sv[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        __syncthreads();
        if (threadIdx.x == 0) {
            vals[(int64_t)blockIdx.x * K + r] = bv;
            idx[(int64_t)blockIdx.x * K + r] = bi == 0x7fffffff ? -1 : bi;     // fewer than K comparable entries (all -inf below / NaN): -1
        }
        pv = bv; pi = bi;
    }
}

}  // namespace

extern "C" int sc_attention_probs_fwd(const void* q, const void* k, float* probs, const uint8_t* key_padding_mask, int B, int H, int L, int head_dim,
                                      int n_rows, int64_t ld_qkv, float scale, void* stream) {
    SC_CHECK_ARG(head_dim >= 1 && head_dim <= MAX_HD, "sc_attention_probs_fwd: head_dim=%d must be in [1, %d]", head_dim, MAX_HD);
    SC_CHECK_ARG(head_dim % 8 == 0 && ld_qkv % 8 == 0, "sc_attention_probs_fwd: head_dim and ld_qkv must be multiples of 8 (16-byte key chunks)");
    SC_CHECK_ARG((((uintptr_t)q | (uintptr_t)k) & 15) == 0, "sc_attention_probs_fwd: q/k must be 16-byte aligned");
    SC_CHECK_ARG(q && k && probs, "sc_attention_probs_fwd: null operand");
    SC_CHECK_ARG(n_rows >= 0 && n_rows <= L, "sc_attention_probs_fwd: n_rows=%d must be in [0, L=%d]", n_rows, L);
    if (B <= 0 || L <= 0 || H <= 0 || n_rows == 0) return 0;
    const int64_t blocks = (int64_t)B * H * ((n_rows + PROB_ROWS_PER_BLOCK - 1) / PROB_ROWS_PER_BLOCK);
    SC_CHECK_ARG(blocks < 0x7fffffff, "sc_attention_probs_fwd: grid too large");
    hipLaunchKernelGGL(attention_probs_kernel, dim3((unsigned)blocks), dim3(PROB_ROWS_PER_BLOCK * 64), PROB_ROWS_PER_BLOCK * head_dim * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, probs, key_padding_mask, B, H, L, head_dim, n_rows, ld_qkv,
                       scale * 1.44269504088896341f);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_topk_rows_f32(const float* x, int64_t ld, int64_t rows, int V, int K, float* vals, int32_t* idx, void* stream) {
    SC_CHECK_ARG(x && vals && idx, "sc_topk_rows_f32: null operand");
    SC_CHECK_ARG(K >= 1 && K <= V, "sc_topk_rows_f32: K=%d must be in [1, V=%d]", K, V);
    SC_CHECK_ARG(ld >= V && rows < 0x7fffffff, "sc_topk_rows_f32: ld=%lld < V=%d or too many rows", (long long)ld, V);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, ld, V, K, vals, idx);
    SC_CHECK_LAUNCH();
    return 0;
}
