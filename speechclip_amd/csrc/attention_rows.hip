// Full-row multi-head attention for ANY head dimension with an arbitrary boolean key-padding mask: the arithmetic of
// torch.nn.MultiheadAttention(batch_first) as the reference's pooling heads call it on whole sequences --
//   TransformerEncoder.forward / .extract_hidden_states   (avssl/module/kw_modules/TransformerModels.py:77-96)
//   MultiheadAttentionAndNorm.forward / .extract_hidden_states (TransformerModels.py:119-129)
// with head dims 96 / 128 (8 heads of 768 / 1024) and 768 / 1024 (1 head).  The hot path does NOT use this kernel (it evaluates the CLS rows
// only, in algebraic form: sc_cls_pool_fwd); this is the boundary's analysis path (feature_extractor_s3prl), kept simple:
// one wave per query row, keys 64 at a time (lane = key), fp32 online softmax, the query in LDS (broadcast reads), P.V with lane = output
// column.  Operands bf16 (q|k|v of sc_gemm_bf16), fp32 arithmetic, bf16 output.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves = 4 consecutive query rows of one (b, h)
constexpr int MAX_HD = 1024;

__global__ __launch_bounds__(256) void attention_rows_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                             bf16_t* __restrict__ out, const uint8_t* __restrict__ pad, int B, int H, int L, int hd,
                                                             int64_t ld_qkv, int64_t ld_out, float scale_log2e) {
    extern __shared__ float sq[];                        // [ROWS_PER_BLOCK][hd] queries (pre-scaled), fp32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qblocks = (L + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
    const int b = bh / H, h = bh - b * H;
    const int i = qb * ROWS_PER_BLOCK + wave;            // this wave's query position
    if (i >= L) return;                                  // (no block-level barrier below)
    float* myq = sq + wave * hd;
    const bf16_t* qrow = q + ((int64_t)b * L + i) * ld_qkv + h * hd;
    for (int d = lane; d < hd; d += 64) myq[d] = bf2f(qrow[d]) * scale_log2e;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nd = (hd + 63) / 64;                       // output columns per lane: d = lane + 64 m
    float acc[MAX_HD / 64];
#pragma unroll
    for (int m = 0; m < MAX_HD / 64; ++m) acc[m] = 0.f;
    float run_max = -INFINITY, run_sum = 0.f;
    const uint8_t* prow = pad ? pad + (int64_t)b * L : nullptr;
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < L && !(prow && prow[j]);
        float s = -INFINITY;
        if (valid) {
            const bf16_t* krow = k + ((int64_t)b * L + j) * ld_qkv + h * hd;
            float dot = 0.f;
            int d = 0;
            for (; d + 8 <= hd; d += 8) {                // 16-byte chunks of this lane's key row; the query comes from LDS (same address: broadcast)
                const uint4 kk = *(const uint4*)(krow + d);
                dot = fmaf(lo2f(kk.x), myq[d], dot);     dot = fmaf(hi2f(kk.x), myq[d + 1], dot);
                dot = fmaf(lo2f(kk.y), myq[d + 2], dot); dot = fmaf(hi2f(kk.y), myq[d + 3], dot);
                dot = fmaf(lo2f(kk.z), myq[d + 4], dot); dot = fmaf(hi2f(kk.z), myq[d + 5], dot);
                dot = fmaf(lo2f(kk.w), myq[d + 6], dot); dot = fmaf(hi2f(kk.w), myq[d + 7], dot);
            }
            for (; d < hd; ++d) dot = fmaf(bf2f(krow[d]), myq[d], dot);
            s = dot;                                      // log2-domain score
        }
        const float cmax = wave_max(s);
        if (cmax == -INFINITY) continue;                 // a chunk of padding only (wave-uniform)
        const float nmax = fmaxf(run_max, cmax);
        const float corr = __builtin_amdgcn_exp2f(run_max - nmax);   // exp2(-inf) = 0 on the first live chunk
        const float p = valid ? __builtin_amdgcn_exp2f(s - nmax) : 0.f;
        run_sum = run_sum * corr + wave_sum(p);
        run_max = nmax;
#pragma unroll
        for (int m = 0; m < MAX_HD / 64; ++m)
            if (m < nd) acc[m] *= corr;
        const int jn = L - j0 < 64 ? L - j0 : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const float pj = __shfl(p, jj, 64);
            if (pj == 0.f) continue;                     // wave-uniform
            const bf16_t* vrow = v + ((int64_t)b * L + j0 + jj) * ld_qkv + h * hd;
#pragma unroll
            for (int m = 0; m < MAX_HD / 64; ++m) {
                const int d = lane + 64 * m;
                if (m < nd && d < hd) acc[m] = fmaf(pj, bf2f(vrow[d]), acc[m]);
            }
        }
    }
    // a row whose keys are ALL masked: torch's softmax over -inf gives NaN; the reference never builds such a row (the CLS keys are always
    // valid).  Zeros here.
    const float inv = run_sum > 0.f ? 1.0f / run_sum : 0.f;
    bf16_t* orow = out + ((int64_t)b * L + i) * ld_out + h * hd;
#pragma unroll
    for (int m = 0; m < MAX_HD / 64; ++m) {
        const int d = lane + 64 * m;
        if (m < nd && d < hd) orow[d] = f2bf(acc[m] * inv);
    }
}

}  // namespace

extern "C" int sc_attention_rows_fwd(const void* q, const void* k, const void* v, void* out, const uint8_t* key_padding_mask, int B, int H, int L,
                                     int head_dim, int64_t ld_qkv, int64_t ld_out, float scale, void* stream) {
    SC_CHECK_ARG(head_dim >= 1 && head_dim <= MAX_HD, "sc_attention_rows_fwd: head_dim=%d must be in [1, %d]", head_dim, MAX_HD);
    SC_CHECK_ARG(head_dim % 8 == 0 && ld_qkv % 8 == 0, "sc_attention_rows_fwd: head_dim and ld_qkv must be multiples of 8 (16-byte key chunks)");
    SC_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "sc_attention_rows_fwd: q/k/v must be 16-byte aligned");
    SC_CHECK_ARG(q && k && v && out, "sc_attention_rows_fwd: null operand");
    if (B <= 0 || L <= 0 || H <= 0) return 0;
    const int64_t blocks = (int64_t)B * H * ((L + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    SC_CHECK_ARG(blocks < 0x7fffffff, "sc_attention_rows_fwd: grid too large");
    hipLaunchKernelGGL(attention_rows_kernel, dim3((unsigned)blocks), dim3(ROWS_PER_BLOCK * 64), ROWS_PER_BLOCK * head_dim * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, key_padding_mask, B, H, L, head_dim, ld_qkv,
                       ld_out, scale * 1.44269504088896341f);
    SC_CHECK_LAUNCH();
    return 0;
}
