// Training kernels of the cascaded tail (kwClip.py:697-916): the backward path from the CLIP text feature to
// the keyword embeddings -- causal attention backward over the K+2 live positions, QuickGELU, the straight-through VQ
// (my_vector_quantizer.py:133-141), the cosine-similarity backward and the train-mode keyword BatchNorm (kw_bn.py:122-131).  All fp32;
// the dense products around them run on the MFMA GEMM with the fp32 gradient split into (hi | lo) bf16 terms (split_hilo_kernel).  Sizes: B x (K+2) = 2560 text rows, B x K = 2048 keyword rows, V <= 49408 sub-words.
#include "common.h"

namespace {

constexpr int MAXL = 16;

// One block (one wave) per (sequence, head): dqkv from d(attention output), probabilities recomputed from the saved bf16 qkv.
// qkv bf16 [B*L, 3W] (q | k | v, head h at columns h*64), dout f32 [B*L, W], dqkv f32 [B*L, 3W].  hd = 64 = the wave.
__global__ __launch_bounds__(64) void attn_small_bwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                            int L, int W, int causal, float scale) {
    __shared__ float sq[MAXL][64], sk[MAXL][64], sv[MAXL][64], sdo[MAXL][64];
    __shared__ float sp[MAXL][MAXL], sds[MAXL][MAXL];
    const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int64_t row0 = (int64_t)b * L;
    for (int i = 0; i < L; ++i) {
        const bf16_t* r = qkv + (row0 + i) * 3 * W + h * 64 + lane;
        sq[i][lane] = __uint_as_float((uint32_t)r[0] << 16);
        sk[i][lane] = __uint_as_float((uint32_t)r[W] << 16);
        sv[i][lane] = __uint_as_float((uint32_t)r[2 * W] << 16);
        sdo[i][lane] = dout[(row0 + i) * W + h * 64 + lane];
    }
    __syncthreads();
    // scores and dP for the (i, j) pairs, one pair per lane and round
    for (int p = lane; p < L * L; p += 64) {
        const int i = p / L, j = p % L;
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < 64; ++d) {
            const int dd = (d + lane) & 63;   // rotate: lanes of a round start on different banks
            s += sq[i][dd] * sk[j][dd];
            dp += sdo[i][dd] * sv[j][dd];
        }
        const bool live = !causal || j <= i;
        sp[i][j] = live ? s * scale : -INFINITY;
        sds[i][j] = live ? dp : 0.f;
    }
    __syncthreads();
    if (lane < L) {      // row softmax and dS = P (dP - sum P dP)
        const int i = lane;
        float mx = -INFINITY;
        for (int j = 0; j < L; ++j) mx = fmaxf(mx, sp[i][j]);
        float den = 0.f;
        for (int j = 0; j < L; ++j) { const float e = __expf(sp[i][j] - mx); sp[i][j] = e; den += e; }
        float dot = 0.f;
        for (int j = 0; j < L; ++j) { sp[i][j] /= den; dot += sp[i][j] * sds[i][j]; }
        for (int j = 0; j < L; ++j) sds[i][j] = sp[i][j] * (sds[i][j] - dot) * scale;
    }
    __syncthreads();
    for (int i = 0; i < L; ++i) {
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < L; ++j) {
            dq += sds[i][j] * sk[j][lane];      // dq_i = sum_j dS_ij k_j
            dk += sds[j][i] * sq[j][lane];      // dk_i = sum_j dS_ji q_j
            dv += sp[j][i] * sdo[j][lane];      // dv_i = sum_j P_ji do_j
        }
        float* o = dqkv + (row0 + i) * 3 * W + h * 64 + lane;
        o[0] = dq; o[W] = dk; o[2 * W] = dv;
    }
}

// QuickGELU x * sigmoid(1.702 x) (clip/model.py QuickGELU): forward to f32 or bf16; backward: dh *= s (1 + 1.702 x (1 - s)).
__global__ __launch_bounds__(256) void quickgelu_kernel(const float* __restrict__ z, void* __restrict__ y, int64_t n, int backward, int out_bf16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = z[i];
    const float s = 1.0f / (1.0f + __expf(-1.702f * x));
    if (backward) {
        ((float*)y)[i] *= s * (1.0f + 1.702f * x * (1.0f - s));
    } else if (out_bf16) {
        const uint32_t u = pack2bf(x * s, 0.f);
        ((bf16_t*)y)[i] = (bf16_t)(u & 0xffffu);
    } else {
        ((float*)y)[i] = x * s;
    }
}

// out bf16 [M, 2K] = (hi | lo) with hi = bf16(a), lo = bf16(a - hi): the two-term operand of the backward products (a @ W^T = [hi|lo] @ [W|W]^T,
// ~16 mantissa bits of the gradient survive the bf16 MFMA GEMM).  K % 4 == 0.
__global__ __launch_bounds__(256) void split_hilo_kernel(const float* __restrict__ a, int64_t lda, bf16_t* __restrict__ out, int64_t M, int K, int nblk) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kq = K / 4;
    if (i >= M * kq) return;
    const int64_t r = i / kq;
    const int c = (int)(i % kq) * 4;
    const f32x4_t v = *(const f32x4_t*)(a + r * lda + c);
    uint2 h, l;
    h.x = pack2bf(v[0], v[1]); h.y = pack2bf(v[2], v[3]);
    l.x = pack2bf(v[0] - lo2f(h.x), v[1] - hi2f(h.x));
    l.y = pack2bf(v[2] - lo2f(h.y), v[3] - hi2f(h.y));
    *(uint2*)(out + r * nblk * K + c) = h;
    *(uint2*)(out + r * nblk * K + K + c) = l;
    if (nblk == 3) *(uint2*)(out + r * nblk * K + 2 * K + c) = h;     // (hi | lo | hi): pairs with a table laid out (hi | hi | lo)
}

struct MaskIds { int n; int id[8]; };
__device__ __forceinline__ bool masked(int v, const MaskIds& m) {
    for (int i = 0; i < m.n; ++i)
        if (m.id[i] == v) return true;
    return false;
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

// Straight-through VQ backward, one block per keyword row: p = softmax(cos / temp) over the unmasked sub-words,
// dcos = p (dprob - sum p dprob) / temp  (in place over dprob), rowdot[r] = sum_v dcos cos  (for the cosine backward).
__global__ __launch_bounds__(256) void vq_st_bwd_kernel(const float* __restrict__ cosv, float* __restrict__ dprob, float* __restrict__ rowdot, int V,
                                                        float inv_temp, MaskIds mk) {
    __shared__ float s_red[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* c = cosv + (int64_t)r * V;
    float* g = dprob + (int64_t)r * V;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256)
        if (!masked(v, mk)) mx = fmaxf(mx, c[v]);
    mx = block_max(mx, s_red);
    float den = 0.f, num = 0.f;
    for (int v = tid; v < V; v += 256)
        if (!masked(v, mk)) { const float e = __expf((c[v] - mx) * inv_temp); den += e; num += e * g[v]; }
    den = block_sum(den, s_red);
    num = block_sum(num, s_red);
    const float dot = num / den;
    float rd = 0.f;
    for (int v = tid; v < V; v += 256) {
        float d = 0.f;
        if (!masked(v, mk)) { const float p = __expf((c[v] - mx) * inv_temp) / den; d = p * (g[v] - dot) * inv_temp; rd += d * c[v]; }
        g[v] = d;
    }
    rd = block_sum(rd, s_red);
    if (tid == 0) rowdot[r] = rd;
}

// da = (G - rowdot * a/|a|) / |a|,  G = dcos @ (emb / |emb|)   (d/da of a.e / (|a||e|))
__global__ __launch_bounds__(256) void cosine_bwd_finish_kernel(const float* __restrict__ a, const float* __restrict__ G, const float* __restrict__ rowdot,
                                                                float* __restrict__ da, int E, float eps) {
    __shared__ float s_red[4];
    const int r = blockIdx.x;
    const float* ar = a + (int64_t)r * E;
    float q = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) q += ar[e] * ar[e];
    q = block_sum(q, s_red);
    const float inv = 1.0f / fmaxf(sqrtf(q), eps), rd = rowdot[r];
    for (int e = threadIdx.x; e < E; e += 256) da[(int64_t)r * E + e] = (G[(int64_t)r * E + e] - rd * ar[e] * inv) * inv;
}

// Keyword BatchNorm, train mode.  Data x f32 [B, K, E] (column j = k*E + e); the reference flattens (B, E, K), so the parameter / running
// statistic of data column j lives at index e*K + k.  One thread per column, rows streamed (coalesced across the block).
__global__ __launch_bounds__(256) void kw_bn_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                              float* __restrict__ run_mean, float* __restrict__ run_var, int B, int K, int E,
                                                              float momentum, float eps) {
    const int j = blockIdx.x * 256 + threadIdx.x, C = K * E;
    if (j >= C) return;
    const int pidx = (j % E) * K + j / E;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[(int64_t)b * C + j];
    const float mean = s / (float)B;
    float q = 0.f;
    for (int b = 0; b < B; ++b) { const float d = x[(int64_t)b * C + j] - mean; q += d * d; }
    const float var = q / (float)B, rstd = rsqrtf(var + eps);
    const float g = gamma[pidx], bt = beta[pidx];
    for (int b = 0; b < B; ++b) y[(int64_t)b * C + j] = (x[(int64_t)b * C + j] - mean) * rstd * g + bt;
    mean_out[j] = mean;
    rstd_out[j] = rstd;
    if (run_mean) {    // nn.BatchNorm1d: running_var takes the unbiased estimate
        run_mean[pidx] = (1.f - momentum) * run_mean[pidx] + momentum * mean;
        run_var[pidx] = (1.f - momentum) * run_var[pidx] + momentum * (B > 1 ? q / (float)(B - 1) : var);
    }
}

__global__ __launch_bounds__(256) void kw_bn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int K, int E) {
    const int j = blockIdx.x * 256 + threadIdx.x, C = K * E;
    if (j >= C) return;
    const int pidx = (j % E) * K + j / E;
    const float m = mean[j], rs = rstd[j], g = gamma[pidx];
    float sg = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(int64_t)b * C + j];
        sg += d * (x[(int64_t)b * C + j] - m) * rs;
        sb += d;
    }
    const float invB = 1.0f / (float)B;
    for (int b = 0; b < B; ++b) {
        const float xh = (x[(int64_t)b * C + j] - m) * rs;
        dx[(int64_t)b * C + j] = g * rs * (dy[(int64_t)b * C + j] - invB * (sb + xh * sg));
    }
    if (dgamma) dgamma[pidx] = sg;
    if (dbeta) dbeta[pidx] = sb;
}

}  // namespace

extern "C" int sc_attn_small_bwd(const void* qkv, const float* dout, float* dqkv, int B, int L, int heads, int head_dim, int causal, void* stream) {
    SC_CHECK_ARG(head_dim == 64, "sc_attn_small_bwd: head_dim=%d (CLIP text towers use 64)", head_dim);
    SC_CHECK_ARG(L >= 1 && L <= MAXL, "sc_attn_small_bwd: L=%d out of range (<= %d live positions)", L, MAXL);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B, heads), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv, dout, dqkv, L, heads * 64, causal,
                       1.0f / sqrtf((float)head_dim));
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_quickgelu_f32(const float* z, void* y_or_dh, int64_t n, int backward, int out_bf16, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(quickgelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, y_or_dh, n, backward, out_bf16);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_vq_st_bwd(const float* cos_scores, float* dprob_inout, float* rowdot, int R, int V, float temp, const int* mask_ids, int n_mask,
                            void* stream) {
    SC_CHECK_ARG(n_mask >= 0 && n_mask <= 8, "sc_vq_st_bwd: n_mask=%d out of range", n_mask);
    SC_CHECK_ARG(temp > 0.f, "sc_vq_st_bwd: temperature %g must be positive", (double)temp);
    if (R <= 0) return 0;
    MaskIds mk; mk.n = n_mask;
    for (int i = 0; i < n_mask; ++i) mk.id[i] = mask_ids[i];
    hipLaunchKernelGGL(vq_st_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, cos_scores, dprob_inout, rowdot, V, 1.0f / temp, mk);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_cosine_bwd_finish(const float* a, const float* G, const float* rowdot, float* da, int R, int E, float eps, void* stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(cosine_bwd_finish_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, a, G, rowdot, da, E, eps);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_kw_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean_out, float* rstd_out, float* running_mean,
                                  float* running_var, int B, int K, int E, float momentum, float eps, void* stream) {
    SC_CHECK_ARG(B >= 1 && K >= 1 && E >= 1, "sc_kw_bn_train_fwd: bad shape B=%d K=%d E=%d", B, K, E);
    SC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "sc_kw_bn_train_fwd: running_mean and running_var go together");
    hipLaunchKernelGGL(kw_bn_train_fwd_kernel, dim3((K * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean_out, rstd_out, running_mean,
                       running_var, B, K, E, momentum, eps);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_kw_bn_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                            int B, int K, int E, void* stream) {
    SC_CHECK_ARG(B >= 1 && K >= 1 && E >= 1, "sc_kw_bn_bwd: bad shape B=%d K=%d E=%d", B, K, E);
    hipLaunchKernelGGL(kw_bn_bwd_kernel, dim3((K * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, dy, gamma, mean, rstd, dx, dgamma, dbeta, B, K, E);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_split_hilo_bf16(const float* a, int64_t lda, void* out, int64_t M, int K, int nblk, void* stream) {
    SC_CHECK_ARG(K > 0 && K % 4 == 0 && lda % 4 == 0, "sc_split_hilo_bf16: K=%d and lda must be multiples of 4", K);
    SC_CHECK_ARG(nblk == 2 || nblk == 3, "sc_split_hilo_bf16: nblk=%d must be 2 (hi|lo) or 3 (hi|lo|hi)", nblk);
    if (M <= 0) return 0;
    const int64_t n = M * (K / 4);
    hipLaunchKernelGGL(split_hilo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, lda, (bf16_t*)out, M, K, nblk);
    SC_CHECK_LAUNCH();
    return 0;
}
