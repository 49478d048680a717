// Cascaded-branch extras (fp32, small): keyword BatchNorm (eval affine), cosine scores against the sub-word
// embedding table, the hard vector-quantiser statistics, and the embedding-row gather.
// Everything that decides an arg-max stays fp32 so the chosen sub-word matches the fp32 reference.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

__global__ __launch_bounds__(256) void kw_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, int64_t total, int K, int D) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / D;
    const int d = (int)(i - r * D), k = (int)(r % K);
    out[i] = fmaf(x[i], scale[k * D + d], shift[k * D + d]);
}

__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, float* __restrict__ inv, int64_t rows, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float q = 0.f;
    for (int e = lane; e < E; e += 64) { float v = x[row * E + e]; q += v * v; }
    q = wave_sum(q);
    if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(q), eps);
}

constexpr int TS = 64, KC = 16, PAD = 4;
// out[i,j] = (a_i . e_j) * inv_a[i] * inv_e[j]
__global__ __launch_bounds__(256) void cosine_tile_kernel(const float* __restrict__ a, const float* __restrict__ e, const float* __restrict__ inv_a,
                                                          const float* __restrict__ inv_e, float* __restrict__ out, int R, int V, int E) {
    __shared__ float sA[KC][TS + PAD], sB[KC][TS + PAD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    for (int k0 = 0; k0 < E; k0 += KC) {
        f32x4_t va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        if (m0 + lr < R && k0 + lk < E) va = *(const f32x4_t*)(a + (int64_t)(m0 + lr) * E + k0 + lk);
        if (n0 + lr < V && k0 + lk < E) vb = *(const f32x4_t*)(e + (int64_t)(n0 + lr) * E + k0 + lk);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { sA[lk + i][lr] = va[i]; sB[lk + i][lr] = vb[i]; }
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = m0 + ty * 4 + i;
        if (gi >= R) continue;
        const float ia = inv_a[gi];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = n0 + tx * 4 + j;
            if (gj < V) out[(int64_t)gi * V + gj] = acc[i][j] * ia * inv_e[gj];
        }
    }
}


// Exact re-evaluation of the near-maximal entries of a score row.  The MFMA path (sc_gemm_bf16 on three-term bf16 splits of the normalised
// operands) is accurate to ~1e-5; every entry within `delta` of the row maximum is recomputed here as (a . e) inv|a| inv|e| in fp32, so the
// arg-max -- the sub-word the quantiser picks -- is decided by fp32 arithmetic as in the reference.  One block per row.
constexpr int MAXCAND = 128;
__global__ __launch_bounds__(256) void cosine_refine_kernel(float* __restrict__ cosv, const float* __restrict__ a, const float* __restrict__ e, int V, int E,
                                                            float delta, float eps) {
    __shared__ float s_red[8];
    __shared__ int s_cand[MAXCAND];
    __shared__ int s_n;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* row = cosv + (int64_t)r * V;
    const float* ar = a + (int64_t)r * E;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) mx = fmaxf(mx, row[v]);
    mx = wave_max(mx);
    if (lane == 0) s_red[wv] = mx;
    if (tid == 0) s_n = 0;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    for (int v = tid; v < V; v += 256)
        if (row[v] >= mx - delta) { const int i = atomicAdd(&s_n, 1); if (i < MAXCAND) s_cand[i] = v; }
    float qa = 0.f;
    for (int k = tid; k < E; k += 256) qa += ar[k] * ar[k];
    qa = wave_sum(qa);
    __syncthreads();
    if (lane == 0) s_red[4 + wv] = qa;
    __syncthreads();
    const float inv_a = 1.0f / fmaxf(sqrtf(s_red[4] + s_red[5] + s_red[6] + s_red[7]), eps);
    const int n = s_n < MAXCAND ? s_n : MAXCAND;
    for (int c = 0; c < n; ++c) {
        const int v = s_cand[c];
        const float* ev = e + (int64_t)v * E;
        float d = 0.f, q = 0.f;
        for (int k = tid; k < E; k += 256) { const float x = ev[k]; d = fmaf(ar[k], x, d); q = fmaf(x, x, q); }
        d = wave_sum(d); q = wave_sum(q);
        __syncthreads();
        if (lane == 0) { s_red[wv] = d; s_red[4 + wv] = q; }
        __syncthreads();
        if (tid == 0) row[v] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) * inv_a * (1.0f / fmaxf(sqrtf(s_red[4] + s_red[5] + s_red[6] + s_red[7]), eps));
    }
}

__device__ __forceinline__ bool is_masked(int v, const int* msk, int nmsk) {
    for (int i = 0; i < nmsk; ++i)
        if (msk[i] == v) return true;
    return false;
}

struct MaskIds { int n; int id[8]; };

// one block per row: arg-max (first index on ties), row max, log-sum-exp, entropy -sum p log(p + 1e-9)
__global__ __launch_bounds__(256) void vq_row_kernel(const float* __restrict__ x, int64_t* __restrict__ targets, float* __restrict__ rmax,
                                                     float* __restrict__ rsum, float* __restrict__ rent, int* __restrict__ hist, int V, MaskIds mk) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_red[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* row = x + (int64_t)r * V;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        if (is_masked(v, mk.id, mk.n)) continue;
        const float t = row[v];
        if (t > best || (t == best && v < bi)) { best = t; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { s_val[wv] = best; s_idx[wv] = bi; }
    __syncthreads();
    best = s_val[0]; bi = s_idx[0];
    for (int w = 1; w < 4; ++w)
        if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) { best = s_val[w]; bi = s_idx[w]; }
    float se = 0.f;
    for (int v = tid; v < V; v += 256)
        if (!is_masked(v, mk.id, mk.n)) se += __expf(row[v] - best);
    se = wave_sum(se);
    if (lane == 0) s_red[wv] = se;
    __syncthreads();
    const float tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    float en = 0.f;
    for (int v = tid; v < V; v += 256)
        if (!is_masked(v, mk.id, mk.n)) { const float p = __expf(row[v] - best) / tot; en -= p * logf(p + 1e-9f); }
    en = wave_sum(en);
    if (lane == 0) s_red[wv] = en;
    __syncthreads();
    if (tid == 0) {
        targets[r] = bi;
        atomicAdd(&hist[bi], 1);       // code usage counts (integer: order-independent)
        rmax[r] = best;
        rsum[r] = tot;
        rent[r] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    }
}

// avg_probs[v] = mean_r softmax(x[r,:])[v].  The rows are split over blockIdx.y (a column per thread alone leaves 193 blocks walking 2048
// rows each); the per-split sums go to colpart [nsplit, V] and are added in a fixed order by vq_col_finish_kernel (deterministic).
__global__ __launch_bounds__(256) void vq_col_kernel(const float* __restrict__ x, const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                     float* __restrict__ colpart, int R, int V, int rows_per_split) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(R, r0 + rows_per_split);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += __expf(x[(int64_t)r * V + v] - rmax[r]) / rsum[r];
    colpart[(int64_t)blockIdx.y * V + v] = acc;
}

// partial[b] = sum over this block's v of avg*log(avg + 1e-7)
__global__ __launch_bounds__(256) void vq_col_finish_kernel(const float* __restrict__ colpart, float* __restrict__ partial, int nsplit, int R, int V,
                                                            MaskIds mk) {
    __shared__ float s_red[4];
    const int v = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float term = 0.f;
    if (v < V && !is_masked(v, mk.id, mk.n)) {
        float acc = 0.f;
        for (int i = 0; i < nsplit; ++i) acc += colpart[(int64_t)i * V + v];
        const float avg = acc / (float)R;
        term = avg * logf(avg + 1e-7f);
    }
    term = wave_sum(term);
    if (lane == 0) s_red[wv] = term;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// stats[0] = code perplexity, stats[1] = prob perplexity; ent_per_t[k] = mean_b rent[b*K + k]
__global__ __launch_bounds__(256) void vq_final_kernel(const int64_t* __restrict__ targets, const int* __restrict__ hist, const float* __restrict__ rent,
                                                       const float* __restrict__ partial, int nparts, float* __restrict__ stats,
                                                       float* __restrict__ ent_per_t, int R, int K) {
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float acc = 0.f;
    for (int r = tid; r < R; r += 256) {
        const int cnt = hist[targets[r]];
        acc += logf((float)cnt / (float)R + 1e-7f);      // sum_v hp log(hp + eps) = (1/R) sum_r log(cnt_r/R + eps)
    }
    acc = wave_sum(acc);
    if (lane == 0) s_red[wv] = acc;
    __syncthreads();
    if (tid == 0) {
        stats[0] = __expf(-(s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)R);
        float p = 0.f;
        for (int i = 0; i < nparts; ++i) p += partial[i];
        stats[1] = __expf(-p);
    }
    const int Bn = R / K;
    for (int k = tid; k < K; k += 256) {
        float e = 0.f;
        for (int b = 0; b < Bn; ++b) e += rent[b * K + k];
        ent_per_t[k] = e / (float)Bn;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ out, int E) {
    const int r = blockIdx.x;
    const int64_t s = idx[r];
    for (int e = threadIdx.x; e < E; e += 256) out[(int64_t)r * E + e] = src[s * E + e];
}

}  // namespace

extern "C" int sc_kw_affine(const float* x, const float* scale, const float* shift, float* out, int64_t rows, int K, int D, void* stream) {
    const int64_t total = rows * D;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(kw_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out, total, K, D);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_cosine_workspace_bytes(int R, int V) { return ((int64_t)R + V) * 4; }

extern "C" int sc_cosine_scores(const float* a, const float* emb, void* workspace, float* out, int R, int V, int E, float eps, void* stream) {
    SC_CHECK_ARG(R > 0 && V > 0 && E > 0 && E % 4 == 0, "sc_cosine_scores: need E %% 4 == 0 (E=%d)", E);
    float* inv_a = (float*)workspace;
    float* inv_e = inv_a + R;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((R + 3) / 4), dim3(256), 0, s, a, inv_a, (int64_t)R, E, eps);
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((V + 3) / 4), dim3(256), 0, s, emb, inv_e, (int64_t)V, E, eps);
    hipLaunchKernelGGL(cosine_tile_kernel, dim3((V + TS - 1) / TS, (R + TS - 1) / TS), dim3(256), 0, s, a, emb, inv_a, inv_e, out, R, V, E);
    SC_CHECK_LAUNCH();
    return 0;
}

static int vq_nsplit(int R) { return R >= 1024 ? 32 : (R >= 64 ? 8 : 1); }

extern "C" int sc_cosine_refine(float* scores, const float* a, const float* emb, int R, int V, int E, float delta, float eps, void* stream) {
    SC_CHECK_ARG(R >= 0 && V > 0 && E > 0 && delta >= 0.f, "sc_cosine_refine: bad arguments");
    if (R == 0) return 0;
    hipLaunchKernelGGL(cosine_refine_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, scores, a, emb, V, E, delta, eps);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_vq_workspace_bytes(int R, int V) { return ((int64_t)3 * R + (V + 255) / 256 + (int64_t)V * (1 + vq_nsplit(R))) * 4; }

extern "C" int sc_vq_fwd(const float* scores, int64_t* targets, float* stats2, float* ent_per_t, void* workspace, int R, int K, int V,
                         const int32_t* host_mask_ids, int n_mask, void* stream) {
    SC_CHECK_ARG(R > 0 && K > 0 && R % K == 0 && V > 0, "sc_vq_fwd: bad sizes R=%d K=%d V=%d", R, K, V);
    SC_CHECK_ARG(n_mask >= 0 && n_mask <= 8, "sc_vq_fwd: at most 8 masked ids");
    MaskIds mk{};
    mk.n = n_mask;
    for (int i = 0; i < n_mask; ++i) mk.id[i] = host_mask_ids[i];
    float* rmax = (float*)workspace;
    float* rsum = rmax + R;
    float* rent = rsum + R;
    float* partial = rent + R;
    const int nparts = (V + 255) / 256, nsplit = vq_nsplit(R), rps = (R + nsplit - 1) / nsplit;
    int* hist = (int*)(partial + nparts);
    float* colpart = (float*)(hist + V);
    hipStream_t s = (hipStream_t)stream;
    SC_CHECK_ARG(hipMemsetAsync(hist, 0, (size_t)V * 4, s) == hipSuccess, "sc_vq_fwd: hipMemsetAsync failed");
    hipLaunchKernelGGL(vq_row_kernel, dim3(R), dim3(256), 0, s, scores, targets, rmax, rsum, rent, hist, V, mk);
    hipLaunchKernelGGL(vq_col_kernel, dim3(nparts, nsplit), dim3(256), 0, s, scores, rmax, rsum, colpart, R, V, rps);
    hipLaunchKernelGGL(vq_col_finish_kernel, dim3(nparts), dim3(256), 0, s, colpart, partial, nsplit, R, V, mk);
    hipLaunchKernelGGL(vq_final_kernel, dim3(1), dim3(256), 0, s, targets, hist, rent, partial, nparts, stats2, ent_per_t, R, K);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_gather_rows(const float* src, const int64_t* idx, float* out, int R, int E, void* stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, src, idx, out, E);
    SC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ retrieval ranks (K15)
// mutualRetrieval (avssl/module/retrieval.py:45-121) sorts every score row and asks whether a candidate carrying the row's answer id
// sits among the first K.  Equivalent rank test, one pass over the row and no sort: rank_i = #{j : s_ij > best positive score of row i}
// (ties resolved towards the lower column index, i.e. a stable descending sort); hit@K  <=>  rank_i < K.  Rows without any positive
// candidate get rank = m.  One wave per row.
namespace {
__global__ __launch_bounds__(256) void retrieval_rank_kernel(const float* __restrict__ score, int64_t ld, const int64_t* __restrict__ own_ids,
                                                             const int64_t* __restrict__ cand_ids, int32_t* __restrict__ rank, int n, int m) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* s = score + (int64_t)row * ld;
    const int64_t id = own_ids[row];
    float best = -INFINITY;
    int best_j = m;
    for (int j = lane; j < m; j += 64)
        if (cand_ids[j] == id) {
            const float v = s[j];
            if (v > best || (v == best && j < best_j)) { best = v; best_j = j; }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(best_j, o, 64);
        if (ob > best || (ob == best && oj < best_j)) { best = ob; best_j = oj; }
    }
    int cnt = 0;
    if (best_j < m)
        for (int j = lane; j < m; j += 64) {
            const float v = s[j];
            cnt += (v > best || (v == best && j < best_j)) ? 1 : 0;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) rank[row] = best_j < m ? cnt : m;
}
}  // namespace

extern "C" int sc_retrieval_ranks(const float* score, int64_t ld, const int64_t* own_ids, const int64_t* cand_ids, int32_t* rank, int n, int m,
                                  void* stream) {
    SC_CHECK_ARG(n > 0 && m > 0 && ld >= m, "sc_retrieval_ranks: bad shape n=%d m=%d ld=%lld", n, m, (long long)ld);
    hipLaunchKernelGGL(retrieval_rank_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, score, ld, own_ids, cand_ids, rank, n, m);
    SC_CHECK_LAUNCH();
    return 0;
}
