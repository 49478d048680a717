// Backward kernels of the trainable tail: parallel branch (kwClip.py:1076-1108 under autograd), layer-mix
// weights (weighted_sum.py:26-45), L2 normalisation and the masked InfoNCE loss (losses.py:185-245), plus Adam / gradient clipping.
// Everything here is fp32.  The frame-level part keeps the forward's algebraic CLS pooling: with scores s_t = z_t.u_r + beta_r,
// p = softmax(s), p' = dropout(p), zbar_r = sum_t p'_t z_t, the gradients are
//     dp'_t = z_t . dzbar_r,   dp = dp' * keep / (1 - pd),   ds_t = p_t (dp_t - sum p dp),
//     dz_t  = sum_r p'_rt dzbar_r + ds_rt u_r,      du_r = sum_t ds_rt z_t,      dbeta_r = 0,
// so the backward is two streaming passes over the frames and never forms K, V or their gradients.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

// (hash32 / keep_elem / drop_thresh: common.h)

// ------------------------------------------------------------------------------------------------ CLS pooling, training forward
// as cls_pool_kernel (attention.hip) but fp32 outputs, the probabilities are kept for the backward and attention dropout is applied.
// x: bf16 [B*T, ld_x]; cls_tok fp32 [NQ, D]; scores fp32 [B*T, R]; cls_scores fp32 [NQ, R];
// p_out fp32 [B, R, NQ+T] (softmax, zero beyond the valid keys), xbar fp32 [B, R, D]
template <int DCH>
__global__ __launch_bounds__(256) void cls_pool_train_fwd_kernel(const bf16_t* __restrict__ x, int64_t ld_x, const float* __restrict__ cls_tok,
                                                                 const float* __restrict__ scores, const float* __restrict__ cls_scores,
                                                                 const int32_t* __restrict__ lens, float* __restrict__ p_out, float* __restrict__ xbar,
                                                                 int T, int NQ, int R, int D, uint32_t seed, uint32_t thresh, float keep_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;                 // [8][NQ + T]  probabilities (after dropout for the pooling)
    float* red = sc + 8 * (NQ + T);           // [4 waves][8][D]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const int nkeys = NQ + len, skeys = NQ + T;
    for (int i = tid; i < nkeys * R; i += 256) {
        const int kk = i / R, r = i - kk * R;
        sc[r * skeys + kk] = kk < NQ ? cls_scores[kk * R + r] : scores[((int64_t)b * T + (kk - NQ)) * R + r];
    }
    __syncthreads();
    for (int r = wave; r < R; r += 4) {
        float* row = sc + r * skeys;
        float* prow = p_out + ((int64_t)b * R + r) * skeys;
        float mx = -INFINITY;
        for (int kk = lane; kk < nkeys; kk += 64) mx = fmaxf(mx, row[kk]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int kk = lane; kk < nkeys; kk += 64) { const float e = expf(row[kk] - mx); row[kk] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int kk = lane; kk < skeys; kk += 64) {
            float pv = kk < nkeys ? row[kk] * inv : 0.f;
            prow[kk] = pv;
            if (thresh && kk < nkeys) pv = keep_elem(seed, (uint32_t)(((int64_t)b * R + r) * skeys + kk), thresh) ? pv * keep_scale : 0.f;
            if (kk < nkeys) row[kk] = pv;
        }
    }
    __syncthreads();
    float acc[8][DCH][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < DCH; ++c) acc[r][c][0] = acc[r][c][1] = acc[r][c][2] = acc[r][c][3] = 0.f;
    for (int kk = wave; kk < nkeys; kk += 4) {
        float zv[DCH][4];
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
            const int e = c * 256 + lane * 4;
            if (e < D) {
                if (kk < NQ) { const f32x4_t t = *(const f32x4_t*)(cls_tok + (int64_t)kk * D + e); zv[c][0] = t[0]; zv[c][1] = t[1]; zv[c][2] = t[2]; zv[c][3] = t[3]; }
                else { const uint2 t = *(const uint2*)(x + ((int64_t)b * T + (kk - NQ)) * ld_x + e); zv[c][0] = lo2f(t.x); zv[c][1] = hi2f(t.x); zv[c][2] = lo2f(t.y); zv[c][3] = hi2f(t.y); }
            } else { zv[c][0] = zv[c][1] = zv[c][2] = zv[c][3] = 0.f; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r >= R) break;
            const float pw = sc[r * skeys + kk];
#pragma unroll
            for (int c = 0; c < DCH; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[r][c][k] += pw * zv[c][k];
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r >= R) break;
#pragma unroll
        for (int c = 0; c < DCH; ++c)
            if (c * 256 + lane * 4 < D)
                *(f32x4_t*)(red + ((wave * 8 + r) * D) + c * 256 + lane * 4) = (f32x4_t){acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]};
    }
    __syncthreads();
    for (int i = tid; i < R * D; i += 256) {
        const int r = i / D, d = i - r * D;
        float s0 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s0 += red[(w * 8 + r) * D + d];
        xbar[((int64_t)b * R + r) * D + d] = s0;
    }
}

// ------------------------------------------------------------------------------------------------ CLS pooling backward, pass A
// dp'_rt = z_t . dzbar_r for every valid key, then (per row r) dropout + softmax backward.  Outputs ds [B,R,NQ+T] and the dropped
// probabilities p' [B,R,NQ+T] that pass B needs (zero beyond the valid keys).
template <int DCH>
__global__ __launch_bounds__(256) void cls_pool_bwd_scores_kernel(const bf16_t* __restrict__ x, int64_t ld_x, const float* __restrict__ cls_tok,
                                                                  const float* __restrict__ p, const float* __restrict__ dzbar,
                                                                  const int32_t* __restrict__ lens, float* __restrict__ ds_out, float* __restrict__ pp_out,
                                                                  int T, int NQ, int R, int D, uint32_t seed, uint32_t thresh, float keep_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dp = (float*)smem;                 // [8][NQ + T]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const int nkeys = NQ + len, skeys = NQ + T;
    float g[8][DCH][4];                       // dzbar_r slices of this lane
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
            const int e = c * 256 + lane * 4;
            f32x4_t t = {0.f, 0.f, 0.f, 0.f};
            if (r < R && e < D) t = *(const f32x4_t*)(dzbar + ((int64_t)b * R + r) * D + e);
            g[r][c][0] = t[0]; g[r][c][1] = t[1]; g[r][c][2] = t[2]; g[r][c][3] = t[3];
        }
    for (int kk = wave; kk < nkeys; kk += 4) {
        float zv[DCH][4];
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
            const int e = c * 256 + lane * 4;
            if (e < D) {
                if (kk < NQ) { const f32x4_t t = *(const f32x4_t*)(cls_tok + (int64_t)kk * D + e); zv[c][0] = t[0]; zv[c][1] = t[1]; zv[c][2] = t[2]; zv[c][3] = t[3]; }
                else { const uint2 t = *(const uint2*)(x + ((int64_t)b * T + (kk - NQ)) * ld_x + e); zv[c][0] = lo2f(t.x); zv[c][1] = hi2f(t.x); zv[c][2] = lo2f(t.y); zv[c][3] = hi2f(t.y); }
            } else { zv[c][0] = zv[c][1] = zv[c][2] = zv[c][3] = 0.f; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r >= R) break;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < DCH; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) d = fmaf(zv[c][k], g[r][c][k], d);
            d = wave_sum(d);
            if (lane == 0) dp[r * skeys + kk] = d;
        }
    }
    __syncthreads();
    for (int r = wave; r < R; r += 4) {
        const float* prow = p + ((int64_t)b * R + r) * skeys;
        float* dsr = ds_out + ((int64_t)b * R + r) * skeys;
        float* ppr = pp_out + ((int64_t)b * R + r) * skeys;
        float c = 0.f;
        for (int kk = lane; kk < nkeys; kk += 64) {
            float m = 1.f;
            if (thresh) m = keep_elem(seed, (uint32_t)(((int64_t)b * R + r) * skeys + kk), thresh) ? keep_scale : 0.f;
            const float dpk = dp[r * skeys + kk] * m;          // d loss / d p_t
            dp[r * skeys + kk] = dpk;
            c += prow[kk] * dpk;
        }
        c = wave_sum(c);
        for (int kk = lane; kk < skeys; kk += 64) {
            float dsv = 0.f, ppv = 0.f;
            if (kk < nkeys) {
                const float pv = prow[kk];
                dsv = pv * (dp[r * skeys + kk] - c);
                float m = 1.f;
                if (thresh) m = keep_elem(seed, (uint32_t)(((int64_t)b * R + r) * skeys + kk), thresh) ? keep_scale : 0.f;
                ppv = pv * m;
            }
            dsr[kk] = dsv;
            ppr[kk] = ppv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ CLS pooling backward, pass B
// One streaming pass over the frames of utterance b and the n hidden layers they were mixed from:
//   dz_t = sum_r p'_rt dzbar_r + ds_rt u_r   (kept in registers, never stored)
//   du_r        = sum_t ds_rt z_t                       -> du [B*S, R, D]
//   dcls_key    = dz of the NQ CLS keys                  -> dcls_key [B*S, NQ, D]
//   dalpha_n    = sum_t dz_t . H_n[b,t]  (H_n layer-normalised without affine when `normalize`)   -> dalpha [B*S, n]
// Grid (B, S): the keys of an utterance are split over S blocks, each writing its own partial row -- every consumer sums these
// outputs over b anyway (colsum / sc_mix_softmax_bwd), so the split costs nothing and gives the memory system 4 x S waves per
// utterance instead of 4.  Per-wave partials meet in ONE [8][D] LDS buffer through ds_add (24 KiB: several blocks per CU).
template <int DCH, int NL>   // NL: compile-time bound of n_layers (16 or 32): per-layer partial sums stay in registers, loads of several layers overlap
__global__ __launch_bounds__(512) void cls_pool_bwd_frames_kernel(const bf16_t* __restrict__ x, int64_t ld_x, const float* __restrict__ cls_tok,
                                                                  const void* __restrict__ hidden, int hidden_f32, int64_t layer_stride, int n_layers, int normalize, float eps,
                                                                  const float* __restrict__ pp, const float* __restrict__ ds, const float* __restrict__ dzbar,
                                                                  const float* __restrict__ u, const int32_t* __restrict__ lens,
                                                                  float* __restrict__ du, float* __restrict__ dcls_key, float* __restrict__ dalpha,
                                                                  int T, int NQ, int R, int D) {
    // dzbar_r and u_r of the block's utterance live in LDS (they are the same for all 8 waves, and 16 x 12 floats per lane in registers
    // would leave one wave per SIMD -- no memory-level parallelism); only the du accumulators stay in registers.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sg = (float*)smem;                 // [8][D] dzbar_r
    float* su = sg + 8 * D;                   // [8][D] u_r
    float* red = su + 8 * D;                  // [8][D] du accumulators (ds_add from the 8 waves)
    float* sda = red + 8 * D;                 // [32]   dalpha
    const int NW = 8;
    const int b = blockIdx.x, S = gridDim.y, sp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t orow = (int64_t)b * S + sp;  // output row of this block
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const int nkeys = NQ + len, skeys = NQ + T;
    for (int i = tid; i < 8 * D; i += 512) {
        const int r = i / D, d = i - r * D;
        sg[i] = r < R ? dzbar[((int64_t)b * R + r) * D + d] : 0.f;
        su[i] = r < R ? u[(int64_t)r * D + d] : 0.f;
        red[i] = 0.f;
    }
    if (tid < 32) sda[tid] = 0.f;
    for (int i = tid; i < NQ * D; i += 512) dcls_key[orow * NQ * D + i] = 0.f;     // rows of CLS keys this block does not own stay zero
    float dua[8][DCH][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < DCH; ++c) dua[r][c][0] = dua[r][c][1] = dua[r][c][2] = dua[r][c][3] = 0.f;
    float da[NL];
#pragma unroll
    for (int n = 0; n < NL; ++n) da[n] = 0.f;
    __syncthreads();
    for (int kk = sp * NW + wave; kk < nkeys; kk += NW * S) {
        float zv[DCH][4], dz[DCH][4];
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
            const int e = c * 256 + lane * 4;
            if (e < D) {
                if (kk < NQ) { const f32x4_t t = *(const f32x4_t*)(cls_tok + (int64_t)kk * D + e); zv[c][0] = t[0]; zv[c][1] = t[1]; zv[c][2] = t[2]; zv[c][3] = t[3]; }
                else { const uint2 t = *(const uint2*)(x + ((int64_t)b * T + (kk - NQ)) * ld_x + e); zv[c][0] = lo2f(t.x); zv[c][1] = hi2f(t.x); zv[c][2] = lo2f(t.y); zv[c][3] = hi2f(t.y); }
            } else { zv[c][0] = zv[c][1] = zv[c][2] = zv[c][3] = 0.f; }
            dz[c][0] = dz[c][1] = dz[c][2] = dz[c][3] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r >= R) break;
            const float ppv = pp[((int64_t)b * R + r) * skeys + kk], dsv = ds[((int64_t)b * R + r) * skeys + kk];
#pragma unroll
            for (int c = 0; c < DCH; ++c) {
                const int e = c * 256 + lane * 4;
                if (e < D) {
                    const f32x4_t g4 = *(const f32x4_t*)(sg + r * D + e), u4 = *(const f32x4_t*)(su + r * D + e);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dz[c][k] = fmaf(ppv, g4[k], fmaf(dsv, u4[k], dz[c][k]));
                        dua[r][c][k] = fmaf(dsv, zv[c][k], dua[r][c][k]);
                    }
                }
            }
        }
        if (kk < NQ) {
#pragma unroll
            for (int c = 0; c < DCH; ++c)
                if (c * 256 + lane * 4 < D)
                    *(f32x4_t*)(dcls_key + (orow * NQ + kk) * D + c * 256 + lane * 4) = (f32x4_t){dz[c][0], dz[c][1], dz[c][2], dz[c][3]};
        } else if (hidden) {
            const int64_t row = (int64_t)b * T + (kk - NQ);
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                if (n >= n_layers) break;
                float hv[DCH][4];
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < DCH; ++c) {
                    const int e = c * 256 + lane * 4;
                    if (e < D) {
                        const int64_t off = (int64_t)n * layer_stride + row * D + e;
                        if (hidden_f32) {
                            const f32x4_t t = *(const f32x4_t*)((const float*)hidden + off);
                            hv[c][0] = t[0]; hv[c][1] = t[1]; hv[c][2] = t[2]; hv[c][3] = t[3];
                        } else {
                            const uint2 t = *(const uint2*)((const bf16_t*)hidden + off);
                            hv[c][0] = lo2f(t.x); hv[c][1] = hi2f(t.x); hv[c][2] = lo2f(t.y); hv[c][3] = hi2f(t.y);
                        }
                    } else { hv[c][0] = hv[c][1] = hv[c][2] = hv[c][3] = 0.f; }
                }
                if (normalize) {
                    float sm = 0.f;
#pragma unroll
                    for (int c = 0; c < DCH; ++c) sm += (hv[c][0] + hv[c][1]) + (hv[c][2] + hv[c][3]);
                    const float mean = wave_sum(sm) / (float)D;
                    float q = 0.f;
#pragma unroll
                    for (int c = 0; c < DCH; ++c)
                        if (c * 256 + lane * 4 < D)
#pragma unroll
                            for (int k = 0; k < 4; ++k) { const float d0 = hv[c][k] - mean; q += d0 * d0; }
                    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
                    for (int c = 0; c < DCH; ++c)
                        if (c * 256 + lane * 4 < D)
#pragma unroll
                            for (int k = 0; k < 4; ++k) s = fmaf(dz[c][k], (hv[c][k] - mean) * rstd, s);
                } else {
#pragma unroll
                    for (int c = 0; c < DCH; ++c)
#pragma unroll
                        for (int k = 0; k < 4; ++k) s = fmaf(dz[c][k], hv[c][k], s);
                }
                da[n] += s;
            }
        }
    }
    if (hidden) {
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            if (n >= n_layers) break;
            const float t = wave_sum(da[n]);
            if (lane == 0) atomicAdd(&sda[n], t);
        }
    }
    // the 8 waves add their du partials into the shared [8][D] buffer
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r >= R) break;
#pragma unroll
        for (int c = 0; c < DCH; ++c)
            if (c * 256 + lane * 4 < D)
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&red[r * D + c * 256 + lane * 4 + k], dua[r][c][k]);
    }
    __syncthreads();
    for (int i = tid; i < R * D; i += 512) du[orow * R * D + i] = red[i];
    if (hidden && tid < n_layers) dalpha[orow * n_layers + tid] = sda[tid];
}

// ------------------------------------------------------------------------------------------------ small row kernels
// LayerNorm backward, fp32.  Kernel 1 (wave per row): dx and the row statistics; kernel 2 (thread per column): dgamma, dbeta.
template <int MAXCH>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                          float* __restrict__ dx, float* __restrict__ stats, int rows, int D, float eps, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float xv[MAXCH][4], gv[MAXCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int e = c * 256 + lane * 4;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
        if (e < D) { a = *(const f32x4_t*)(x + (int64_t)row * D + e); d = *(const f32x4_t*)(dy + (int64_t)row * D + e); w = *(const f32x4_t*)(gamma + e); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { xv[c][k] = a[k]; gv[c][k] = d[k] * w[k]; s += a[k]; }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
        if (c * 256 + lane * 4 < D)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d0 = xv[c][k] - mean; q += d0 * d0; }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
        if (c * 256 + lane * 4 < D)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float xh = (xv[c][k] - mean) * rstd; sg += gv[c][k]; sgx += gv[c][k] * xh; }
    sg = wave_sum(sg) / (float)D;
    sgx = wave_sum(sgx) / (float)D;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int e = c * 256 + lane * 4;
        if (e < D) {
            f32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float xh = (xv[c][k] - mean) * rstd; o[k] = rstd * (gv[c][k] - sg - xh * sgx); }
            if (accumulate) o += *(const f32x4_t*)(dx + (int64_t)row * D + e);
            *(f32x4_t*)(dx + (int64_t)row * D + e) = o;
        }
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
__global__ void ln_bwd_cols_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float g = 0.f, bsum = 0.f;
    for (int r = 0; r < rows; ++r) {
        const float dyv = dy[(int64_t)r * D + d];
        g += dyv * (x[(int64_t)r * D + d] - stats[2 * r]) * stats[2 * r + 1];
        bsum += dyv;
    }
    dgamma[d] += g;
    dbeta[d] += bsum;
}

// dz = dh * gelu'(z) (exact erf form), in place on dh
__global__ void gelu_bwd_kernel(const float* __restrict__ z, float* __restrict__ dh, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = z[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    dh[i] *= cdf + v * pdf;
}
// y = gelu(z) exact, fp32 (training forward of the branch FFN)
__global__ void gelu_fwd_kernel(const float* __restrict__ z, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = z[i];
    y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
}
// out[c] (+)= sum_r x[r, c].  Grid (column blocks of 64, row chunks): tall inputs are reduced by several blocks per column block through
// atomics on a pre-zeroed / pre-existing `out`.
__global__ void colsum_kernel(const float* __restrict__ x, int64_t ld, int rows, int cols, float* __restrict__ out, int accumulate, int rows_per_block) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;                       // 4 row-interleaved partial sums per column
    __shared__ float part[4][64];
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + sub; r < r1; r += 4) s += x[(int64_t)r * ld + c];
    part[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < cols) {
        s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (gridDim.y > 1) atomicAdd(out + c, s);
        else out[c] = accumulate ? out[c] + s : s;
    }
}
// y = x / |x|:  dx = (dy - y (y . dy)) / |x|
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float ss = 0.f, dot = 0.f;
    for (int d = lane; d < D; d += 64) { const float a = x[(int64_t)row * D + d]; ss += a * a; dot += a * dy[(int64_t)row * D + d]; }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float inv = rsqrtf(ss);
    for (int d = lane; d < D; d += 64) {
        const float a = x[(int64_t)row * D + d];
        dx[(int64_t)row * D + d] = inv * (dy[(int64_t)row * D + d] - a * dot * inv * inv);
    }
}
// elementwise dropout with the hash RNG: y = x * keep / (1 - p)  (in place allowed)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint32_t seed, uint32_t thresh, float keep_scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = keep_elem(seed, (uint32_t)i, thresh) ? x[i] * keep_scale : 0.f;
}
// out[r, c] = alpha * a[r, c] + b[r % b_rows, c]   (residual adds of the branch: b_rows = rows, or 1 for the broadcast CLS token)
__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rows, int cols, int b_rows, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    out[i] = fmaf(alpha, a[i], b[(int64_t)(r % b_rows) * cols + c]);
}
// layer-mix weights: alpha = softmax(w);  dw_n = alpha_n (dalpha_n - sum alpha dalpha) with dalpha = column sums of dalpha_b [B, n]
__global__ void mix_softmax_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dalpha_b, int B, int n, float* __restrict__ dw) {
    __shared__ float da[64], al[64];
    const int i = threadIdx.x;
    if (i < n) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dalpha_b[(int64_t)b * n + i];
        da[i] = s;
    }
    __syncthreads();
    if (i == 0) {
        float mx = -INFINITY;
        for (int k = 0; k < n; ++k) mx = fmaxf(mx, w[k]);
        float den = 0.f;
        for (int k = 0; k < n; ++k) { al[k] = expf(w[k] - mx); den += al[k]; }
        float c = 0.f;
        for (int k = 0; k < n; ++k) { al[k] /= den; c += al[k] * da[k]; }
        for (int k = 0; k < n; ++k) dw[k] += al[k] * (da[k] - c);
    }
}

// ------------------------------------------------------------------------------------------------ masked InfoNCE backward
// G = d loss / d logits  [Bg, Bg]  and  dinv = sum_ij G_ij (a_i . b_j)  (gradient of the logit scale), from the forward's row / column
// partial sums (workspace of sc_infonce_fwd).  dA = inv_t * G . B is then one sc_sgemm.
constexpr int TSN = 64, KCN = 16, PADN = 4;
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const int64_t* __restrict__ ids,
                                                          const float* __restrict__ prow, const float* __restrict__ pcol, float* __restrict__ G,
                                                          float* __restrict__ dinv_part, int Bg, int E, float inv_t, float margin, int dcl, int ntiles,
                                                          float wa, float wb) {
    __shared__ float sA[KCN][TSN + PADN], sB[KCN][TSN + PADN];
    __shared__ float sR[TSN], sC[TSN], red[4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * TSN, n0 = blockIdx.x * TSN;
    if (tid < TSN) {
        float r = 0.f, c = 0.f;
        for (int t = 0; t < ntiles; ++t) {
            if (m0 + tid < Bg) r += prow[(int64_t)t * Bg + m0 + tid];
            if (n0 + tid < Bg) c += pcol[(int64_t)t * Bg + n0 + tid];
        }
        sR[tid] = r; sC[tid] = c;
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    for (int k0 = 0; k0 < E; k0 += KCN) {
        f32x4_t va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        if (m0 + lr < Bg && k0 + lk < E) va = *(const f32x4_t*)(a + (int64_t)(m0 + lr) * E + k0 + lk);
        if (n0 + lr < Bg && k0 + lk < E) vb = *(const f32x4_t*)(b + (int64_t)(n0 + lr) * E + k0 + lk);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { sA[lk + i][lr] = va[i]; sB[lk + i][lr] = vb[i]; }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KCN; ++kk) {
            const f32x4_t a4 = *(const f32x4_t*)&sA[kk][ty * 4];
            const f32x4_t b4 = *(const f32x4_t*)&sB[kk][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
    }
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = m0 + ty * 4 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = n0 + tx * 4 + j;
            if (gi < Bg && gj < Bg) {
                const float sdot = acc[i][j];
                float l = sdot * inv_t;
                const bool diag = gi == gj;
                if (diag && margin > 0.f) l -= margin;
                bool neg = ids ? (ids[gi] != ids[gj]) : !diag;
                if (!dcl && diag) neg = true;
                const float e = neg ? __expf(l) : 0.f;
                float gv = wa * (e / sR[ty * 4 + i]) + wb * (e / sC[tx * 4 + j]);
                if (diag) gv -= wa + wb;
                G[(int64_t)gi * Bg + gj] = gv;
                dsum += gv * sdot;
            }
        }
    }
    dsum = wave_sum(dsum);
    if ((tid & 63) == 0) red[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) dinv_part[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_small_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

// ------------------------------------------------------------------------------------------------ optimizer
// sum of squares of a flat fp32 buffer -> partial[blockIdx.x] (fp64 accumulation inside the block)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = g[i]; s += v * v; }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void finish_norm_kernel(const double* __restrict__ partial, int n, float max_norm, float* __restrict__ out2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += partial[i];
        const float norm = (float)sqrt(s);
        out2[0] = norm;                                                  // total gradient norm
        out2[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;   // torch.nn.utils.clip_grad_norm_ coefficient
    }
}
// torch.optim.Adam (L2 weight decay added to the gradient, bias-corrected), gradient pre-scaled by *clip_coef
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            const float* __restrict__ clip_coef, float lr, float beta1, float beta2, float eps, float weight_decay,
                            float bc1, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * (clip_coef ? clip_coef[0] : 1.0f);
    const float pi = p[i];
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

}  // namespace

// ================================================================================================ C ABI
#define POOL_DISPATCH(KERNEL, LDS, ...)                                                                                 \
    do {                                                                                                                \
        if (D <= 256) { (void)hipFuncSetAttribute((const void*)KERNEL<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); hipLaunchKernelGGL(KERNEL<1>, dim3(B), dim3(256), LDS, s, __VA_ARGS__); } \
        else if (D <= 512) { (void)hipFuncSetAttribute((const void*)KERNEL<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); hipLaunchKernelGGL(KERNEL<2>, dim3(B), dim3(256), LDS, s, __VA_ARGS__); } \
        else if (D <= 768) { (void)hipFuncSetAttribute((const void*)KERNEL<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); hipLaunchKernelGGL(KERNEL<3>, dim3(B), dim3(256), LDS, s, __VA_ARGS__); } \
        else { (void)hipFuncSetAttribute((const void*)KERNEL<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); hipLaunchKernelGGL(KERNEL<4>, dim3(B), dim3(256), LDS, s, __VA_ARGS__); } \
    } while (0)

static int pool_args_ok(int T, int NQ, int R, int D, int64_t ld_x, const char* who) {
    SC_CHECK_ARG(NQ >= 1 && R >= NQ && R <= 8, "%s: need 1 <= NQ <= R <= 8 (NQ=%d R=%d)", who, NQ, R);
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0 && ld_x % 4 == 0, "%s: D=%d must be a multiple of 4, <= 1024", who, D);
    SC_CHECK_ARG((8 * (NQ + T) + 4 * 8 * D) * 4 <= 160 * 1024, "%s: T=%d / D=%d too large for LDS", who, T, D);
    return 0;
}

extern "C" int sc_cls_pool_train_fwd(const void* x, int64_t ld_x, const float* cls_tok, const float* scores, const float* cls_scores,
                                     const int32_t* lens, float* p_out, float* xbar, int B, int T, int NQ, int R, int D,
                                     float drop_p, uint32_t seed, void* stream) {
    if (pool_args_ok(T, NQ, R, D, ld_x, "sc_cls_pool_train_fwd")) return -1;
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_cls_pool_train_fwd: drop_p=%f", drop_p);
    if (B <= 0) return 0;
    const int lds = (8 * (NQ + T) + 4 * 8 * D) * 4;
    hipStream_t s = (hipStream_t)stream;
    POOL_DISPATCH(cls_pool_train_fwd_kernel, lds, (const bf16_t*)x, ld_x, cls_tok, scores, cls_scores, lens, p_out, xbar, T, NQ, R, D, seed,
                  drop_thresh(drop_p), 1.0f / (1.0f - drop_p));
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_cls_pool_bwd(const void* x, int64_t ld_x, const float* cls_tok, const void* hidden, int hidden_f32, int64_t layer_stride,
                               int n_layers, int normalize, const float* p, const float* dzbar, const float* u, const int32_t* lens, float* ds_ws,
                               float* pp_ws, float* du, float* dcls_key, float* dalpha, int B, int T, int NQ, int R, int D, int nsplit,
                               float drop_p, uint32_t seed, void* stream) {
    if (pool_args_ok(T, NQ, R, D, ld_x, "sc_cls_pool_bwd")) return -1;
    SC_CHECK_ARG(n_layers >= 0 && n_layers <= 32, "sc_cls_pool_bwd: n_layers=%d (max 32)", n_layers);
    SC_CHECK_ARG(nsplit >= 1 && nsplit <= 64, "sc_cls_pool_bwd: nsplit=%d (1..64)", nsplit);
    if (B <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int lds_a = 8 * (NQ + T) * 4;
    POOL_DISPATCH(cls_pool_bwd_scores_kernel, lds_a, (const bf16_t*)x, ld_x, cls_tok, p, dzbar, lens, ds_ws, pp_ws, T, NQ, R, D, seed,
                  drop_thresh(drop_p), 1.0f / (1.0f - drop_p));
    SC_CHECK_LAUNCH();
    const int lds_b = (3 * 8 * D + 32) * 4;
    const dim3 grid_b(B, nsplit);
#define FRAMES_LAUNCH(DCH_) do { if (n_layers <= 16) FRAMES_LAUNCH2(DCH_, 16); else FRAMES_LAUNCH2(DCH_, 32); } while (0)
#define FRAMES_LAUNCH2(DCH_, NL_) do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)cls_pool_bwd_frames_kernel<DCH_, NL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);                   \
    hipLaunchKernelGGL((cls_pool_bwd_frames_kernel<DCH_, NL_>), grid_b, dim3(512), lds_b, s, (const bf16_t*)x, ld_x, cls_tok, (const void*)(n_layers ? hidden : nullptr), \
                       hidden_f32, layer_stride, n_layers, normalize, 1e-5f, pp_ws, ds_ws, dzbar, u, lens, du, dcls_key, dalpha, T, NQ, R, D); } while (0)
    if (D <= 256) FRAMES_LAUNCH(1);
    else if (D <= 512) FRAMES_LAUNCH(2);
    else if (D <= 768) FRAMES_LAUNCH(3);
    else FRAMES_LAUNCH(4);
#undef FRAMES_LAUNCH
#undef FRAMES_LAUNCH2
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, float* stats_ws,
                                int rows, int D, float eps, int accumulate_dx, void* stream) {
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0, "sc_layernorm_bwd: D=%d must be a multiple of 4, <= 1024", D);
    if (rows <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ln_bwd_rows_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, s, x, dy, gamma, dx, stats_ws, rows, D, eps, accumulate_dx);
    SC_CHECK_LAUNCH();
    if (dgamma && dbeta) {
        hipLaunchKernelGGL(ln_bwd_cols_kernel, dim3((D + 63) / 64), dim3(64), 0, s, x, dy, stats_ws, dgamma, dbeta, rows, D);
        SC_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sc_gelu_f32(const float* z, float* y_or_dh, int64_t n, int backward, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (backward) hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, y_or_dh, n);
    else hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, y_or_dh, n);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_colsum(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int cb = (cols + 63) / 64;
    int chunks = 1;
    if (rows >= 256 && cb < 512) chunks = min((rows + 63) / 64, max(1, 1024 / cb));
    const int rpb = (rows + chunks - 1) / chunks;
    chunks = (rows + rpb - 1) / rpb;
    if (chunks > 1 && !accumulate) { if (hipMemsetAsync(out, 0, (size_t)cols * 4, s) != hipSuccess) { sc_set_error("sc_colsum: memset failed"); return -2; } }
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, chunks), dim3(256), 0, s, x, ld, rows, cols, out, accumulate, rpb);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_l2norm_bwd(const float* x, const float* dy, float* dx, int rows, int D, void* stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, dx, rows, D);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_dropout_f32(const float* x, float* y, int64_t n, float drop_p, uint32_t seed, void* stream) {
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_dropout_f32: drop_p=%f", drop_p);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, seed, drop_thresh(drop_p),
                       1.0f / (1.0f - drop_p));
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_add_rows_f32(const float* a, const float* b, float* out, int rows, int cols, int b_rows, float alpha, void* stream) {
    SC_CHECK_ARG(rows > 0 && cols > 0 && b_rows > 0, "sc_add_rows_f32: bad shape");
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)(((int64_t)rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, rows, cols, b_rows, alpha);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_mix_softmax_bwd(const float* w, const float* dalpha_b, int B, int n, float* dw, void* stream) {
    SC_CHECK_ARG(n > 0 && n <= 64, "sc_mix_softmax_bwd: n=%d (max 64)", n);
    hipLaunchKernelGGL(mix_softmax_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, dalpha_b, B, n, dw);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_infonce_bwd_workspace_bytes(int Bg) {
    const int64_t nt = (Bg + TSN - 1) / TSN;
    return nt * nt * 4;
}
// fwd_workspace: the buffer sc_infonce_fwd filled for the same inputs.  G: [Bg, Bg] fp32 out.  dinv_out[0] = d loss / d inv_temperature.
extern "C" int sc_infonce_bwd(const float* feat_a, const float* feat_b, const int64_t* ids, const void* fwd_workspace, void* bwd_workspace,
                              float* G, float* dinv_out, int Bg, int E, float inv_temperature, float margin, int dcl, int a2b, int b2a,
                              void* stream) {
    SC_CHECK_ARG(Bg > 0 && E > 0 && E % 4 == 0, "sc_infonce_bwd: E=%d must be a positive multiple of 4", E);
    SC_CHECK_ARG(a2b || b2a, "sc_infonce_bwd: a2b and b2a cannot both be false");
    const int nt = (Bg + TSN - 1) / TSN;
    const float* prow = (const float*)fwd_workspace;
    const float* pcol = prow + (int64_t)nt * Bg;
    const float scale = (a2b && b2a ? 0.5f : 1.0f) / (float)Bg;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(infonce_bwd_kernel, dim3(nt, nt), dim3(256), 0, s, feat_a, feat_b, ids, prow, pcol, G, (float*)bwd_workspace, Bg, E,
                       inv_temperature, margin, dcl, nt, a2b ? scale : 0.f, b2a ? scale : 0.f);
    SC_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(256), 0, s, (const float*)bwd_workspace, nt * nt, dinv_out);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_grad_norm_workspace_bytes(void) { return 1024 * 8; }
// out2[0] = |g|_2 over the flat buffer, out2[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0)
extern "C" int sc_grad_norm(const float* g, int64_t n, float max_norm, void* workspace, float* out2, void* stream) {
    SC_CHECK_ARG(n > 0, "sc_grad_norm: empty buffer");
    const int blocks = (int)(n / 256 + 1 < 1024 ? n / 256 + 1 : 1024);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, g, n, (double*)workspace);
    SC_CHECK_LAUNCH();
    hipLaunchKernelGGL(finish_norm_kernel, dim3(1), dim3(1), 0, s, (const double*)workspace, blocks, max_norm, out2);
    SC_CHECK_LAUNCH();
    return 0;
}
extern "C" int sc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* clip_coef, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, void* stream) {
    SC_CHECK_ARG(n > 0 && step >= 1, "sc_adam_step: n=%lld step=%d", (long long)n, step);
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, clip_coef, lr, beta1, beta2,
                       eps, weight_decay, bc1, sqrtf(bc2));
    SC_CHECK_LAUNCH();
    return 0;
}
