// Row-wise HBM-bound kernels: LayerNorm (+GELU), weighted layer sum, L2 normalise, wave layer-norm.
// One 64-lane wave owns one row (D <= 1024); reductions are wave shuffles, loads are 8/16 B per lane.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

constexpr int MAXC = 4;  // D <= 1024: up to 4 chunks of 256 elements, 4 elements per lane per chunk

template <bool IN_F32>
__device__ __forceinline__ void load_row(const void* base, int64_t off, int D, int lane, float (&v)[MAXC][4]) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int e = c * 256 + lane * 4;
        if (e < D) {
            if (IN_F32) {
                f32x4_t t = *(const f32x4_t*)((const float*)base + off + e);
                v[c][0] = t[0]; v[c][1] = t[1]; v[c][2] = t[2]; v[c][3] = t[3];
            } else {
                uint2 t = *(const uint2*)((const bf16_t*)base + off + e);
                v[c][0] = lo2f(t.x); v[c][1] = hi2f(t.x); v[c][2] = lo2f(t.y); v[c][3] = hi2f(t.y);
            }
        } else {
            v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
        }
    }
}

__device__ __forceinline__ void row_stats(const float (&v)[MAXC][4], int D, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int e = c * 256 + lane * 4;
        if (e < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { float d = v[c][i] - mean; q += d * d; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

// out = [gelu]( (x - mean) * rstd * gamma + beta ).  Each wave owns RPW rows and issues all their loads up front
// (more bytes in flight per CU: a single 1.5 KB row per wave leaves the HBM pipe half empty).
constexpr int RPW = 2;
template <bool IN_F32, bool OUT_F32, bool OUT_F16 = false>     // OUT_F16: the 16-bit output is IEEE half (SC_LN_OUT_F16)
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, int64_t ld_in, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out, int64_t ld_out,
                                                        int64_t rows, int D, float eps, int gelu) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float v[RPW][MAXC][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
        load_row<IN_F32>(x, row * ld_in, D, lane, v[r]);
    }
    float g4[MAXC][4], b4[MAXC][4];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int e = c * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g4[c][i] = (gamma && e < D) ? gamma[e + i] : 1.f;
            b4[c][i] = (beta && e < D) ? beta[e + i] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        float mean, rstd;
        row_stats(v[r], D, lane, eps, mean, rstd);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            int e = c * 256 + lane * 4;
            if (e < D) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y = (v[r][c][i] - mean) * rstd * g4[c][i] + b4[c][i];
                    o[i] = gelu ? gelu_erf(y) : y;
                }
                if (OUT_F32) {
                    *(f32x4_t*)((float*)out + row * ld_out + e) = (f32x4_t){o[0], o[1], o[2], o[3]};
                } else {
                    uint2 p; p.x = pack2x<OUT_F16>(o[0], o[1]); p.y = pack2x<OUT_F16>(o[2], o[3]);
                    *(uint2*)((bf16_t*)out + row * ld_out + e) = p;
                }
            }
        }
    }
}

// Fast path for the hot shape (D = 768, bf16 -> bf16, affine): a wave owns TWO consecutive rows = 192 chunks of 16 B = exactly 3 chunks
// per lane (16-byte loads/stores; chunk c = lane + 64*i belongs to row c / 96).  4 x this per step is 25 LayerNorms over [128000, 768].
// DROPRES: the row that is normalised is residual + dropout(x) (train-mode frozen encoder: x = LN(x + dropout(branch)), one pass instead of two;
// same counter-based mask as sc_dropout_bf16 -- element index = row * 768 + column).
template <bool DROPRES>
__global__ __launch_bounds__(256) void layernorm768_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ out, int64_t rows, float eps, const bf16_t* __restrict__ residual,
                                                           uint32_t drop_seed, uint32_t drop_thresh_, float keep_scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;
    float v[3][8];
    int rsel[3], col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        rsel[i] = c >= 96;
        col[i] = (c - rsel[i] * 96) * 8;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (!rsel[i] || two) u = *(const uint4*)(x + (row0 + rsel[i]) * 768 + col[i]);
        v[i][0] = lo2f(u.x); v[i][1] = hi2f(u.x); v[i][2] = lo2f(u.y); v[i][3] = hi2f(u.y);
        v[i][4] = lo2f(u.z); v[i][5] = hi2f(u.z); v[i][6] = lo2f(u.w); v[i][7] = hi2f(u.w);
        if (DROPRES) {
            uint4 r = make_uint4(0, 0, 0, 0);
            if (!rsel[i] || two) r = *(const uint4*)(residual + (row0 + rsel[i]) * 768 + col[i]);
            const float rv[8] = {lo2f(r.x), hi2f(r.x), lo2f(r.y), hi2f(r.y), lo2f(r.z), hi2f(r.z), lo2f(r.w), hi2f(r.w)};
            const uint32_t e0 = (uint32_t)((row0 + rsel[i]) * 768 + col[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] = rv[k] + (keep_elem(drop_seed, e0 + k, drop_thresh_) ? v[i][k] * keep_scale : 0.f);
        }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[i][k];
        if (rsel[i]) s1 += t; else s0 += t;
    }
    const float m0 = wave_sum(s0) * (1.0f / 768.0f), m1 = wave_sum(s1) * (1.0f / 768.0f);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float mm = rsel[i] ? m1 : m0;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mm; t += d * d; }
        if (rsel[i]) q1 += t; else q0 += t;
    }
    const float r0 = rsqrtf(wave_sum(q0) * (1.0f / 768.0f) + eps), r1 = rsqrtf(wave_sum(q1) * (1.0f / 768.0f) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (rsel[i] && !two) continue;
        const float mm = rsel[i] ? m1 : m0, rr = rsel[i] ? r1 : r0;
        const f32x4_t g0 = *(const f32x4_t*)(gamma + col[i]), g1 = *(const f32x4_t*)(gamma + col[i] + 4);
        const f32x4_t b0 = *(const f32x4_t*)(beta + col[i]), b1 = *(const f32x4_t*)(beta + col[i] + 4);
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = (v[i][k] - mm) * rr * g0[k] + b0[k];
            o[4 + k] = (v[i][4 + k] - mm) * rr * g1[k] + b1[k];
        }
        uint4 u;
        u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]); u.z = pack2bf(o[4], o[5]); u.w = pack2bf(o[6], o[7]);
        *(uint4*)(out + (row0 + rsel[i]) * 768 + col[i]) = u;
    }
}

// Fast path for the pre-LN models' residual streams (HuBERT-large, ViT-L/14: fp32 [rows, 1024] -> bf16, affine; 99 launches per P-large step): a lane owns 8
// consecutive columns of each 512-column half (two 16-byte loads per half, ONE 16-byte store per half -- the generic kernel's 4-column ownership stores 8
// bytes per lane), a wave owns two consecutive rows and issues all eight loads up front.
template <bool OUT_F16>
__global__ __launch_bounds__(256) void layernorm1024f_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             bf16_t* __restrict__ out, int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;
    f32x4_t v[2][2][2];                                  // [row][half][quad]
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float* xr = x + (row0 + (two ? r : 0)) * 1024 + lane * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) { v[r][h][0] = *(const f32x4_t*)(xr + h * 512); v[r][h][1] = *(const f32x4_t*)(xr + h * 512 + 4); }
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) s += (v[r][h][q][0] + v[r][h][q][1]) + (v[r][h][q][2] + v[r][h][q][3]);
        mean[r] = wave_sum(s) * (1.0f / 1024.0f);
        float qq = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = v[r][h][q][i] - mean[r]; qq += d * d; }
        rstd[r] = rsqrtf(wave_sum(qq) * (1.0f / 1024.0f) + eps);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = h * 512 + lane * 8;
        const f32x4_t g0 = *(const f32x4_t*)(gamma + col), g1 = *(const f32x4_t*)(gamma + col + 4);
        const f32x4_t b0 = *(const f32x4_t*)(beta + col), b1 = *(const f32x4_t*)(beta + col + 4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r && !two) continue;
            const f32x4_t o0 = (v[r][h][0] - mean[r]) * rstd[r] * g0 + b0, o1 = (v[r][h][1] - mean[r]) * rstd[r] * g1 + b1;
            uint4 u;
            u.x = pack2x<OUT_F16>(o0[0], o0[1]); u.y = pack2x<OUT_F16>(o0[2], o0[3]); u.z = pack2x<OUT_F16>(o1[0], o1[1]); u.w = pack2x<OUT_F16>(o1[2], o1[3]);
            *(uint4*)(out + (row0 + r) * 1024 + col) = u;
        }
    }
}

// Fast path for the HuBERT-large feature extractor (extractor_mode = layer_norm: LayerNorm over the 512 channels + GELU after every conv,
// 8.3 GB per 64-utterance step): D = 512 bf16 -> bf16, a row is exactly one 16-byte chunk per lane; a wave owns four consecutive rows and
// issues their loads up front.
// fp32 rows of 768 (the pre-LN residual stream of CLIP ViT-B: ln_1 / ln_2 / ln_post read it, the GEMMs behind them take bf16): two rows per wave, 16-byte loads
// (lane owns columns q * 256 + lane * 4 .. + 3, q = 0..2), 8-byte bf16 stores.  The generic kernel spent ~1 450 instructions per row pair on this shape (round 6).
__global__ __launch_bounds__(256) void layernorm768f_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ out, int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;
    f32x4_t v[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float* xr = x + (row0 + (two ? r : 0)) * 768 + lane * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) v[r][q] = *(const f32x4_t*)(xr + q * 256);
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) s += (v[r][q][0] + v[r][q][1]) + (v[r][q][2] + v[r][q][3]);
        mean[r] = wave_sum(s) * (1.0f / 768.0f);
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[r][q][i] - mean[r]; qq += d * d; }
        rstd[r] = rsqrtf(wave_sum(qq) * (1.0f / 768.0f) + eps);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int col = q * 256 + lane * 4;
        const f32x4_t g4 = *(const f32x4_t*)(gamma + col), b4 = *(const f32x4_t*)(beta + col);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r && !two) continue;
            const f32x4_t o = (v[r][q] - mean[r]) * rstd[r] * g4 + b4;
            uint2 u;
            u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]);
            *(uint2*)(out + (row0 + r) * 768 + col) = u;
        }
    }
}

template <bool GELU>
__global__ __launch_bounds__(256) void layernorm512_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ out, int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= rows) return;
    uint4 u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
        u[r] = *(const uint4*)(x + row * 512 + lane * 8);
    }
    const f32x4_t g0 = *(const f32x4_t*)(gamma + lane * 8), g1 = *(const f32x4_t*)(gamma + lane * 8 + 4);
    const f32x4_t b0 = *(const f32x4_t*)(beta + lane * 8), b1 = *(const f32x4_t*)(beta + lane * 8 + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (row0 + r >= rows) break;
        float v[8] = {lo2f(u[r].x), hi2f(u[r].x), lo2f(u[r].y), hi2f(u[r].y), lo2f(u[r].z), hi2f(u[r].z), lo2f(u[r].w), hi2f(u[r].w)};
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
        const float mean = wave_sum(t) * (1.0f / 512.0f);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[k] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / 512.0f) + eps);
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = (v[k] - mean) * rstd * g0[k] + b0[k];
            o[4 + k] = (v[4 + k] - mean) * rstd * g1[k] + b1[k];
        }
        if (GELU) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const f32x2_t y = gelu_poly2((f32x2_t){o[k], o[k + 1]});
                o[k] = y[0]; o[k + 1] = y[1];
            }
        }
        uint4 w;
        w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]); w.z = pack2bf(o[4], o[5]); w.w = pack2bf(o[6], o[7]);
        *(uint4*)(out + (row0 + r) * 512 + lane * 8) = w;
    }
}

// out[m,:] = sum_i softmax(w)_i * (normalize ? LN_noaffine(h_i[m,:]) : h_i[m,:])
template <bool IN_F32>
__global__ __launch_bounds__(256) void weighted_sum_kernel(const void* __restrict__ hidden, int64_t layer_stride, const float* __restrict__ w,
                                                           void* __restrict__ out, int n, int64_t rows, int D, int normalize,
                                                           float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) mx = fmaxf(mx, w[i]);
    float den = 0.f;
    for (int i = 0; i < n; ++i) den += __expf(w[i] - mx);
    float acc[MAXC][4];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
    for (int i = 0; i < n; ++i) {
        const float wi = __expf(w[i] - mx) / den;
        float v[MAXC][4];
        load_row<IN_F32>(hidden, (int64_t)i * layer_stride + row * D, D, lane, v);
        if (normalize) {
            float mean, rstd;
            row_stats(v, D, lane, eps, mean, rstd);
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c][k] += wi * ((v[c][k] - mean) * rstd);
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c][k] += wi * v[c][k];
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int e = c * 256 + lane * 4;
        if (e < D) {
            uint2 p; p.x = pack2bf(acc[c][0], acc[c][1]); p.y = pack2bf(acc[c][2], acc[c][3]);
            *(uint2*)((bf16_t*)out + row * D + e) = p;
        }
    }
}

template <bool IN_F32>
__global__ __launch_bounds__(256) void l2norm_kernel(const void* __restrict__ x, int64_t ld_in, float* __restrict__ out, int64_t rows, int D, float norm_floor) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[MAXC][4];
    load_row<IN_F32>(x, row * ld_in, D, lane, v);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) q += v[c][i] * v[c][i];
    // norm_floor = 0: kwClip.py:1436 divides by the plain norm (no eps); 1e-8: the clamp of F.cosine_similarity's operands (kwClip.py:889-897)
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), norm_floor);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        int e = c * 256 + lane * 4;
        if (e < D) *(f32x4_t*)(out + row * D + e) = (f32x4_t){v[c][0] * inv, v[c][1] * inv, v[c][2] * inv, v[c][3] * inv};
    }
}

// per-utterance F.layer_norm(wav[:len], (len,)) in place over the unpadded samples (eps 1e-5); pad stays 0.
// (round 6: one block per utterance and three dword passes left 192 of 256 CUs idle in the first, un-overlapped kernel of a normalize = True model's step: 0.27 ms for
//  41 MB at B = 64.  Now WSEG blocks per utterance; each computes the utterance's statistics itself -- one pass, 16-byte loads, sum and sum of squares in fp64, served
//  from L2 for all but the first block -- and normalises its own segment.)
constexpr int WSEG = 8;
__global__ __launch_bounds__(1024) void wave_layernorm_kernel(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ lens,
                                                              int64_t ld, float eps) {
    __shared__ double red[2][16];
    __shared__ float stat[2];
    const int b = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int len = lens[b];
    const float* x = in + (int64_t)b * ld;
    float* y = out + (int64_t)b * ld;
    double s = 0.0, q = 0.0;
    const bool vec = ((ld & 3) == 0) && ((((uintptr_t)in) & 15) == 0);
    const int len4 = vec ? (len & ~3) : 0;
    for (int i = tid * 4; i < len4; i += 4096) {
        const f32x4_t v = *(const f32x4_t*)(x + i);
        s += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
        q += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
    }
    for (int i = len4 + tid; i < len; i += 1024) { const double v = (double)x[i]; s += v; q += v * v; }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane == 0) { red[0][wv] = s; red[1][wv] = q; }
    __syncthreads();
    if (tid == 0) {
        double t = 0, u = 0;
        for (int i = 0; i < 16; ++i) { t += red[0][i]; u += red[1][i]; }
        const double mean = t / len;
        double var = u / len - mean * mean;
        var = var > 0 ? var : 0;
        stat[0] = (float)mean; stat[1] = rsqrtf((float)var + eps);
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const int64_t per = ((ld + WSEG - 1) / WSEG + 3) & ~(int64_t)3;
    const int64_t lo = (int64_t)seg * per, hi = lo + per < ld ? lo + per : ld;
    if (vec && (((uintptr_t)out) & 15) == 0) {
        for (int64_t i = lo + tid * 4; i < hi; i += 4096) {       // (ld % 4 == 0 and lo % 4 == 0: whole quads)
            f32x4_t v = *(const f32x4_t*)(x + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (i + k) < len ? (v[k] - mean) * rstd : 0.f;
            *(f32x4_t*)(y + i) = v;
        }
    } else {
        for (int64_t i = lo + tid; i < hi; i += 1024) y[i] = i < len ? (x[i] - mean) * rstd : 0.f;
    }
}

}  // namespace

extern "C" int sc_layernorm(const void* x, int64_t ld_in, const float* gamma, const float* beta, void* out, int64_t ld_out,
                            int64_t rows, int D, float eps, int flags, void* stream) {
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0, "sc_layernorm: D=%d must be a multiple of 4, <= 1024", D);
    SC_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "sc_layernorm: gamma and beta must both be given or both null");
    SC_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0, "sc_layernorm: leading dims must be multiples of 4");
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 4 * RPW - 1) / (4 * RPW))), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int gelu = (flags & SC_LN_GELU) ? 1 : 0;
    const bool out16h = flags & SC_LN_OUT_F16;
    SC_CHECK_ARG(!out16h || ((flags & SC_LN_IN_F32) && !(flags & SC_LN_OUT_F32)), "sc_layernorm: SC_LN_OUT_F16 goes with SC_LN_IN_F32 and a 16-bit output (the pre-LN residual stream)");
    if (D == 768 && flags == 0 && gamma && ld_in == 768 && ld_out == 768 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL((layernorm768_kernel<false>), dim3((unsigned)((rows + 7) / 8)), block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, eps,
                           (const bf16_t*)nullptr, 0u, 0u, 1.0f);
        SC_CHECK_LAUNCH();
        return 0;
    }
    if (D == 512 && (flags & ~SC_LN_GELU) == 0 && gamma && ld_in == 512 && ld_out == 512 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        const dim3 g512((unsigned)((rows + 15) / 16));
        if (gelu) hipLaunchKernelGGL((layernorm512_kernel<true>), g512, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, eps);
        else hipLaunchKernelGGL((layernorm512_kernel<false>), g512, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, eps);
        SC_CHECK_LAUNCH();
        return 0;
    }
    const bool in32 = flags & SC_LN_IN_F32, out32 = flags & SC_LN_OUT_F32;
    if (D == 1024 && in32 && !out32 && !gelu && gamma && ld_in == 1024 && ld_out == 1024 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        if (out16h) hipLaunchKernelGGL(layernorm1024f_kernel<true>, dim3((unsigned)((rows + 7) / 8)), block, 0, s, (const float*)x, gamma, beta, (bf16_t*)out, rows, eps);
        else hipLaunchKernelGGL(layernorm1024f_kernel<false>, dim3((unsigned)((rows + 7) / 8)), block, 0, s, (const float*)x, gamma, beta, (bf16_t*)out, rows, eps);
        SC_CHECK_LAUNCH();
        return 0;
    }
    if (D == 768 && in32 && !out32 && !out16h && !gelu && gamma && ld_in == 768 && ld_out == 768 && (((uintptr_t)x & 15) | ((uintptr_t)out & 7)) == 0) {
        hipLaunchKernelGGL(layernorm768f_kernel, dim3((unsigned)((rows + 7) / 8)), block, 0, s, (const float*)x, gamma, beta, (bf16_t*)out, rows, eps);
        SC_CHECK_LAUNCH();
        return 0;
    }
    if (in32 && out32) hipLaunchKernelGGL((layernorm_kernel<true, true>), grid, block, 0, s, x, ld_in, gamma, beta, out, ld_out, rows, D, eps, gelu);
    else if (in32 && out16h) hipLaunchKernelGGL((layernorm_kernel<true, false, true>), grid, block, 0, s, x, ld_in, gamma, beta, out, ld_out, rows, D, eps, gelu);
    else if (in32) hipLaunchKernelGGL((layernorm_kernel<true, false>), grid, block, 0, s, x, ld_in, gamma, beta, out, ld_out, rows, D, eps, gelu);
    else if (out32) hipLaunchKernelGGL((layernorm_kernel<false, true>), grid, block, 0, s, x, ld_in, gamma, beta, out, ld_out, rows, D, eps, gelu);
    else hipLaunchKernelGGL((layernorm_kernel<false, false>), grid, block, 0, s, x, ld_in, gamma, beta, out, ld_out, rows, D, eps, gelu);
    SC_CHECK_LAUNCH();
    return 0;
}

// out = LayerNorm(residual + dropout(x)) for bf16 [rows, 768] (the two post-LN sites of a HuBERT-base layer in train mode); returns 1 for any other
// width (the caller then runs sc_dropout_bf16 + sc_layernorm).
extern "C" int sc_dropout_add_layernorm_bf16(const void* x, const void* residual, const float* gamma, const float* beta, void* out, int64_t rows, int D,
                                             float eps, float drop_p, uint32_t seed, void* stream) {
    SC_CHECK_ARG(x && residual && gamma && beta && out, "sc_dropout_add_layernorm_bf16: null operand");
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_dropout_add_layernorm_bf16: drop_p=%f must be in [0, 1)", (double)drop_p);
    if (D != 768 || ((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)out) & 15) != 0) || rows * 768 >= 0xffffffffLL) return 1;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL((layernorm768_kernel<true>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma, beta, (bf16_t*)out,
                       rows, eps, (const bf16_t*)residual, seed, drop_thresh(drop_p), 1.0f / (1.0f - drop_p));
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_weighted_sum_fwd(const void* hidden, int64_t layer_stride, const float* weights, void* out, int n_layers,
                                   int64_t rows, int D, int flags, float eps, void* stream) {
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0, "sc_weighted_sum: D=%d must be a multiple of 4, <= 1024", D);
    SC_CHECK_ARG(n_layers > 0 && n_layers <= 64, "sc_weighted_sum: n_layers=%d out of range", n_layers);
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int normalize = (flags & SC_WS_NORMALIZE) ? 1 : 0;
    if (flags & SC_WS_IN_F32)
        hipLaunchKernelGGL((weighted_sum_kernel<true>), grid, block, 0, (hipStream_t)stream, hidden, layer_stride, weights, out, n_layers, rows, D, normalize, eps);
    else
        hipLaunchKernelGGL((weighted_sum_kernel<false>), grid, block, 0, (hipStream_t)stream, hidden, layer_stride, weights, out, n_layers, rows, D, normalize, eps);
    SC_CHECK_LAUNCH();
    return 0;
}

// Packed -> padded row layout (the API boundary of the padding-free engine): out[l][b][t][:] = t < rows_b - halo ? src[l][row_off[b] + t][:] : 0,
// rows of `row_bytes` bytes (a multiple of 16), rows_b = row_off[b + 1] - row_off[b].  One wave per output row.  `halo` trailing rows of every
// utterance are NOT copied (the packed engine's receptive-field row holds inexact conv features that read the neighbouring utterance: ADVICE r3).
__global__ __launch_bounds__(256) void unpack_rows_kernel(const char* __restrict__ src, int64_t src_layer_stride, const int32_t* __restrict__ row_off,
                                                          char* __restrict__ out, int64_t out_layer_stride, int B, int T_out, int row_bytes, int halo) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (int64_t)B * T_out) return;
    const int b = (int)(r / T_out), t = (int)(r - (int64_t)b * T_out);
    const int l = blockIdx.y;
    const int rows_b = row_off[b + 1] - row_off[b] - halo;
    const char* s = src + l * src_layer_stride + ((int64_t)row_off[b] + t) * row_bytes;
    char* o = out + l * out_layer_stride + r * row_bytes;
    for (int c = lane * 16; c < row_bytes; c += 64 * 16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < rows_b) v = *(const uint4*)(s + c);
        *(uint4*)(o + c) = v;
    }
}

extern "C" int sc_unpack_rows(const void* src, int64_t src_layer_stride_bytes, const int32_t* row_off, void* out, int64_t out_layer_stride_bytes, int n_layers,
                              int B, int T_out, int row_bytes, int halo, void* stream) {
    SC_CHECK_ARG(src && row_off && out && n_layers >= 1 && n_layers <= 65535 && row_bytes > 0 && row_bytes % 16 == 0 && halo >= 0, "sc_unpack_rows: bad arguments (row_bytes must be a multiple of 16)");
    if (B <= 0 || T_out <= 0) return 0;
    const int64_t rows = (int64_t)B * T_out;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((rows + 3) / 4), n_layers), dim3(256), 0, (hipStream_t)stream, (const char*)src, src_layer_stride_bytes,
                       row_off, (char*)out, out_layer_stride_bytes, B, T_out, row_bytes, halo);
    SC_CHECK_LAUNCH();
    return 0;
}


// `normalize_hiddenstates` with `normalize_type` method1 / method2 (speech_encoder_plus.py:572-592), IN PLACE on the stacked hidden states
// x = [n_layers][B][Tp][D] (bf16 post-LN / f32 pre-LN residual stream), as the reference overwrites `layer_results[i]`:
//   method1: row /= (||row||_2 + 1e-8)          method2: row /= mean_{t < T} ||x[i, b, t, :]||_2   (ALL T frames of the padded batch, padded frames included)
// MODE 0: norms[row] = ||row||; MODE 1: method1 in place; MODE 2: row *= inv[row / Tp] in place (inv from hidden_group_inv_mean_kernel).  One wave per row.
template <bool IN_F32, int MODE>
__global__ __launch_bounds__(256) void hidden_rownorm_kernel(void* __restrict__ x, int64_t rows, int D, int Tp, float* __restrict__ norms,
                                                             const float* __restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[MAXC][4];
    load_row<IN_F32>(x, row * D, D, lane, v);
    float scale;
    if (MODE == 2) {
        scale = inv[row / Tp];
    } else {
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) q += (v[c][0] * v[c][0] + v[c][1] * v[c][1]) + (v[c][2] * v[c][2] + v[c][3] * v[c][3]);
        const float nrm = sqrtf(wave_sum(q));
        if (MODE == 0) { if (lane == 0) norms[row] = nrm; return; }
        scale = 1.0f / (nrm + 1e-8f);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int e = c * 256 + lane * 4;
        if (e < D) {
            if (IN_F32) *(f32x4_t*)((float*)x + row * D + e) = (f32x4_t){v[c][0] * scale, v[c][1] * scale, v[c][2] * scale, v[c][3] * scale};
            else { uint2 p_; p_.x = pack2bf(v[c][0] * scale, v[c][1] * scale); p_.y = pack2bf(v[c][2] * scale, v[c][3] * scale); *(uint2*)((bf16_t*)x + row * D + e) = p_; }
        }
    }
}

// inv[g] = T / sum_{t < T} norms[g * Tp + t]   for every (layer, utterance) group g; fixed summation order (one wave per group).
__global__ __launch_bounds__(64) void hidden_group_inv_mean_kernel(const float* __restrict__ norms, float* __restrict__ inv, int Tp, int T) {
    const int64_t g = blockIdx.x;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 64) s += norms[g * Tp + t];
    s = wave_sum(s);
    if (threadIdx.x == 0) inv[g] = (float)T / s;
}

extern "C" int sc_hidden_normalize(void* hidden, int in_f32, int n_layers, int B, int Tp, int T, int D, int method, float* workspace, void* stream) {
    SC_CHECK_ARG(hidden && n_layers > 0 && B > 0 && Tp > 0 && T > 0 && T <= Tp && D > 0 && D <= 1024 && D % 4 == 0, "sc_hidden_normalize: bad shape (D a multiple of 4, <= 1024; T <= Tp)");
    SC_CHECK_ARG(method == 1 || method == 2, "sc_hidden_normalize: method=%d (1: unit frames, 2: utterance-mean frame norm)", method);
    SC_CHECK_ARG(method == 1 || workspace, "sc_hidden_normalize: method 2 needs a workspace of n_layers * B * (Tp + 1) floats");
    const int64_t rows = (int64_t)n_layers * B * Tp;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define SC_HN(F32, MODE, NORMS, INV) hipLaunchKernelGGL((hidden_rownorm_kernel<F32, MODE>), grid, block, 0, s, hidden, rows, D, Tp, NORMS, INV)
    if (method == 1) {
        if (in_f32) SC_HN(true, 1, nullptr, nullptr); else SC_HN(false, 1, nullptr, nullptr);
    } else {
        float* norms = workspace;
        float* inv = workspace + rows;
        if (in_f32) SC_HN(true, 0, norms, nullptr); else SC_HN(false, 0, norms, nullptr);
        hipLaunchKernelGGL(hidden_group_inv_mean_kernel, dim3((unsigned)(n_layers * B)), dim3(64), 0, s, norms, inv, Tp, T);
        if (in_f32) SC_HN(true, 2, nullptr, inv); else SC_HN(false, 2, nullptr, inv);
    }
#undef SC_HN
    SC_CHECK_LAUNCH();
    return 0;
}

// Deterministic split-K finish: out[m, n] = act( sum_s part[s][m][n] + bias[n] ) + residual[m][n], partials summed in the fixed order s = 0..S-1
// (no atomics: the eval path is bitwise run-to-run stable).  act: SC_ACT_NONE / SC_ACT_GELU (erf form).  n_total = M * N, a multiple of 4.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t n_total, int N, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, int64_t ldr, float* __restrict__ out, int act) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n_total) return;
    f32x4_t acc = *(const f32x4_t*)(part + i);
    for (int s = 1; s < S; ++s) acc += *(const f32x4_t*)(part + (int64_t)s * n_total + i);
    const int64_t m = i / N;
    const int n = (int)(i - m * N);
    if (bias) acc += *(const f32x4_t*)(bias + n);
    if (act == SC_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = gelu_erf_precise(acc[r]);
    }
    if (residual) acc += *(const f32x4_t*)(residual + m * ldr + n);
    *(f32x4_t*)(out + i) = acc;
}

extern "C" int sc_splitk_reduce_f32(const float* partials, int nsplit, int64_t M, int N, const float* bias, const float* residual, int64_t ldr, float* out,
                                    int act, void* stream) {
    SC_CHECK_ARG(partials && out && nsplit >= 1 && N > 0 && N % 4 == 0 && (act == SC_ACT_NONE || act == SC_ACT_GELU) && (!residual || ldr % 4 == 0),
                 "sc_splitk_reduce_f32: bad arguments (N and ldr multiples of 4, act none / gelu)");
    if (M <= 0) return 0;
    const int64_t n_total = M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n_total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, nsplit, n_total, N, bias,
                       residual, ldr, out, act);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_l2norm_fwd(const void* x, int64_t ld_in, float* out, int64_t rows, int D, int flags, void* stream) {
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0, "sc_l2norm: D=%d must be a multiple of 4, <= 1024", D);
    if (rows <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const float floor_ = (flags & SC_L2NORM_CLAMP) ? 1e-8f : 0.0f;
    if (flags & SC_L2NORM_IN_F32) hipLaunchKernelGGL((l2norm_kernel<true>), grid, block, 0, (hipStream_t)stream, x, ld_in, out, rows, D, floor_);
    else hipLaunchKernelGGL((l2norm_kernel<false>), grid, block, 0, (hipStream_t)stream, x, ld_in, out, rows, D, floor_);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_wave_layernorm(const float* wav, float* out, const int32_t* lens, int B, int64_t ld, float eps, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(wave_layernorm_kernel, dim3(B, WSEG), dim3(1024), 0, (hipStream_t)stream, wav, out, lens, ld, eps);
    SC_CHECK_LAUNCH();
    return 0;
}
