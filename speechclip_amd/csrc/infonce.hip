// Batched masked InfoNCE (avssl/module/losses.py:185-245) on the gathered global batch, fp32.
//   logits = A.B^T * inv_t (diag -= margin);  neg[i,j] = (id_i != id_j) | (i == j unless dcl)
//   loss = 1/2 [ mean_i(-l_ii + log sum_j e^{l_ij} neg_ij) + mean_j(-l_jj + log sum_i e^{l_ij} neg_ij) ]
// Kernel 1: 64x64 logit tiles (fp32 FMA, LDS-staged), masked exp, per-tile row/column partial sums
// written to [tiles][Bg] slabs (deterministic: no atomics).  Kernel 2: reduce slabs -> scalars.
// No MAX_EYE=256 limit: works for any global batch (2048 at 8 GPUs x 256).
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

constexpr int TS = 64, KC = 16, PAD = 4;

__global__ __launch_bounds__(256) void infonce_tile_kernel(const float* __restrict__ a, const float* __restrict__ b, const int64_t* __restrict__ ids,
                                                           float* __restrict__ prow, float* __restrict__ pcol, float* __restrict__ pos, int Bg, int E,
                                                           float inv_t, float margin, int dcl) {
    __shared__ float sA[KC][TS + PAD], sB[KC][TS + PAD];
    __shared__ float colred[16][TS];
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
        if (m0 + lr < Bg && k0 + lk < E) va = *(const f32x4_t*)(a + (int64_t)(m0 + lr) * E + k0 + lk);
        if (n0 + lr < Bg && k0 + lk < E) vb = *(const f32x4_t*)(b + (int64_t)(n0 + lr) * E + k0 + lk);
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
    float rsum[4] = {0.f, 0.f, 0.f, 0.f}, csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = m0 + ty * 4 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gj = n0 + tx * 4 + j;
            float e = 0.f;
            if (gi < Bg && gj < Bg) {
                float l = acc[i][j] * inv_t;
                const bool diag = gi == gj;
                if (diag) { if (margin > 0.f) l -= margin; pos[gi] = l; }
                bool neg = ids ? (ids[gi] != ids[gj]) : !diag;
                if (!dcl && diag) neg = true;
                e = neg ? __expf(l) : 0.f;
            }
            rsum[i] += e;
            csum[j] += e;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r = rsum[i];
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 8, 64);
        const int gi = m0 + ty * 4 + i;
        if (tx == 0 && gi < Bg) prow[(int64_t)blockIdx.x * Bg + gi] = r;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) colred[ty][tx * 4 + j] = csum[j];
    __syncthreads();
    if (tid < TS) {
        float c = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) c += colred[y][tid];
        if (n0 + tid < Bg) pcol[(int64_t)blockIdx.y * Bg + n0 + tid] = c;
    }
}

__global__ __launch_bounds__(256) void infonce_final_kernel(const float* __restrict__ prow, const float* __restrict__ pcol, const float* __restrict__ pos,
                                                            float* __restrict__ out, int Bg, int ntiles, int a2b, int b2a) {
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double sa = 0.0, sb = 0.0;
    for (int i = tid; i < Bg; i += 256) {
        float rs = 0.f, cs = 0.f;
        for (int t = 0; t < ntiles; ++t) { rs += prow[(int64_t)t * Bg + i]; cs += pcol[(int64_t)t * Bg + i]; }
        sa += (double)(-pos[i] + logf(rs));
        sb += (double)(-pos[i] + logf(cs));
    }
    sa = wave_sum_d(sa); sb = wave_sum_d(sb);
    if (lane == 0) { red[0][wv] = sa; red[1][wv] = sb; }
    __syncthreads();
    if (tid == 0) {
        const double la = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / Bg;
        const double lb = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / Bg;
        double loss = (a2b ? la : 0.0) + (b2a ? lb : 0.0);
        if (a2b && b2a) loss *= 0.5;
        out[0] = (float)loss; out[1] = (float)la; out[2] = (float)lb;
    }
}

}  // namespace

extern "C" int64_t sc_infonce_workspace_bytes(int Bg) {
    const int64_t nt = (Bg + TS - 1) / TS;
    return (2 * nt * Bg + Bg) * 4;
}

extern "C" int sc_infonce_fwd(const float* feat_a, const float* feat_b, const int64_t* ids, void* workspace, float* out3, int Bg, int E,
                              float inv_temperature, float margin, int dcl, int a2b, int b2a, void* stream) {
    SC_CHECK_ARG(Bg > 0 && E > 0 && E % 4 == 0, "sc_infonce_fwd: E=%d must be a positive multiple of 4", E);
    SC_CHECK_ARG(a2b || b2a, "sc_infonce_fwd: a2b and b2a cannot both be false");
    const int nt = (Bg + TS - 1) / TS;
    float* prow = (float*)workspace;
    float* pcol = prow + (int64_t)nt * Bg;
    float* pos = pcol + (int64_t)nt * Bg;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(infonce_tile_kernel, dim3(nt, nt), dim3(256), 0, s, feat_a, feat_b, ids, prow, pcol, pos, Bg, E, inv_temperature, margin, dcl);
    SC_CHECK_LAUNCH();
    hipLaunchKernelGGL(infonce_final_kernel, dim3(1), dim3(256), 0, s, prow, pcol, pos, out3, Bg, nt, a2b, b2a);
    SC_CHECK_LAUNCH();
    return 0;
}
