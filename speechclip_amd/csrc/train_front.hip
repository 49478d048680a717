// Backward of the HuBERT FRONT END for `audio_encoder.trainable: true` without layer lists (avssl/module/speech_encoder_plus.py:399-401: nothing
// is frozen, so the conv feature extractor, post_extract_proj, layer_norm and the positional conv train too; [3P fairseq] scales the feature
// extractor's gradient by feature_grad_mult).  The GEMM-shaped parts run on sc_gemm_bf16 / sc_gemm_bf16_batched / sc_posconv_conv from Python
// (speechclip_amd/train_front.py); this file holds what they cannot do:
//   sc_posconv_finish_train   training forward of the positional-conv tail: u = conv + bias (regrouped to [B*Tp, D]) and s = mask(x) + gelu(u),
//                             both kept for the backward (the eval kernel fuses them with the LayerNorm and keeps neither)
//   sc_posconv_dgrad_finish   dx = mask(ds + time-reversed regroup of the transposed conv): the input gradient of the grouped conv is the SAME
//                             "pad Kw/2, drop the last output" conv applied to the time-reversed gradient with in/out channels swapped
//   sc_reverse_rows_bf16      out[b, t, :] = in[b, T-1-t, :]
//   sc_conv0_bwd              conv layer 0 (Conv1d(1 -> C, k=10, s=5, no bias) -> GroupNorm(C groups) over time -> GELU) backward from the wave:
//                             per-(b, c) partial gradients of the conv weight [10], gamma and beta (summed over b by sc_colsum)
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// u[row, e] = conv[b, g, t, ci] + bias[e];  s[row, e] = (t < valid[b] ? x[row, e] : 0) + gelu(u)
__global__ __launch_bounds__(256) void posconv_finish_train_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ valid, const bf16_t* __restrict__ conv,
                                                                   const float* __restrict__ bias, bf16_t* __restrict__ u, bf16_t* __restrict__ s, int B, int Tp,
                                                                   int D, int G) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= (int64_t)B * Tp * D) return;
    const int64_t row = idx / D;
    const int e = (int)(idx - row * D);
    const int b = (int)(row / Tp), t = (int)(row - (int64_t)b * Tp);
    const int cg = D / G, g = e / cg, ci = e - g * cg;
    const uint2 cv = *(const uint2*)(conv + (((int64_t)b * G + g) * Tp + t) * cg + ci);
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < valid[b]) {
        const uint2 xx = *(const uint2*)(x + idx);
        xv[0] = lo2f(xx.x); xv[1] = hi2f(xx.x); xv[2] = lo2f(xx.y); xv[3] = hi2f(xx.y);
    }
    const float uv[4] = {lo2f(cv.x) + bias[e], hi2f(cv.x) + bias[e + 1], lo2f(cv.y) + bias[e + 2], hi2f(cv.y) + bias[e + 3]};
    uint2 uo, so;
    uo.x = pack2bf(uv[0], uv[1]); uo.y = pack2bf(uv[2], uv[3]);
    // gelu of the bf16-ROUNDED pre-activation: the backward differentiates gelu at exactly the value it is handed
    const float ur[4] = {lo2f(uo.x), hi2f(uo.x), lo2f(uo.y), hi2f(uo.y)};
    so.x = pack2bf(xv[0] + gelu_erf_precise(ur[0]), xv[1] + gelu_erf_precise(ur[1]));
    so.y = pack2bf(xv[2] + gelu_erf_precise(ur[2]), xv[3] + gelu_erf_precise(ur[3]));
    *(uint2*)(u + idx) = uo;
    *(uint2*)(s + idx) = so;
}

// dx[row, e] = t < valid[b] ? ds[row, e] + convT[b, g, Tp-1-t, ci] : 0
__global__ __launch_bounds__(256) void posconv_dgrad_finish_kernel(const bf16_t* __restrict__ convT, const bf16_t* __restrict__ ds, const int32_t* __restrict__ valid,
                                                                   bf16_t* __restrict__ dx, int B, int Tp, int D, int G) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= (int64_t)B * Tp * D) return;
    const int64_t row = idx / D;
    const int e = (int)(idx - row * D);
    const int b = (int)(row / Tp), t = (int)(row - (int64_t)b * Tp);
    uint2 o = make_uint2(0u, 0u);
    if (t < valid[b]) {
        const int cg = D / G, g = e / cg, ci = e - g * cg;
        const uint2 cv = *(const uint2*)(convT + (((int64_t)b * G + g) * Tp + (Tp - 1 - t)) * cg + ci);
        const uint2 dv = *(const uint2*)(ds + idx);
        o.x = pack2bf(lo2f(cv.x) + lo2f(dv.x), hi2f(cv.x) + hi2f(dv.x));
        o.y = pack2bf(lo2f(cv.y) + lo2f(dv.y), hi2f(cv.y) + hi2f(dv.y));
    }
    *(uint2*)(dx + idx) = o;
}

__global__ __launch_bounds__(256) void reverse_rows_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int T, int D) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= (int64_t)B * T * D) return;
    const int64_t row = idx / D;
    const int e = (int)(idx - row * D);
    const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
    *(uint2*)(out + ((int64_t)b * T + (T - 1 - t)) * D + e) = *(const uint2*)(in + idx);
}

// conv layer 0 backward.  Block = (b, 64 channels); lane = channel, the 4 waves split the frames.  Three sweeps over the T0 frames, the conv
// recomputed from the wave each time (10 FMAs): (A) mean / variance of u over time, (B) S1 = sum dzhat, S2 = sum dzhat uhat, dgamma, dbeta,
// (C) du = rstd (dzhat - S1/T - uhat S2/T) and dw[j] += du wav[5 t + j].  dy rows >= T0 of an utterance do not exist in the reference (alignment
// padding of the channels-last buffer) and are not read.
constexpr int C0_K = 10, C0_S = 5;
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ wav, int64_t ld, const float* __restrict__ w, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const bf16_t* __restrict__ dy, float* __restrict__ part, int C, int T0,
                                                        int P, float eps) {
    __shared__ float red[4][64][12];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the frame index below is wave-uniform,
    const int b = blockIdx.y, c = blockIdx.x * 64 + lane;                                            // so the ten wave samples of a frame are scalar loads
    const float* wv = wav + (int64_t)b * ld;
    float wk[C0_K];
#pragma unroll
    for (int j = 0; j < C0_K; ++j) wk[j] = w[c * C0_K + j];
    auto conv_at = [&](int t, float (&x)[C0_K]) -> float {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < C0_K; ++j) { x[j] = wv[t * C0_S + j]; a = fmaf(wk[j], x[j], a); }     // wave-uniform address: scalar loads
        return a;
    };
    auto block_sum = [&](float (&v)[12], int n) {      // sums v[0..n) over the 4 waves; result in every thread
        __syncthreads();
        for (int i = 0; i < n; ++i) red[wave][lane][i] = v[i];
        __syncthreads();
        for (int i = 0; i < n; ++i) v[i] = (red[0][lane][i] + red[1][lane][i]) + (red[2][lane][i] + red[3][lane][i]);
    };
    float x[C0_K];
    float acc[12];
    // (A) statistics in ONE sweep: sum and sum of squares in fp64 (the frames of a 10 s wave are ~3e4 terms; E[u^2] - E[u]^2 in fp32 would
    // lose the variance of a channel with a DC offset)
    double su = 0.0, sq = 0.0;
    for (int t = wave; t < T0; t += 4) { const double v = (double)conv_at(t, x); su += v; sq += v * v; }
    {
        __shared__ double dred[4][64][2];
        dred[wave][lane][0] = su; dred[wave][lane][1] = sq;
        __syncthreads();
        su = (dred[0][lane][0] + dred[1][lane][0]) + (dred[2][lane][0] + dred[3][lane][0]);
        sq = (dred[0][lane][1] + dred[1][lane][1]) + (dred[2][lane][1] + dred[3][lane][1]);
    }
    const double mean_d = su / (double)T0;
    const float mean = (float)mean_d;
    const float rstd = rsqrtf((float)(sq / (double)T0 - mean_d * mean_d) + eps);
    const float gm = gamma[c], bt = beta[c];
    const bf16_t* dyb = dy + (int64_t)b * P * C + c;
    // (B)
    float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f;
    for (int t = wave; t < T0; t += 4) {
        const float uh = (conv_at(t, x) - mean) * rstd;
        const float dz = bf2f(dyb[(int64_t)t * C]) * gelu_grad(fmaf(gm, uh, bt));
        dg = fmaf(dz, uh, dg);
        db += dz;
        const float dzh = dz * gm;
        s1 += dzh;
        s2 = fmaf(dzh, uh, s2);
    }
    acc[0] = s1; acc[1] = s2; acc[2] = dg; acc[3] = db;
    block_sum(acc, 4);
    const float m1 = acc[0] / (float)T0, m2 = acc[1] / (float)T0;
    dg = acc[2]; db = acc[3];
    // (C)
    float dw[C0_K];
#pragma unroll
    for (int j = 0; j < C0_K; ++j) dw[j] = 0.f;
    for (int t = wave; t < T0; t += 4) {
        const float uh = (conv_at(t, x) - mean) * rstd;
        const float dz = bf2f(dyb[(int64_t)t * C]) * gelu_grad(fmaf(gm, uh, bt));
        const float du = rstd * (dz * gm - m1 - uh * m2);
#pragma unroll
        for (int j = 0; j < C0_K; ++j) dw[j] = fmaf(du, x[j], dw[j]);
    }
#pragma unroll
    for (int j = 0; j < C0_K; ++j) acc[j] = dw[j];
    block_sum(acc, C0_K);
    if (wave == 0) {
        float* o = part + ((int64_t)b * C + c) * 12;
#pragma unroll
        for (int j = 0; j < C0_K; ++j) o[j] = acc[j];
        o[10] = dg;
        o[11] = db;
    }
}

// conv layer 0 of the LayerNorm extractor (HuBERT-large: Conv1d(1 -> C, k 10, s 5, bias) -> LayerNorm(C) -> GELU): the LayerNorm / GELU part of the
// backward runs on the row kernels, which leaves dw[c, j] = sum_t du[t, c] wav[5 t + j] and dbias[c] = sum_t du[t, c] per utterance.
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ wav, int64_t ld, const bf16_t* __restrict__ du, float* __restrict__ part,
                                                          int C, int T0, int P) {
    __shared__ float red[4][64][12];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, c = blockIdx.x * 64 + lane;
    const float* wv = wav + (int64_t)b * ld;
    const bf16_t* dub = du + (int64_t)b * P * C + c;
    float acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.f;
    for (int t = wave; t < T0; t += 4) {
        const float g = bf2f(dub[(int64_t)t * C]);
#pragma unroll
        for (int j = 0; j < C0_K; ++j) acc[j] = fmaf(g, wv[t * C0_S + j], acc[j]);
        acc[10] += g;
    }
    for (int i = 0; i < 11; ++i) red[wave][lane][i] = acc[i];
    __syncthreads();
    if (wave == 0) {
        float* o = part + ((int64_t)b * C + c) * 12;
        for (int i = 0; i < 11; ++i) o[i] = (red[0][lane][i] + red[1][lane][i]) + (red[2][lane][i] + red[3][lane][i]);
        o[11] = 0.f;
    }
}

}  // namespace

extern "C" int sc_conv0_wgrad(const float* wav, int64_t ld, const void* du, float* part, int B, int C, int T0, int P, void* stream) {
    SC_CHECK_ARG(wav && du && part, "sc_conv0_wgrad: null operand");
    SC_CHECK_ARG(C % 64 == 0 && B <= 65535, "sc_conv0_wgrad: C must be a multiple of 64, B <= 65535");
    SC_CHECK_ARG(T0 >= 1 && P >= T0 && ld >= (int64_t)(T0 - 1) * C0_S + C0_K, "sc_conv0_wgrad: T0=%d P=%d ld=%lld inconsistent", T0, P, (long long)ld);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(conv0_wgrad_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream, wav, ld, (const bf16_t*)du, part, C, T0, P);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_posconv_finish_train(const void* x, const int32_t* valid, const void* conv, const float* bias, void* u, void* s, int B, int Tp, int D, int G,
                                       void* stream) {
    SC_CHECK_ARG(x && valid && conv && bias && u && s, "sc_posconv_finish_train: null operand");
    SC_CHECK_ARG(D % G == 0 && (D / G) % 4 == 0, "sc_posconv_finish_train: D/G must be a multiple of 4");
    const int64_t n4 = (int64_t)B * Tp * D / 4;
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(posconv_finish_train_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, valid,
                       (const bf16_t*)conv, bias, (bf16_t*)u, (bf16_t*)s, B, Tp, D, G);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_posconv_dgrad_finish(const void* convT, const void* ds, const int32_t* valid, void* dx, int B, int Tp, int D, int G, void* stream) {
    SC_CHECK_ARG(convT && ds && valid && dx, "sc_posconv_dgrad_finish: null operand");
    SC_CHECK_ARG(D % G == 0 && (D / G) % 4 == 0, "sc_posconv_dgrad_finish: D/G must be a multiple of 4");
    const int64_t n4 = (int64_t)B * Tp * D / 4;
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(posconv_dgrad_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)convT, (const bf16_t*)ds,
                       valid, (bf16_t*)dx, B, Tp, D, G);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_reverse_rows_bf16(const void* in, void* out, int B, int T, int D, void* stream) {
    SC_CHECK_ARG(in && out && in != out, "sc_reverse_rows_bf16: null operand or in-place");
    SC_CHECK_ARG(D % 4 == 0, "sc_reverse_rows_bf16: D must be a multiple of 4");
    const int64_t n4 = (int64_t)B * T * D / 4;
    if (n4 <= 0) return 0;
    hipLaunchKernelGGL(reverse_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, B, T, D);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_conv0_bwd(const float* wav, int64_t ld, const float* w, const float* gamma, const float* beta, const void* dy, float* part, int B, int C, int T0,
                            int P, float eps, void* stream) {
    SC_CHECK_ARG(wav && w && gamma && beta && dy && part, "sc_conv0_bwd: null operand");
    SC_CHECK_ARG(C % 64 == 0 && B <= 65535, "sc_conv0_bwd: C must be a multiple of 64, B <= 65535");
    SC_CHECK_ARG(T0 >= 1 && P >= T0 && ld >= (int64_t)(T0 - 1) * C0_S + C0_K, "sc_conv0_bwd: T0=%d P=%d ld=%lld inconsistent", T0, P, (long long)ld);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(conv0_bwd_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream, wav, ld, w, gamma, beta, (const bf16_t*)dy, part, C, T0, P, eps);
    SC_CHECK_LAUNCH();
    return 0;
}
