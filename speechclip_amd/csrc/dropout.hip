// Train-mode dropouts of the FROZEN HuBERT encoder.  Lightning's model.train() puts every sub-module into train mode, so while the reference
// trains the pooling heads its frozen encoder still applies the checkpoint's dropouts: dropout_input on the projected features
// (avssl/module/speech_encoder_plus.py:87), F.dropout after the positional conv (:42) and, inside every [3P fairseq]
// TransformerSentenceEncoderLayer, on the attention probabilities (sc_attention_fwd_dropout), after out_proj (dropout1), after the activation
// (dropout2 = activation_dropout) and after fc2 (dropout3).
//   sc_dropout_bf16: out = [residual +] dropout(x), bf16, element i kept iff hash(seed, i) >= p 2^32 and scaled by 1 / (1 - p); in place allowed.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ residual, bf16_t* __restrict__ out, int64_t n,
                                                           uint32_t seed, uint32_t thresh, float keep_scale) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const uint2 xv = *(const uint2*)(x + i);
    float v[4] = {lo2f(xv.x), hi2f(xv.x), lo2f(xv.y), hi2f(xv.y)};
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = keep_elem(seed, (uint32_t)(i + j), thresh) ? v[j] * keep_scale : 0.f;
    if (residual) {
        const uint2 rv = *(const uint2*)(residual + i);
        v[0] += lo2f(rv.x); v[1] += hi2f(rv.x); v[2] += lo2f(rv.y); v[3] += hi2f(rv.y);
    }
    uint2 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
    *(uint2*)(out + i) = o;
}

}  // namespace

extern "C" int sc_dropout_bf16(const void* x, const void* residual, void* out, int64_t n, float drop_p, uint32_t seed, void* stream) {
    SC_CHECK_ARG(x && out, "sc_dropout_bf16: null operand");
    SC_CHECK_ARG(n % 4 == 0 && n < 0xffffffffLL, "sc_dropout_bf16: n=%lld must be a multiple of 4 and fit 32 bits", (long long)n);
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_dropout_bf16: drop_p=%f must be in [0, 1)", (double)drop_p);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)residual,
                       (bf16_t*)out, n, seed, drop_thresh(drop_p), 1.0f / (1.0f - drop_p));
    SC_CHECK_LAUNCH();
    return 0;
}
