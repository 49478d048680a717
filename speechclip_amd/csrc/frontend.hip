// Front-end kernels of the two towers (HBM-bound segments):
//   * HuBERT conv layer 0 (Conv1d 1->C, k=10, s=5) fused with GroupNorm(C groups)+GELU.  The GroupNorm
//     statistics over the whole time axis are obtained WITHOUT a first conv pass: with
//     y[c,t] = sum_j w[c,j] x[5t+j],  sum_t y = w_c . s  and  sum_t y^2 = w_c^T R w_c  where
//     s[j] = sum_t x[5t+j], R[j,j'] = sum_t x[5t+j] x[5t+j'] (10 + 55 numbers per utterance, fp64).
//   * positional grouped conv (k=128, 16 groups): pack into a group-major zero-padded layout so each
//     group becomes an overlapping-row GEMM, and the finishing gather + bias + GELU + residual (+LayerNorm).
//   * CLIP ViT stem: patchify (im2col for the stride=patch conv) and class/positional embedding + ln_pre.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

constexpr int CK = 10, CS = 5;        // conv0 kernel / stride
constexpr int NSTAT = CK + CK * (CK + 1) / 2;  // 65
constexpr int NSPLIT = 16;
#ifndef SC_CONV0_FB_DEFAULT
#define SC_CONV0_FB_DEFAULT 512
#endif
#ifndef SC_CONV0_THREADS_DEFAULT
#define SC_CONV0_THREADS_DEFAULT 512      // 8 waves share one 32 KiB copy of the W fragments: 6 instead of 4 waves per SIMD fit (LDS was the limit); -0.17 ms, profiles/r06_conv0_ab.txt
#endif

// (round 6: products and the per-thread / per-wave partial sums in fp32 -- a thread sees <= 8 frames, a wave 512: ~1e-7 relative --, fp64 only across waves, splits and
//  in conv0_coef_kernel where the variance's cancellation happens; 16 instead of 8 splits per utterance.  The all-fp64 form spent half its time in 65 double-precision wave
//  reductions and ran at 0.12 of the HBM roofline as the first, un-overlapped kernel of the step.)
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ wav, int64_t ld, int T0, double* __restrict__ partial) {
    __shared__ double red[4][NSTAT];
    const int b = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* x = wav + (int64_t)b * ld;
    const int per = (T0 + NSPLIT - 1) / NSPLIT;
    const int t_lo = sp * per, t_hi = min(T0, t_lo + per);
    float acc[NSTAT];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) acc[i] = 0.f;
    for (int t = t_lo + tid; t < t_hi; t += 256) {
        float xv[CK];
#pragma unroll
        for (int j = 0; j < CK; ++j) xv[j] = x[(int64_t)CS * t + j];
        int idx = CK;
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            acc[j] += xv[j];
#pragma unroll
            for (int k = j; k < CK; ++k) { acc[idx] = fmaf(xv[j], xv[k], acc[idx]); ++idx; }
        }
    }
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) {
        const float r = wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = (double)r;
    }
    __syncthreads();
    if (tid < NSTAT) partial[((int64_t)b * NSPLIT + sp) * NSTAT + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// coef[b][c] = (scale, shift) so that GroupNorm(conv)[c,t] = conv[c,t]*scale + shift
__global__ __launch_bounds__(256) void conv0_coef_kernel(const double* __restrict__ partial, const float* __restrict__ w, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float2* __restrict__ coef, int C, int T0, float eps) {
    __shared__ double st[NSTAT];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < NSTAT) {
        double t = 0;
        for (int sp = 0; sp < NSPLIT; ++sp) t += partial[((int64_t)b * NSPLIT + sp) * NSTAT + tid];
        st[tid] = t;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double wv[CK];
        for (int j = 0; j < CK; ++j) wv[j] = (double)w[c * CK + j];
        double sum = 0, sq = 0;
        int idx = CK;
        for (int j = 0; j < CK; ++j) {
            sum += wv[j] * st[j];
            for (int k = j; k < CK; ++k) sq += (j == k ? 1.0 : 2.0) * wv[j] * wv[k] * st[idx++];
        }
        const double mean = sum / T0;
        double var = sq / T0 - mean * mean;
        var = var > 0 ? var : 0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const double sc = (double)gamma[c] * rstd;
        coef[(int64_t)b * C + c] = make_float2((float)sc, (float)((double)beta[c] - mean * sc));
    }
}

// out[b][t][c] (channels-last bf16, P rows per utterance).  mode 0: GroupNorm coefficients + GELU; mode 1: raw conv + bias.
// A lane owns 8 adjacent channels (held as 4 float2 so conv / normalise / GELU polynomial issue as v_pk_* ops), a wave one
// frame at a time: one 16-byte store per lane = one full 1 KiB row per wave instruction, and the 10 broadcast LDS reads of
// the frame's samples are shared by 8 channels.  The kernel is VALU-bound (the 8.4 GB output is 1.3 ms of HBM time).
constexpr int TT = 64;
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ wav, int64_t ld, int64_t L, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float2* __restrict__ coef, bf16_t* __restrict__ out,
                                                        int C, int T0, int P, int mode) {
    __shared__ float xs[TT * CS + CK];
    const int b = blockIdx.y, t0 = blockIdx.x * TT, tid = threadIdx.x;
    const float* x = wav + (int64_t)b * ld;
    for (int i = tid; i < TT * CS + CK; i += 256) {
        int64_t si = (int64_t)t0 * CS + i;
        xs[i] = si < L ? x[si] : 0.f;
    }
    const int lane = tid & 63, wave = tid >> 6;
    for (int cb = 0; cb < C; cb += 512) {            // C <= 512 in every shipped config: one pass
        const int c0 = cb + lane * 8;
        const bool active = c0 < C;
        f32x2_t w2[4][CK], sc[4], sh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = active ? c0 + 2 * q : 0;
#pragma unroll
            for (int j = 0; j < CK; ++j) w2[q][j] = (f32x2_t){w[c * CK + j], w[(c + 1) * CK + j]};
            sc[q] = (f32x2_t){1.f, 1.f};
            sh[q] = (f32x2_t){0.f, 0.f};
            if (mode == 0) {
                const float2 a = coef[(int64_t)b * C + c], d = coef[(int64_t)b * C + c + 1];
                sc[q] = (f32x2_t){a.x, d.x}; sh[q] = (f32x2_t){a.y, d.y};
            } else if (bias) {
                sh[q] = (f32x2_t){bias[c], bias[c + 1]};
            }
        }
        __syncthreads();
        for (int f = wave; f < TT; f += 4) {
            const int t = t0 + f;
            if (t >= P) break;
            uint4 o = make_uint4(0, 0, 0, 0);
            if (t < T0) {
                f32x2_t y[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < CK; ++j) {
                    const float xv = xs[f * CS + j];
                    const f32x2_t x2 = {xv, xv};
#pragma unroll
                    for (int q = 0; q < 4; ++q) y[q] = w2[q][j] * x2 + y[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    y[q] = y[q] * sc[q] + sh[q];
                    if (mode == 0) y[q] = gelu_erf2(y[q]);
                }
                o = make_uint4(pack2bf(y[0][0], y[0][1]), pack2bf(y[1][0], y[1][1]), pack2bf(y[2][0], y[2][1]), pack2bf(y[3][0], y[3][1]));
            }
            if (active) *(uint4*)(out + ((int64_t)b * P + t) * C + c0) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------- conv0 on the matrix cores
// The VALU form above spends 40 packed FMAs per 8 channels per frame on the 10-tap dot products.  Here a 16-frame x 16-channel block
// is ONE v_mfma_f32_16x16x32_bf16: the 32 k-slots carry  x_hi.w_hi (taps 0-9) + x_lo.w_hi (10-19) + x_hi.w_lo (20-29)  with x = x_hi +
// x_lo and w' = w_hi + w_lo the bf16 splits of the fp32 samples / weights (w' = w * GroupNorm scale of this utterance and channel, so
// only the shift remains for the epilogue).  The dropped x_lo.w_lo term is 2^-16 relative.  What is left on the VALU is the shift, the
// GELU polynomial and the bf16 pack; the output leaves through the GEMM epilogue's quad-contiguous store (permlane swap + crossbar
// transpose: 16 frames x 64 contiguous bytes per store instruction).  Block = one utterance x FB frames, 4 waves x 16-frame groups.
// W fragments, built once per utterance: wfrag[b][cb][lane] (16 bytes) = channel cb*16 + (lane & 15), k-slots (lane >> 4)*8 .. +7
// (slot -> (kind, tap): 0-9 w_hi, 10-19 w_hi, 20-29 w_lo, 30 / 31 the shift's bf16 hi / lo halves), weights pre-multiplied by the GroupNorm scale of (b, channel).
__global__ __launch_bounds__(256) void conv0_wfrag_kernel(const float* __restrict__ w, const float2* __restrict__ coef, const float* __restrict__ bias, bf16x8_t* __restrict__ wfrag,
                                                          int C, int mode) {
    const int b = blockIdx.x, ncb = C / 16;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int cb = i >> 6, lane = i & 63, frow = lane & 15, fk = lane >> 4;
        const int c = cb * 16 + frow;
        const float scl = (cb < ncb) ? (mode == 0 ? coef[(int64_t)b * C + c].x : 1.0f) : 0.f;
        const float shift = (cb < ncb) ? (mode == 0 ? coef[(int64_t)b * C + c].y : (bias ? bias[c] : 0.f)) : 0.f;      // (mode 2: conv + bias here, LayerNorm in the forward kernel)
        const __bf16 shift_hi = (__bf16)shift;
        bf16x8_t f;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            const int slot = fk * 8 + sidx;
            const int tap = slot % CK, kind = slot / CK;
            const float wv = (cb < ncb && slot < 30) ? w[c * CK + tap] * scl : 0.f;
            const __bf16 hi = (__bf16)wv;
            f[sidx] = slot == 30 ? shift_hi : slot == 31 ? (__bf16)(shift - (float)shift_hi) : kind == 2 ? (__bf16)(wv - (float)hi) : hi;
        }
        wfrag[((int64_t)b * 32 + cb) * 64 + lane] = f;
    }
}
template <int FB>   // frames per block
__global__ __launch_bounds__(512) void conv0_mfma_kernel(const float* __restrict__ wav, int64_t ld, int64_t L, const bf16x8_t* __restrict__ wfrag,
                                                         const float* __restrict__ bias, const float2* __restrict__ coef, bf16_t* __restrict__ out,
                                                         int C, int T0, int P_uniform, int mode, const int32_t* __restrict__ row_off, int row_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_c0[];
    bf16x8_t* wl = (bf16x8_t*)smem_c0;                    // [32][64] W fragments (32 KiB)
    float* lnp = (float*)(smem_c0 + 32 * 64 * 16);        // mode 2: LayerNorm gamma[512] | beta[512] of the extractor's layer 0 (HuBERT-large: extractor_mode "layer_norm")
    float* xs = lnp + 1024;                               // FB * 5 + 8 samples
    const int b = blockIdx.y, t0 = blockIdx.x * FB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // packed batches: utterance b owns output rows [row_scale * row_off[b], row_scale * row_off[b + 1]) (row_off counts transformer frames,
    // row_scale = the conv stack's total stride after layer 0); frames >= T0 are written as zeros as in the uniform layout
    const int P = row_off ? row_scale * (row_off[b + 1] - row_off[b]) : P_uniform;
    const int64_t orow0 = row_off ? (int64_t)row_scale * row_off[b] : (int64_t)b * P;
    if (t0 >= P) return;
    const float* x = wav + (int64_t)b * ld;
    const int nfr = min(FB, P - t0);
    const int nthr = blockDim.x, nwav = nthr >> 6;      // 256 or 512 threads (the W fragments in LDS are shared by all waves of the block)
    for (int i = tid; i < nfr * CS + CK; i += nthr) {
        const int64_t si = (int64_t)t0 * CS + i;
        xs[i] = si < L ? x[si] : 0.f;
    }
    for (int i = tid; i < 32 * 64; i += nthr) wl[i] = wfrag[(int64_t)b * 32 * 64 + i];
    float ln_eps = 0.f;
    if (mode == 2) {      // coef = gamma[C] | beta[C] | eps (floats)
        const float* lp = (const float*)coef;
        for (int c = tid; c < 1024; c += nthr) lnp[c] = (c & 511) < C ? lp[(c >> 9) * C + (c & 511)] : 0.f;
        ln_eps = lp[2 * C];
    }
    const int frow = lane & 15, fk = lane >> 4;
    const int ncb = C / 16;
    const int srow = lane >> 2, schunk = lane & 3;       // epilogue lane geometry (as gemm256_kernel)
    const int bperm = ((((schunk & 1) << 1) | (schunk >> 1)) * 16 + srow) << 2;
    __syncthreads();
    const int ngroups = nfr / 16;                         // P is a multiple of 64
    for (int gidx = wave; gidx < ngroups; gidx += nwav) {
        const int tg = t0 + gidx * 16;                    // first frame of the group
        // X fragment: frame tg + frow, slots fk*8..+7 (0-9 x_hi, 10-19 x_lo, 20-29 x_hi, 30-31 one: the shift's slots)
        bf16x8_t xf;
        const float* xr = xs + (gidx * 16 + frow) * CS;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            const int slot = fk * 8 + sidx;
            const int tap = slot % CK, kind = slot / CK;
            const float xv = slot < 30 ? xr[tap] : 1.0f;
            const __bf16 hi = (__bf16)xv;
            xf[sidx] = (kind == 1 && slot < 30) ? (__bf16)(xv - (float)hi) : hi;
        }
        const bool live_row = (tg + srow) < T0;           // frames in [T0, P) are written as zeros
        const bool all_live = tg + 16 <= T0;              // (wave-uniform: only an utterance's last group has dead frames)
        bf16_t* orow = out + (orow0 + tg + srow) * C + schunk * 8;
        // mode 2 (conv + bias -> LayerNorm over the C channels of every frame -> GELU): a first pass over the chunks for the per-frame sums (a lane's 4 x 4 values of
        // a chunk all belong to frame lane & 15; the 4 lane groups of a frame meet by two shuffles), then the same MFMAs again for the output -- the kernel is bound by
        // its stores, the matrix work is ~1 % of it.  Saves the separate LayerNorm + GELU pass over the largest activation of the extractor (4.2 GB of traffic at B = 64).
        float ln_a = 1.f, ln_b = 0.f;                     // y_norm = y * ln_a + ln_b (then * gamma + beta per channel)
        if (mode == 2) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q * 4 >= ncb) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[(q * 4 + j) * 64 + lane], xf, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    s1 += (a4[0] + a4[1]) + (a4[2] + a4[3]);
                    s2 += (a4[0] * a4[0] + a4[1] * a4[1]) + (a4[2] * a4[2] + a4[3] * a4[3]);
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            const float mean = s1 / (float)C;
            const float var = fmaxf(s2 / (float)C - mean * mean, 0.f);
            ln_a = rsqrtf(var + ln_eps);
            ln_b = -mean * ln_a;
        }
        f32x4_t acc[2][4];                                // two 64-channel chunks in flight: MFMAs of chunk q+1 issue before the epilogue of q
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j * 64 + lane], xf, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {                     // 64 channels at a time
            if (q * 4 >= ncb) break;
            if ((q + 1) * 4 < ncb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[(q + 1) & 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[((q + 1) * 4 + j) * 64 + lane], xf, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            uint2 pk[4];
            if (mode == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t g4 = *(const f32x4_t*)(lnp + (q * 4 + j) * 16 + fk * 4), b4 = *(const f32x4_t*)(lnp + 512 + (q * 4 + j) * 16 + fk * 4);
                    acc[q & 1][j] = (acc[q & 1][j] * ln_a + ln_b) * g4 + b4;
                }
            }
            if (mode != 1) {      // all 8 value pairs of the chunk side by side (gelu_poly2_x8: the per-pair Horner chain is latency-bound -- one hazard s_nop per v_pk_fma_f16)
                f32x2_t xs8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t v4 = acc[q & 1][j];
                    xs8[2 * j] = (f32x2_t){v4[0], v4[1]}; xs8[2 * j + 1] = (f32x2_t){v4[2], v4[3]};
                }
                gelu_poly2_x8(xs8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { pk[j].x = pack2bf(xs8[2 * j][0], xs8[2 * j][1]); pk[j].y = pack2bf(xs8[2 * j + 1][0], xs8[2 * j + 1][1]); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4_t v4 = acc[q & 1][j];
                    pk[j].x = pack2bf(v4[0], v4[1]);
                    pk[j].y = pack2bf(v4[2], v4[3]);
                }
            }
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].x, pk[2 * jp + 1].x, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].y, pk[2 * jp + 1].y, false, false);
                uint4 o = make_uint4(__builtin_amdgcn_ds_bpermute(bperm, r0[0]), __builtin_amdgcn_ds_bpermute(bperm, r1[0]),
                                     __builtin_amdgcn_ds_bpermute(bperm, r0[1]), __builtin_amdgcn_ds_bpermute(bperm, r1[1]));
                if (!all_live && !live_row) o = make_uint4(0, 0, 0, 0);
#if defined(SC_CONV0_NT) && SC_CONV0_NT      // build option for A/B runs: streaming policy on the 8.4 GB output
                typedef unsigned __attribute__((ext_vector_type(4))) u32x4_nt_t;
                __builtin_nontemporal_store((u32x4_nt_t){o.x, o.y, o.z, o.w}, (u32x4_nt_t*)(orow + q * 64 + jp * 32));
#else
                *(uint4*)(orow + q * 64 + jp * 32) = o;
#endif
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- pos-conv
// xg[b][g][Kw/2 + t][c] = t < valid[b] ? x[b][t][g*cg + c] : 0, rows [0,Kw/2) and [Kw/2+Tp, Tp+Kw) zero.
__global__ __launch_bounds__(256) void posconv_pack_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ valid, bf16_t* __restrict__ xg,
                                                           int Tp, int D, int G, int Kw) {
    const int cg = D / G;
    const int rows = Tp + Kw;
    const int b = blockIdx.z, g = blockIdx.y;
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4);  // 16 rows per block, 16 threads per row
    if (r >= rows) return;
    const int t = r - Kw / 2;
    const bool live = t >= 0 && t < Tp && t < valid[b];
    bf16_t* dst = xg + (((int64_t)b * G + g) * rows + r) * cg;
    const bf16_t* src = x + ((int64_t)b * Tp + (live ? t : 0)) * D + g * cg;
    for (int c = (threadIdx.x & 15) * 4; c < cg; c += 64) {
        uint2 v = make_uint2(0u, 0u);
        if (live) v = *(const uint2*)(src + c);
        *(uint2*)(dst + c) = v;
    }
}

// out[b,t,:] = [LN]( mask(x)[b,t,:] + gelu(cg_out[b,g,t,c] + bias) )
// CG: channels per group as a compile-time constant (48: HuBERT-base 768 / 16, 64: HuBERT-large 1024 / 16; 0 = runtime): the per-chunk `e / cg` was a ~25-instruction
// integer division sequence, four times per row, in a kernel that is bound by vector issue (1 140 instructions per 768-element row; round 6).
template <bool OUT_F32, int CG>
__global__ __launch_bounds__(256) void posconv_finish_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ valid, const bf16_t* __restrict__ conv,
                                                             const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             void* __restrict__ out, int B, int Tp, int D, int G, float eps,
                                                             const int32_t* __restrict__ row_off, int64_t total_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    int b, t;
    int64_t conv_row0;          // first conv element row of (utterance b, group 0): conv slab of utterance b = [G][rows_b][cg]
    if (row_off) {              // packed batch: binary search of the owning utterance (row_off[b] <= row < row_off[b + 1])
        int lo = 0, hi = B;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)row_off[mid] <= row) lo = mid; else hi = mid; }
        b = lo; t = (int)(row - row_off[b]); Tp = row_off[b + 1] - row_off[b]; conv_row0 = (int64_t)row_off[b] * G;
    } else { b = (int)((unsigned)row / (unsigned)Tp); t = (int)(row - (int64_t)b * Tp); conv_row0 = (int64_t)b * G * Tp; }      // (rows < 2^31: host check)
    const int cg = CG ? CG : D / G;
    const bool live = t < valid[b];
    // (round 6: 16-byte loads of bias / gamma / beta instead of one dword per element, the GELU as the packed-half polynomial of the GEMM epilogues for all value pairs
    //  of the row side by side -- the result is rounded to bf16 behind the LayerNorm anyway --: ~1 140 -> ~520 vector / scalar instructions per row)
    float v[4][4];
    f32x2_t gp[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = c * 256 + lane * 4;
        gp[2 * c] = gp[2 * c + 1] = (f32x2_t){0.f, 0.f};
        if (e < D) {
            const int g = e / cg, ci = e - g * cg;  // cg % 4 == 0 so the 4 elements stay in one group
            const uint2 cv = *(const uint2*)(conv + (conv_row0 + (int64_t)g * Tp + t) * cg + ci);
            const f32x4_t b4 = *(const f32x4_t*)(bias + e);
            gp[2 * c] = (f32x2_t){lo2f(cv.x) + b4[0], hi2f(cv.x) + b4[1]};
            gp[2 * c + 1] = (f32x2_t){lo2f(cv.y) + b4[2], hi2f(cv.y) + b4[3]};
        }
    }
    gelu_poly2_x8(gp);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = c * 256 + lane * 4;
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
        if (e < D) {
            uint2 xx = make_uint2(0u, 0u);
            if (live) xx = *(const uint2*)(x + row * D + e);
            v[c][0] = lo2f(xx.x) + gp[2 * c][0]; v[c][1] = hi2f(xx.x) + gp[2 * c][1];
            v[c][2] = lo2f(xx.y) + gp[2 * c + 1][0]; v[c][3] = hi2f(xx.y) + gp[2 * c + 1][1];
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (gamma) {
        mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = c * 256 + lane * 4;
            if (e < D) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { float d = v[c][i] - mean; q += d * d; }
            }
        }
        rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = c * 256 + lane * 4;
        if (e < D) {
            f32x4_t o = {v[c][0], v[c][1], v[c][2], v[c][3]};
            if (gamma) {
                const f32x4_t g4 = *(const f32x4_t*)(gamma + e), be4 = *(const f32x4_t*)(beta + e);
                o = (o - mean) * rstd * g4 + be4;
            }
            if (OUT_F32) *(f32x4_t*)((float*)out + row * D + e) = o;
            else { uint2 p; p.x = pack2bf(o[0], o[1]); p.y = pack2bf(o[2], o[3]); *(uint2*)((bf16_t*)out + row * D + e) = p; }
        }
    }
}

// -------------------------------------------------------------------------------------------- ViT stem
// cols[b*np + gy*g + gx][c*p*p + py*p + px] = img[b][c][gy*p+py][gx*p+px]; columns [3*p*p, Kpad) zero.
__global__ __launch_bounds__(256) void vit_patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ cols, int R, int p, int Kpad) {
    const int g = R / p, np = g * g;
    const int row = blockIdx.x, b = row / np, pi = row - b * np, gy = pi / g, gx = pi - gy * g;
    const int K = 3 * p * p;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        float val = 0.f;
        if (k < K) {
            const int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
            val = img[(((int64_t)b * 3 + c) * R + gy * p + py) * R + gx * p + px];
        }
        cols[(int64_t)row * Kpad + k] = f2bf(val);
    }
}

// x0[b][tk][:] = ln_pre( (tk == 0 ? class_emb : patch[b*np + tk - 1]) + pos[tk] )   (f32 residual stream)
__global__ __launch_bounds__(256) void vit_embed_kernel(const bf16_t* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
                                                        int64_t rows, int ntok, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t b = row / ntok;
    const int tk = (int)(row - b * ntok);
    float v[4][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = c * 256 + lane * 4;
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
        if (e < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float base = tk == 0 ? cls[e + i] : bf2f(patch[(b * (ntok - 1) + tk - 1) * D + e + i]);
                v[c][i] = base + pos[(int64_t)tk * D + e + i];
            }
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = c * 256 + lane * 4;
        if (e < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { float d = v[c][i] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = c * 256 + lane * 4;
        if (e < D)
            *(f32x4_t*)(out + row * D + e) = (f32x4_t){(v[c][0] - mean) * rstd * gamma[e] + beta[e], (v[c][1] - mean) * rstd * gamma[e + 1] + beta[e + 1],
                                                        (v[c][2] - mean) * rstd * gamma[e + 2] + beta[e + 2], (v[c][3] - mean) * rstd * gamma[e + 3] + beta[e + 3]};
    }
}

}  // namespace

extern "C" int64_t sc_conv0_stats_workspace_bytes(int B) { return (int64_t)B * NSPLIT * NSTAT * 8; }

extern "C" int sc_conv0_gn_coef(const float* wav, int64_t ld, const float* w, const float* gamma, const float* beta, void* workspace,
                                float* coef, int B, int C, int T0, float eps, void* stream) {
    SC_CHECK_ARG(B > 0 && C > 0 && T0 > 0, "sc_conv0_gn_coef: bad sizes B=%d C=%d T0=%d", B, C, T0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv0_stats_kernel, dim3(B, NSPLIT), dim3(256), 0, s, wav, ld, T0, (double*)workspace);
    SC_CHECK_LAUNCH();
    hipLaunchKernelGGL(conv0_coef_kernel, dim3(B), dim3(256), 0, s, (const double*)workspace, w, gamma, beta, (float2*)coef, C, T0, eps);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_conv0_wfrag_workspace_bytes(int B) { return (int64_t)B * 32 * 64 * 16; }

static int conv0_fwd_impl(const float* wav, int64_t ld, int64_t L, const float* w, const float* bias, const float* coef, void* out, int B,
                          int C, int T0, int P, int mode, void* wfrag_ws, void* stream, const int32_t* row_off, int row_scale) {
    SC_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= 512, "sc_conv0_fwd: C=%d must be a multiple of 8, <= 512", C);
    SC_CHECK_ARG(row_off == nullptr || (C % 64 == 0 && P % 64 == 0 && row_scale % 64 == 0), "sc_conv0_fwd_packed: needs C %% 64 == 0 and row_scale %% 64 == 0");
    SC_CHECK_ARG(mode == 0 || mode == 1 || mode == 2, "sc_conv0_fwd: mode=%d must be 0 (GroupNorm + GELU), 1 (conv + bias) or 2 (conv + bias + LayerNorm + GELU)", mode);
    SC_CHECK_ARG(mode == 1 || coef != nullptr, "sc_conv0_fwd: GroupNorm mode needs coef, LayerNorm mode gamma | beta | eps");
    SC_CHECK_ARG(mode != 2 || (C % 64 == 0 && P % 64 == 0), "sc_conv0_fwd: the LayerNorm mode exists in the matrix-core form only (C %% 64 == 0, P %% 64 == 0)");
    SC_CHECK_ARG((P >= T0 || row_off) && B > 0 && B <= 65535, "sc_conv0_fwd: need P >= T0 and 0 < B <= 65535");
    static const bool force_valu = getenv("SC_CONV0_VALU") != nullptr;
    if (C % 64 == 0 && P % 64 == 0 && (!force_valu || row_off || mode == 2)) {      // matrix-core form (every shipped config: C = 512)
        SC_CHECK_ARG(wfrag_ws != nullptr, "sc_conv0_fwd: the matrix-core form needs the W-fragment workspace (sc_conv0_wfrag_workspace_bytes)");
        hipLaunchKernelGGL(conv0_wfrag_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, w, (const float2*)coef, bias, (bf16x8_t*)wfrag_ws, C, mode);
        SC_CHECK_LAUNCH();
        static const int fb = getenv("SC_CONV0_FB") ? atoi(getenv("SC_CONV0_FB")) : SC_CONV0_FB_DEFAULT;   // frames per block
        static const int thr = getenv("SC_CONV0_THREADS") ? atoi(getenv("SC_CONV0_THREADS")) : SC_CONV0_THREADS_DEFAULT;   // 256 or 512 threads per block
#define CONV0_LAUNCH(FB_) do {                                                                                                                   \
        const int lds = 32 * 64 * 16 + (1024 + FB_ * CS + 16) * 4;                                                                                  \
        (void)hipFuncSetAttribute((const void*)conv0_mfma_kernel<FB_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                           \
        hipLaunchKernelGGL(conv0_mfma_kernel<FB_>, dim3((P + FB_ - 1) / FB_, B), dim3(thr == 512 ? 512 : 256), lds, (hipStream_t)stream, wav, ld, L,                  \
                           (const bf16x8_t*)wfrag_ws, bias, (const float2*)coef, (bf16_t*)out, C, T0, P, mode, row_off, row_scale); } while (0)
        if (fb == 128) CONV0_LAUNCH(128); else if (fb == 512) CONV0_LAUNCH(512); else if (fb == 1024) CONV0_LAUNCH(1024); else CONV0_LAUNCH(256);
#undef CONV0_LAUNCH
    } else {
        hipLaunchKernelGGL(conv0_fwd_kernel, dim3((P + TT - 1) / TT, B), dim3(256), 0, (hipStream_t)stream, wav, ld, L, w, bias, (const float2*)coef,
                           (bf16_t*)out, C, T0, P, mode);
    }
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_conv0_fwd(const float* wav, int64_t ld, int64_t L, const float* w, const float* bias, const float* coef, void* out, int B,
                            int C, int T0, int P, int mode, void* wfrag_ws, void* stream) {
    return conv0_fwd_impl(wav, ld, L, w, bias, coef, out, B, C, T0, P, mode, wfrag_ws, stream, nullptr, 0);
}

// Packed (padding-free) batches: utterance b writes rows [row_scale * row_off[b], row_scale * row_off[b + 1]) of `out` (row_off: B + 1 device
// ints in transformer frames; row_scale = product of the strides of conv layers 1.., 64 for HuBERT; Pmax = row_scale * max rows per utterance).
// The GroupNorm statistics (`coef`) are those of the PADDED length T0: zero samples contribute nothing to sum / sum of squares, the divisor stays T0.
extern "C" int sc_conv0_fwd_packed(const float* wav, int64_t ld, int64_t L, const float* w, const float* bias, const float* coef, void* out, int B,
                                   int C, int T0, const int32_t* row_off, int row_scale, int Pmax, int mode, void* wfrag_ws, void* stream) {
    SC_CHECK_ARG(row_off != nullptr && row_scale > 0 && Pmax > 0, "sc_conv0_fwd_packed: row_off, row_scale and Pmax are required");
    return conv0_fwd_impl(wav, ld, L, w, bias, coef, out, B, C, T0, Pmax, mode, wfrag_ws, stream, row_off, row_scale);
}

extern "C" int sc_posconv_pack(const void* x, const int32_t* valid, void* xg, int B, int Tp, int D, int G, int Kw, void* stream) {
    SC_CHECK_ARG(D % G == 0 && (D / G) % 4 == 0, "sc_posconv_pack: D/G must be a multiple of 4");
    SC_CHECK_ARG(B <= 65535 && G <= 65535, "sc_posconv_pack: grid limits");
    hipLaunchKernelGGL(posconv_pack_kernel, dim3((Tp + Kw + 15) / 16, G, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, valid, (bf16_t*)xg, Tp,
                       D, G, Kw);
    SC_CHECK_LAUNCH();
    return 0;
}

static int posconv_finish_impl(const void* x, const int32_t* valid, const void* conv, const float* bias, const float* gamma, const float* beta,
                               void* out, int B, int Tp, int D, int G, int out_f32, float eps, void* stream, const int32_t* row_off, int64_t rows) {
    SC_CHECK_ARG(D <= 1024 && D % G == 0 && (D / G) % 4 == 0, "sc_posconv_finish: D<=1024 and D/G multiple of 4 required");
    SC_CHECK_ARG(rows > 0 && rows < 0x7fffffffLL, "sc_posconv_finish: rows=%lld out of range", (long long)rows);
    dim3 grid((unsigned)((rows + 3) / 4));
    const int cgv = D / G;
#define PF_LAUNCH(F32_, CG_) hipLaunchKernelGGL((posconv_finish_kernel<F32_, CG_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, valid, (const bf16_t*)conv, bias, gamma, beta, out, B, Tp, D, G, eps, row_off, rows)
    if (out_f32) { if (cgv == 64) PF_LAUNCH(true, 64); else if (cgv == 48) PF_LAUNCH(true, 48); else PF_LAUNCH(true, 0); }
    else { if (cgv == 64) PF_LAUNCH(false, 64); else if (cgv == 48) PF_LAUNCH(false, 48); else PF_LAUNCH(false, 0); }
#undef PF_LAUNCH
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_posconv_finish(const void* x, const int32_t* valid, const void* conv, const float* bias, const float* gamma, const float* beta,
                                 void* out, int B, int Tp, int D, int G, int out_f32, float eps, void* stream) {
    return posconv_finish_impl(x, valid, conv, bias, gamma, beta, out, B, Tp, D, G, out_f32, eps, stream, nullptr, (int64_t)B * Tp);
}

// Packed batches (see sc_posconv_conv_packed): rows [row_off[b], row_off[b + 1]) belong to utterance b, total_rows = row_off[B].
extern "C" int sc_posconv_finish_packed(const void* x, const int32_t* valid, const int32_t* row_off, const void* conv, const float* bias, const float* gamma,
                                        const float* beta, void* out, int B, int64_t total_rows, int D, int G, int out_f32, float eps, void* stream) {
    SC_CHECK_ARG(row_off != nullptr && valid != nullptr && total_rows > 0, "sc_posconv_finish_packed: row_off, valid and total_rows are required");
    return posconv_finish_impl(x, valid, conv, bias, gamma, beta, out, B, 0, D, G, out_f32, eps, stream, row_off, total_rows);
}

extern "C" int sc_vit_patchify(const float* img, void* cols, int B, int R, int p, int Kpad, void* stream) {
    SC_CHECK_ARG(R % p == 0 && Kpad >= 3 * p * p && Kpad % 64 == 0, "sc_vit_patchify: R%%p==0 and Kpad>=3p^2, Kpad%%64==0 required");
    const int np = (R / p) * (R / p);
    hipLaunchKernelGGL(vit_patchify_kernel, dim3(B * np), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)cols, R, p, Kpad);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_vit_embed(const void* patch, const float* cls, const float* pos, const float* gamma, const float* beta, float* out, int B,
                            int ntok, int D, float eps, void* stream) {
    SC_CHECK_ARG(D <= 1024 && D % 4 == 0, "sc_vit_embed: D=%d must be a multiple of 4, <= 1024", D);
    const int64_t rows = (int64_t)B * ntok;
    hipLaunchKernelGGL(vit_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)patch, cls, pos, gamma, beta,
                       out, rows, ntok, D, eps);
    SC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ image side of the collate path
// out[b][c][y][x] = (u8[b][y][x][c] / 255 - mean[c]) / std[c]: torchvision ToTensor + Normalize of CLIP's `_transform` (the PIL resize /
// centre crop before it stay on the host) -- the uint8 crop crosses PCIe (150 KB per image instead of 602 KB of fp32).
namespace {
__global__ __launch_bounds__(256) void image_normalize_kernel(const uint8_t* __restrict__ u8, float* __restrict__ out, int64_t npix_per_img, float m0, float m1,
                                                              float m2, float is0, float is1, float is2, int64_t total_pix) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // pixel index over the batch
    if (i >= total_pix) return;
    const int64_t b = i / npix_per_img, r = i - b * npix_per_img;
    const uint8_t* px = u8 + i * 3;
    float* o = out + b * 3 * npix_per_img + r;
    o[0] = ((float)px[0] / 255.0f - m0) * is0;
    o[npix_per_img] = ((float)px[1] / 255.0f - m1) * is1;
    o[2 * npix_per_img] = ((float)px[2] / 255.0f - m2) * is2;
}
}  // namespace

extern "C" int sc_image_normalize_u8(const void* u8_hwc, float* out_chw, int B, int H, int W, const float* mean3, const float* std3, void* stream) {
    SC_CHECK_ARG(B >= 0 && H > 0 && W > 0 && mean3 && std3, "sc_image_normalize_u8: bad arguments");
    SC_CHECK_ARG(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "sc_image_normalize_u8: std must be positive");
    const int64_t npix = (int64_t)H * W, total = npix * B;
    if (total == 0) return 0;
    hipLaunchKernelGGL(image_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)u8_hwc, out_chw, npix,
                       mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2], total);
    SC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ K1: crop + right-pad a batch of waves
// out[b, j] = j < lens[b] ? wav[b, starts[b] + j] : 0   (train-mode random crop to max_audio_len, audio_transforms.py:5-23, then the
// zero right-padding of preprocess_input, speech_encoder_plus.py:510-518) -- one launch instead of one slice copy per utterance.
namespace {
__global__ void crop_pad_kernel(const float* __restrict__ wav, int64_t ld, const int32_t* __restrict__ starts, const int32_t* __restrict__ lens,
                                float* __restrict__ out, int Lout) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Lout) return;
    out[(int64_t)b * Lout + j] = j < lens[b] ? wav[(int64_t)b * ld + starts[b] + j] : 0.f;
}
}  // namespace

extern "C" int sc_crop_pad(const float* wav, int64_t ld, const int32_t* starts, const int32_t* lens, float* out, int B, int Lout, void* stream) {
    SC_CHECK_ARG(B > 0 && Lout > 0 && wav && starts && lens && out, "sc_crop_pad: bad arguments");
    hipLaunchKernelGGL(crop_pad_kernel, dim3((Lout + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, wav, ld, starts, lens, out, Lout);
    SC_CHECK_LAUNCH();
    return 0;
}
