// Backward pieces for fine-tuning HuBERT transformer layers (reference: speech_encoder_plus.py:416-446
// `trainable` / `reinit_layers` / `unfreeze_layers` -- the selected fairseq TransformerSentenceEncoderLayers train, the feature extractor,
// positional conv and projections stay frozen).  The dense products run on the MFMA GEMM of gemm.hip (dX = dY W on transposed weight copies;
// dW = dY^T X as a split-K batched GEMM over transposed activations); this file holds what the GEMM cannot do:
//   sc_transpose_bf16        batched, strided [R, C] -> [C, Rpad] transposes (operands of the TN products, zero padded to the GEMM's K rule)
//   sc_attn_softmax_bwd      rows of P = softmax(scale S + key mask) and dS = scale P (dP - rowsum(dO . O)) from S = Q K^T, dP = dO V^T
//   sc_gelu_bwd_bf16         du = dh gelu'(u) (exact erf form)
//   sc_layernorm_bwd_bf16    dx from (x, dy, gamma) per row + per-chunk partial column sums for dgamma / dbeta
//   sc_colsum_bf16           bias gradients: column sums of a bf16 [rows, cols] matrix (two-stage, deterministic)
//   sc_axpy_bf16             y += alpha x (the layer mix's share of a hidden state's gradient)
//   sc_cls_pool_dz           gradient of the mixed frames out of the pooling head's backward workspaces (sc_cls_pool_bwd keeps it in registers)
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int64_t stride_in, bf16_t* __restrict__ out,
                                                             int64_t ld_out, int64_t stride_out, int R, int C, int Rpad) {
    __shared__ bf16_t tile[64][66];
    const int z = blockIdx.z, r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const bf16_t* src = in + (int64_t)z * stride_in;
    bf16_t* dst = out + (int64_t)z * stride_out;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 4 row groups of 16
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty * 16 + k, c = c0 + tx;
        tile[ty * 16 + k][tx] = (r < R && c < C) ? src[(int64_t)r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty * 16 + k, r = r0 + tx;
        if (c < C && r < Rpad) dst[(int64_t)c * ld_out + r] = tile[tx][ty * 16 + k];
    }
}

// The same with 8-byte global accesses (4 elements per lane along the contiguous axis on both sides) for the aligned case: cols, ld_in, ld_out,
// rows_padded, the strides all multiples of 4 and 8-byte aligned bases.  A 64 x 64 tile: 16 lanes x 4 elements per row on the way in, per column on
// the way out.
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int64_t stride_in, bf16_t* __restrict__ out,
                                                                 int64_t ld_out, int64_t stride_out, int R, int C, int Rpad) {
    __shared__ bf16_t tile[64][68];
    const int z = blockIdx.z, r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const bf16_t* src = in + (int64_t)z * stride_in;
    bf16_t* dst = out + (int64_t)z * stride_out;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;            // 16 lanes x 4 elements = 64 contiguous, 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + k * 16 + rr, c = c0 + q * 4;
        uint2 v = make_uint2(0u, 0u);
        if (r < R && c < C) v = *(const uint2*)(src + (int64_t)r * ld_in + c);      // C % 4 == 0: a 4-group is entirely inside or outside
        *(uint2*)&tile[k * 16 + rr][q * 4] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k * 16 + rr, r = r0 + q * 4;
        if (c < C && r < Rpad) {
            const int cl = k * 16 + rr;
            uint2 o;
            o.x = (uint32_t)tile[q * 4 + 0][cl] | ((uint32_t)tile[q * 4 + 1][cl] << 16);
            o.y = (uint32_t)tile[q * 4 + 2][cl] | ((uint32_t)tile[q * 4 + 3][cl] << 16);
            *(uint2*)(dst + (int64_t)c * ld_out + r) = o;
        }
    }
}

// one wave per (batch z, query row i): S / dP rows f32 [L] (row stride ld), keys j >= klen[z] masked
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ S, const float* __restrict__ dP, int64_t ld, int64_t stride,
                                                               const bf16_t* __restrict__ dO, int64_t ld_do, const bf16_t* __restrict__ O, int64_t ld_o,
                                                               int64_t row_stride_z, const int32_t* __restrict__ klens, bf16_t* __restrict__ P,
                                                               bf16_t* __restrict__ dS, int L, int Lp, float scale, uint32_t drop_seed,
                                                               uint32_t drop_thresh_, float keep_scale, uint32_t elem_base, uint32_t elem_stride_z, int heads) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), z = blockIdx.y;
    if (i >= Lp) return;
    bf16_t* prow = P + (int64_t)z * stride + (int64_t)i * ld;
    bf16_t* drow = dS + (int64_t)z * stride + (int64_t)i * ld;
    // heads > 0: z = b * heads + h over ALL heads (dO / O point at column 0, rows of utterance b, head h at column h * 64); else z = utterance
    const int zb = heads > 0 ? z / heads : z, zh = heads > 0 ? z - zb * heads : 0;
    const int klen = klens ? klens[zb] : L;
    if (i >= L) {                                                       // padding rows of the [Lp, Lp] images: zeros (they are K-dim padding of the TN products)
        for (int j = lane; j < Lp; j += 64) { prow[j] = 0; drow[j] = 0; }
        return;
    }
    const float* srow = S + (int64_t)z * stride + (int64_t)i * ld;
    const float* gprow = dP + (int64_t)z * stride + (int64_t)i * ld;
    // D_i = dO_i . O_i over the 64 head dims (one per lane)
    const int64_t r = (int64_t)zb * row_stride_z + i;
    const float dd = wave_sum(bf2f(dO[r * ld_do + zh * 64 + lane]) * bf2f(O[r * ld_o + zh * 64 + lane]));
    float mx = -INFINITY;
    for (int j = lane; j < klen; j += 64) mx = fmaxf(mx, srow[j] * scale);
    mx = wave_max(mx);
    float den = 0.f;
    for (int j = lane; j < klen; j += 64) den += __expf(srow[j] * scale - mx);
    den = wave_sum(den);
    const float inv = den > 0.f ? 1.0f / den : 0.f;
    for (int j = lane; j < Lp; j += 64) {
        float p = 0.f, ds = 0.f;
        if (j < klen) {
            p = __expf(srow[j] * scale - mx) * inv;
            float gp = gprow[j];
            if (drop_thresh_) {      // attention dropout of the forward (sc_attention_fwd_dropout): the same mask -- one hash per pair of adjacent keys,
                // 16 bits each (common.h hash_pair).  dP = m dP_dropped, P_dropped = m P (what dV = P_dropped^T dO multiplies); sum_k P_ik dP_ik = dO_i . O_i still holds
                const uint32_t hb = hash_pair(drop_seed, (elem_base + (uint32_t)z * elem_stride_z + (uint32_t)i) * (uint32_t)((L + 1) >> 1) + ((uint32_t)j >> 1));
                const float m = ((j & 1) ? (hb >> 16) : (hb & 0xffffu)) >= drop_thresh_ ? keep_scale : 0.f;
                gp *= m;
                ds = p * (gp - dd) * scale;
                p *= m;
            } else
            ds = p * (gp - dd) * scale;
        }
        prow[j] = f2bf(p);
        drow[j] = f2bf(ds);
    }
}

// Fused front half of the attention backward for head dim 64: recomputes S = Q K^T and dP = dO V^T on the matrix cores straight from the packed rows
// (every MFMA fragment is ONE 16-byte global load: both products contract along the contiguous head dimension, so no LDS staging is needed) and writes
// P (dropped, if the forward dropped) and dS = scale P (m dP - dO.O) as bf16 [B*H, Lp, Lp] images -- the fp32 S / dP images of the unfused path
// (two batched GEMMs + sc_attn_softmax_bwd_heads: 6 GB of traffic per layer at B = 256) never exist.  One wave per 16 queries of one (b, h);
// pass 1 runs the online max / sum over the key blocks, pass 2 recomputes each 16 x 16 block and emits it.
// MFMA 16x16x32: first operand = key rows, second = query rows -> lane l holds query (l & 15), keys 4 (l >> 4) + r of the block.
__global__ __launch_bounds__(256) void attn_bwd_probs_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                             int64_t ld_qkv, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, int64_t ld_o,
                                                             const int32_t* __restrict__ klens, bf16_t* __restrict__ P, bf16_t* __restrict__ dS, int H, int L,
                                                             int Lp, float scale, uint32_t drop_seed, uint32_t drop_thresh_, float keep_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.y, b = z / H, h = z - b * H;
    const int i0 = (blockIdx.x * 4 + wave) * 16;                       // this wave's 16 queries
    if (i0 >= Lp) return;
    const int qi = lane & 15, fk = lane >> 4;
    const int i = i0 + qi;
    bf16_t* prow = P + ((int64_t)z * Lp + i) * Lp;
    bf16_t* drow = dS + ((int64_t)z * Lp + i) * Lp;
    const int klen = klens ? min(klens[b], L) : L;
    const int nkb = (klen + 15) / 16;                                   // key blocks that hold at least one valid key
    const int ic = i < L ? i : L - 1;                                   // rows >= L: clamp the loads, write zeros
    const int64_t row = (int64_t)b * L + ic;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    // fragments of this lane's query row: k-slots 8 fk .. 8 fk + 7 of each 32-wide half of the 64 head dims
    bf16x8_t qf[2], dof[2];
    float dpart = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        qf[c] = *(const bf16x8_t*)(q + row * ld_qkv + h * 64 + c * 32 + fk * 8);
        const uint4 du = *(const uint4*)(dO + row * ld_o + h * 64 + c * 32 + fk * 8);      // raw bits: the dot below reads them as bf16 pairs
        const uint4 ou = *(const uint4*)(O + row * ld_o + h * 64 + c * 32 + fk * 8);
        dof[c] = __builtin_bit_cast(bf16x8_t, du);
        dpart += lo2f(du.x) * lo2f(ou.x) + hi2f(du.x) * hi2f(ou.x) + lo2f(du.y) * lo2f(ou.y) + hi2f(du.y) * hi2f(ou.y)
               + lo2f(du.z) * lo2f(ou.z) + hi2f(du.z) * hi2f(ou.z) + lo2f(du.w) * lo2f(ou.w) + hi2f(du.w) * hi2f(ou.w);
    }
    dpart += __shfl_xor(dpart, 16, 64);
    dpart += __shfl_xor(dpart, 32, 64);                                 // D_i = dO_i . O_i (all 64 dims)
    auto kfrag = [&](const bf16_t* base, int kb, int c) -> bf16x8_t {  // row (16 kb + qi) of K or V, clamped to the utterance
        int key = kb * 16 + qi;
        key = key < L ? key : L - 1;
        return *(const bf16x8_t*)(base + ((int64_t)b * L + key) * ld_qkv + h * 64 + c * 32 + fk * 8);
    };
    // ---- pass 1: row maximum and sum of exp over the valid keys
    float m = -INFINITY, l = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag(k, kb, c), qf[c], sacc, 0, 0, 0);
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb * 16 + fk * 4 + r;
            sacc[r] = key < klen ? sacc[r] * scale : -INFINITY;
            bm = fmaxf(bm, sacc[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) ps += __expf(sacc[r] - mn);
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l = l * __expf(m - mn) + ps;
        m = mn;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const uint32_t row_id = (uint32_t)((b * H + h) * L + ic), pairs = (uint32_t)((L + 1) >> 1);
    // ---- pass 2: every 16 x 16 block again, with dP, and out
    const int nkb_all = Lp / 16;
    for (int kb = 0; kb < nkb_all; ++kb) {
        uint2 po = make_uint2(0u, 0u), so = make_uint2(0u, 0u);
        if (kb < nkb) {          // wave-uniform: every lane feeds the MFMAs (its lane index is also the KEY row of the K / V fragments); rows >= L are zeroed below
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag(k, kb, c), qf[c], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag(v, kb, c), dof[c], pacc, 0, 0, 0);
            }
            float pv[4], dv[4];
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const int key = kb * 16 + fk * 4 + r;
                float m0 = 1.f, m1 = 1.f;
                if (drop_thresh_) {
                    const uint32_t hb = hash_pair(drop_seed, row_id * pairs + ((uint32_t)key >> 1));
                    m0 = (hb & 0xffffu) >= drop_thresh_ ? keep_scale : 0.f;
                    m1 = (hb >> 16) >= drop_thresh_ ? keep_scale : 0.f;
                }
                const float p0 = key < klen ? __expf(sacc[r] * scale - m) * inv : 0.f;
                const float p1 = key + 1 < klen ? __expf(sacc[r + 1] * scale - m) * inv : 0.f;
                dv[r] = p0 * (pacc[r] * m0 - dpart) * scale;
                dv[r + 1] = p1 * (pacc[r + 1] * m1 - dpart) * scale;
                pv[r] = p0 * m0;
                pv[r + 1] = p1 * m1;
            }
            if (i < L) {
                po.x = pack2bf(pv[0], pv[1]); po.y = pack2bf(pv[2], pv[3]);
                so.x = pack2bf(dv[0], dv[1]); so.y = pack2bf(dv[2], dv[3]);
            }
        }
        *(uint2*)(prow + kb * 16 + fk * 4) = po;
        *(uint2*)(drow + kb * 16 + fk * 4) = so;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_bf16_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ dh, bf16_t* __restrict__ du, int64_t n) {
    auto d = [](float x) { return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x); };
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;      // 16 bytes per lane; the tail (n % 8, n even) goes pair by pair
    if (i >= n) return;
    if (i + 8 <= n) {
        const uint4 uu = *(const uint4*)(u + i), gg = *(const uint4*)(dh + i);
        uint4 o;
        o.x = pack2bf(lo2f(gg.x) * d(lo2f(uu.x)), hi2f(gg.x) * d(hi2f(uu.x)));
        o.y = pack2bf(lo2f(gg.y) * d(lo2f(uu.y)), hi2f(gg.y) * d(hi2f(uu.y)));
        o.z = pack2bf(lo2f(gg.z) * d(lo2f(uu.z)), hi2f(gg.z) * d(hi2f(uu.z)));
        o.w = pack2bf(lo2f(gg.w) * d(lo2f(uu.w)), hi2f(gg.w) * d(hi2f(uu.w)));
        *(uint4*)(du + i) = o;
        return;
    }
    for (int64_t j = i; j < n; j += 2) {
        const uint32_t uu = *(const uint32_t*)(u + j), gg = *(const uint32_t*)(dh + j);
        *(uint32_t*)(du + j) = pack2bf(lo2f(gg) * d(lo2f(uu)), hi2f(gg) * d(hi2f(uu)));
    }
}

// LayerNorm backward, bf16 rows: dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)); per-chunk partial sums of dy xhat (dgamma) and dy (dbeta).
// One wave per row, a block (4 waves) owns a chunk of `rows_per_block` consecutive rows and writes ONE partial row: part[chunk][0][D], [1][D].
__global__ __launch_bounds__(256) void ln_bwd_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ gamma,
                                                          bf16_t* __restrict__ dx, float* __restrict__ part, int64_t rows, int D, float eps,
                                                          int rows_per_block) {
    __shared__ float sg[4][1024], sb[4][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_block;
    const int nd = D / 64;                                             // D % 64 == 0, <= 1024
    float ag[16], ab[16], g[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { ag[m] = 0.f; ab[m] = 0.f; g[m] = (m < nd) ? gamma[lane + 64 * m] : 0.f; }
    for (int64_t row = row_begin + w; row < row_begin + rows_per_block && row < rows; row += 4) {
        float xv[16], gv[16];
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) { xv[m] = (m < nd) ? bf2f(x[row * D + lane + 64 * m]) : 0.f; s += xv[m]; }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) if (m < nd) { const float d = xv[m] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) if (m < nd) {
            const float dyv = bf2f(dy[row * D + lane + 64 * m]);
            const float xh = (xv[m] - mean) * rstd;
            gv[m] = dyv * g[m];
            s1 += gv[m]; s2 += gv[m] * xh;
            ag[m] += dyv * xh; ab[m] += dyv;
            xv[m] = xh;
        }
        s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int m = 0; m < 16; ++m) if (m < nd) dx[row * D + lane + 64 * m] = f2bf(rstd * (gv[m] - s1 - xv[m] * s2));
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) if (m < nd) { sg[w][lane + 64 * m] = ag[m]; sb[w][lane + 64 * m] = ab[m]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        part[((int64_t)blockIdx.x * 2 + 0) * D + c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
        part[((int64_t)blockIdx.x * 2 + 1) * D + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
    }
}

// The same for D % 256 == 0 (512 / 768 / 1024): a lane owns 4 consecutive elements of every 256-element group (8-byte loads and stores instead of
// 2-byte ones) -- the scalar form above moved 0.56 TB/s on [82 k, 768], this one is HBM-bound.
__global__ __launch_bounds__(256) void ln_bwd_bf16_vec_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ gamma,
                                                              bf16_t* __restrict__ dx, float* __restrict__ part, int64_t rows, int D, float eps,
                                                              int rows_per_block) {
    __shared__ float sg[4][1024], sb[4][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row_begin = (int64_t)blockIdx.x * rows_per_block;
    const int ng = D / 256;                                            // <= 4
    float ag[4][4], ab[4][4], g[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[m][j] = 0.f; ab[m][j] = 0.f; g[m][j] = (m < ng) ? gamma[m * 256 + lane * 4 + j] : 0.f; }
    for (int64_t row = row_begin + w; row < row_begin + rows_per_block && row < rows; row += 4) {
        float xv[4][4], gv[4][4];
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (m < ng) {
            const uint2 u = *(const uint2*)(x + row * D + m * 256 + lane * 4);
            xv[m][0] = lo2f(u.x); xv[m][1] = hi2f(u.x); xv[m][2] = lo2f(u.y); xv[m][3] = hi2f(u.y);
            s += (xv[m][0] + xv[m][1]) + (xv[m][2] + xv[m][3]);
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (m < ng)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = xv[m][j] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (m < ng) {
            const uint2 u = *(const uint2*)(dy + row * D + m * 256 + lane * 4);
            const float dv[4] = {lo2f(u.x), hi2f(u.x), lo2f(u.y), hi2f(u.y)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[m][j] - mean) * rstd;
                gv[m][j] = dv[j] * g[m][j];
                s1 += gv[m][j]; s2 += gv[m][j] * xh;
                ag[m][j] += dv[j] * xh; ab[m][j] += dv[j];
                xv[m][j] = xh;
            }
        }
        s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (m < ng) {
            uint2 o;
            o.x = pack2bf(rstd * (gv[m][0] - s1 - xv[m][0] * s2), rstd * (gv[m][1] - s1 - xv[m][1] * s2));
            o.y = pack2bf(rstd * (gv[m][2] - s1 - xv[m][2] * s2), rstd * (gv[m][3] - s1 - xv[m][3] * s2));
            *(uint2*)(dx + row * D + m * 256 + lane * 4) = o;
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) if (m < ng)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sg[w][m * 256 + lane * 4 + j] = ag[m][j]; sb[w][m * 256 + lane * 4 + j] = ab[m][j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        part[((int64_t)blockIdx.x * 2 + 0) * D + c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
        part[((int64_t)blockIdx.x * 2 + 1) * D + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
    }
}

__global__ __launch_bounds__(256) void colsum_bf16_stage1_kernel(const bf16_t* __restrict__ x, int64_t ld, int64_t rows, int cols, float* __restrict__ part,
                                                                 int rows_per_block) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {                                       // four independent loads in flight per thread
        s0 += bf2f(x[r * ld + c]); s1 += bf2f(x[(r + 1) * ld + c]); s2 += bf2f(x[(r + 2) * ld + c]); s3 += bf2f(x[(r + 3) * ld + c]);
    }
    for (; r < r1; ++r) s0 += bf2f(x[r * ld + c]);
    part[(int64_t)blockIdx.x * cols + c] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ part, int nparts, int cols, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    // few blocks (cols / 256), many partial rows: keep eight loads in flight per thread (fixed summation order: deterministic)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 8 <= nparts; p += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += part[(int64_t)(p + j) * cols + c];
    }
    for (; p < nparts; ++p) a[0] += part[(int64_t)p * cols + c];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    out[c] = accumulate ? out[c] + s : s;
}

__global__ __launch_bounds__(256) void axpy_bf16_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x, float alpha, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    const uint32_t yy = *(const uint32_t*)(y + i), xx = *(const uint32_t*)(x + i);
    *(uint32_t*)(y + i) = pack2bf(lo2f(yy) + alpha * lo2f(xx), hi2f(yy) + alpha * hi2f(xx));
}

// dz[b, t, :] = sum_r pp[b, r, NQ + t] dzbar[b, r, :] + ds[b, r, NQ + t] u[r, :]   for t < lens[b], else 0  (train.hip's frame-level algebra)
__global__ __launch_bounds__(256) void cls_pool_dz_kernel(const float* __restrict__ pp, const float* __restrict__ ds, const float* __restrict__ dzbar,
                                                          const float* __restrict__ u, const int32_t* __restrict__ lens, bf16_t* __restrict__ dz, int T, int NQ,
                                                          int R, int D, int64_t ld_dz) {
    const int b = blockIdx.y, t = blockIdx.x;
    bf16_t* out = dz + ((int64_t)b * T + t) * ld_dz;
    if (t >= lens[b]) {
        for (int d = threadIdx.x; d < D; d += 256) out[d] = 0;
        return;
    }
    const int Lk = NQ + T;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) {
            const int64_t pi = ((int64_t)b * R + r) * Lk + NQ + t;
            acc += pp[pi] * dzbar[((int64_t)b * R + r) * D + d] + ds[pi] * u[(int64_t)r * D + d];
        }
        out[d] = f2bf(acc);
    }
}

}  // namespace

extern "C" int sc_transpose_bf16(const void* in, int64_t ld_in, int64_t stride_in, void* out, int64_t ld_out, int64_t stride_out, int rows, int cols,
                                 int rows_padded, int batch, void* stream) {
    SC_CHECK_ARG(in && out && rows > 0 && cols > 0 && rows_padded >= rows && batch > 0 && batch <= 65535, "sc_transpose_bf16: bad arguments");
    // ld_in < cols is allowed: an overlapping-row (sliding-window) view of the input, read-only
    SC_CHECK_ARG(ld_out >= rows_padded && ld_in >= 1, "sc_transpose_bf16: leading dimensions too small");
    dim3 grid((rows_padded + 63) / 64, (cols + 63) / 64, batch);
    const bool vec = cols % 4 == 0 && rows_padded % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && stride_in % 4 == 0 && stride_out % 4 == 0 &&
                     (((uintptr_t)in | (uintptr_t)out) & 7) == 0;
    if (vec) {
        hipLaunchKernelGGL(transpose_bf16_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, stride_in, (bf16_t*)out, ld_out,
                           stride_out, rows, cols, rows_padded);
        SC_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, stride_in, (bf16_t*)out, ld_out, stride_out,
                       rows, cols, rows_padded);
    SC_CHECK_LAUNCH();
    return 0;
}

static int attn_softmax_bwd_impl(const float* S, const float* dP, int64_t ld, int64_t stride, const void* dO, int64_t ld_do, const void* O, int64_t ld_o,
                                 int64_t rows_per_batch, const int32_t* klens, void* P, void* dS, int L, int Lp, int batch, float scale, float drop_p,
                                 uint32_t seed, int H, int h, void* stream, int all_heads = 0) {
    SC_CHECK_ARG(S && dP && dO && O && P && dS && L > 0 && Lp >= L && batch > 0 && batch <= 65535 && ld >= Lp, "sc_attn_softmax_bwd: bad arguments");
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && H >= 1 && h >= 0 && h < H, "sc_attn_softmax_bwd_dropout: bad dropout arguments");
    // forward element index = ((b*H + h)*L + i)*L + j with b = z
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((Lp + 3) / 4, batch), dim3(256), 0, (hipStream_t)stream, S, dP, ld, stride, (const bf16_t*)dO, ld_do,
                       (const bf16_t*)O, ld_o, rows_per_batch, klens, (bf16_t*)P, (bf16_t*)dS, L, Lp, scale, seed, drop_thresh16(drop_p), 1.0f / (1.0f - drop_p),
                       all_heads ? 0u : (uint32_t)h * (uint32_t)L, (all_heads ? 1u : (uint32_t)H) * (uint32_t)L,      // row id = (b*H + h)*L + i
                       all_heads ? H : 0);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_attn_softmax_bwd(const float* S, const float* dP, int64_t ld, int64_t stride, const void* dO, int64_t ld_do, const void* O, int64_t ld_o,
                                   int64_t rows_per_batch, const int32_t* klens, void* P, void* dS, int L, int Lp, int batch, float scale, void* stream) {
    return attn_softmax_bwd_impl(S, dP, ld, stride, dO, ld_do, O, ld_o, rows_per_batch, klens, P, dS, L, Lp, batch, scale, 0.f, 0u, 1, 0, stream);
}

// the same with the forward's attention dropout (sc_attention_fwd_dropout, same seed): head h of H, batch z = utterance
extern "C" int sc_attn_softmax_bwd_dropout(const float* S, const float* dP, int64_t ld, int64_t stride, const void* dO, int64_t ld_do, const void* O,
                                           int64_t ld_o, int64_t rows_per_batch, const int32_t* klens, void* P, void* dS, int L, int Lp, int batch,
                                           float scale, float drop_p, uint32_t seed, int H, int h, void* stream) {
    return attn_softmax_bwd_impl(S, dP, ld, stride, dO, ld_do, O, ld_o, rows_per_batch, klens, P, dS, L, Lp, batch, scale, drop_p, seed, H, h, stream);
}

// all heads in one launch: batch = B * H images (z = b*H + h), dO / O = the [rows, H*64] matrices themselves, klens per utterance; drop_p = 0: no dropout
extern "C" int sc_attn_softmax_bwd_heads(const float* S, const float* dP, int64_t ld, int64_t stride, const void* dO, int64_t ld_do, const void* O,
                                         int64_t ld_o, int64_t rows_per_batch, const int32_t* klens, void* P, void* dS, int L, int Lp, int B, int H,
                                         float scale, float drop_p, uint32_t seed, void* stream) {
    SC_CHECK_ARG(B > 0 && H > 0 && (int64_t)B * H <= 65535, "sc_attn_softmax_bwd_heads: B*H must be in [1, 65535]");
    return attn_softmax_bwd_impl(S, dP, ld, stride, dO, ld_do, O, ld_o, rows_per_batch, klens, P, dS, L, Lp, B * H, scale, drop_p, seed, H, 0, stream, 1);
}

// P / dS bf16 [B*H, Lp, Lp] (rows >= L and keys >= klens[b] zero) from the packed q | k | v rows (row b*L + t, head h at column h*64), dO and O:
// the fused form of (S = Q K^T, dP = dO V^T as batched GEMMs) + sc_attn_softmax_bwd_heads.  drop_p > 0: the forward ran sc_attention_fwd_dropout(seed).
extern "C" int sc_attn_bwd_probs(const void* q, const void* k, const void* v, int64_t ld_qkv, const void* dO, const void* O, int64_t ld_o,
                                 const int32_t* klens, void* P, void* dS, int B, int H, int L, int Lp, float scale, float drop_p, uint32_t seed,
                                 void* stream) {
    SC_CHECK_ARG(q && k && v && dO && O && P && dS, "sc_attn_bwd_probs: null operand");
    SC_CHECK_ARG(B > 0 && H > 0 && (int64_t)B * H <= 65535 && L > 0 && Lp >= L && Lp % 16 == 0, "sc_attn_bwd_probs: bad sizes (Lp must be a multiple of 16)");
    SC_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dO | (uintptr_t)O) & 15) == 0),
                 "sc_attn_bwd_probs: rows must be 16-byte aligned");
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_attn_bwd_probs: drop_p=%f must be in [0, 1)", (double)drop_p);
    hipLaunchKernelGGL(attn_bwd_probs_kernel, dim3((Lp + 63) / 64, B * H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, ld_qkv, (const bf16_t*)dO, (const bf16_t*)O, ld_o, klens, (bf16_t*)P, (bf16_t*)dS, H, L, Lp, scale, seed,
                       drop_thresh16(drop_p), 1.0f / (1.0f - drop_p));
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_gelu_bwd_bf16(const void* u, const void* dh, void* du, int64_t n, void* stream) {
    SC_CHECK_ARG(n % 2 == 0, "sc_gelu_bwd_bf16: n=%lld must be even", (long long)n);
    if (n <= 0) return 0;
    SC_CHECK_ARG((((uintptr_t)u | (uintptr_t)dh | (uintptr_t)du) & 15) == 0, "sc_gelu_bwd_bf16: operands must be 16-byte aligned");
    hipLaunchKernelGGL(gelu_bwd_bf16_kernel, dim3((unsigned)(((n + 7) / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u, (const bf16_t*)dh,
                       (bf16_t*)du, n);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_layernorm_bwd_bf16_partials(int64_t rows) {   // number of partial rows the kernel writes: part is f32 [partials, 2, D]
    const int64_t rpb = rows >= 4096 ? 32 : 4;
    return (rows + rpb - 1) / rpb;
}

extern "C" int sc_layernorm_bwd_bf16(const void* x, const void* dy, const float* gamma, void* dx, float* part, int64_t rows, int D, float eps, void* stream) {
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 64 == 0, "sc_layernorm_bwd_bf16: D=%d must be a multiple of 64, <= 1024", D);
    if (rows <= 0) return 0;
    const int rpb = rows >= 4096 ? 32 : 4;
    if (D % 256 == 0)
        hipLaunchKernelGGL(ln_bwd_bf16_vec_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const bf16_t*)dy, gamma, (bf16_t*)dx, part, rows, D, eps, rpb);
    else
    hipLaunchKernelGGL(ln_bwd_bf16_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, gamma,
                       (bf16_t*)dx, part, rows, D, eps, rpb);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t sc_colsum_bf16_workspace_bytes(int64_t rows, int cols) {
    const int64_t rpb = rows >= 65536 ? 128 : 64;
    return ((rows + rpb - 1) / rpb) * (int64_t)cols * 4;
}

extern "C" int sc_colsum_bf16(const void* x, int64_t ld, int64_t rows, int cols, float* ws, float* out, int accumulate, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    const int rpb = rows >= 65536 ? 128 : 64;
    const int nparts = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_bf16_stage1_kernel, dim3(nparts, (cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, rows, cols, ws, rpb);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, nparts, cols, out, accumulate);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_axpy_bf16(void* y, const void* x, float alpha, int64_t n, void* stream) {
    SC_CHECK_ARG(n % 2 == 0, "sc_axpy_bf16: n=%lld must be even", (long long)n);
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_bf16_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)y, (const bf16_t*)x, alpha, n);
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_cls_pool_dz(const float* pp, const float* ds, const float* dzbar, const float* u, const int32_t* lens, void* dz, int B, int T, int NQ, int R,
                              int D, int64_t ld_dz, void* stream) {
    SC_CHECK_ARG(pp && ds && dzbar && u && lens && dz && B > 0 && T > 0 && B <= 65535, "sc_cls_pool_dz: bad arguments");
    hipLaunchKernelGGL(cls_pool_dz_kernel, dim3(T, B), dim3(256), 0, (hipStream_t)stream, pp, ds, dzbar, u, lens, (bf16_t*)dz, T, NQ, R, D, ld_dz);
    SC_CHECK_LAUNCH();
    return 0;
}
