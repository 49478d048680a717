// Grouped positional convolution of the HuBERT encoder (speech_encoder_plus.py:32-40; fairseq pos_conv: Conv1d(D, D, k = Kw, pad = Kw/2,
// groups = G) + SamePad) on the matrix cores WITHOUT materialising the sliding windows.
//
// For group g the conv is  out[t, n] = sum_{tap, c} x[t - Kw/2 + tap, g*cg + c] * w[g, n, tap*cg + c]  -- a GEMM whose A row t is the
// contiguous run of Kw*cg elements starting at element t*cg of the group's [frames][cg] slab (rows overlap by all but cg elements).  The
// generic GEMM streams every row's full K through L2 -> LDS (128x redundant); here a block keeps the UNIQUE window ((MB + Kw) x cg
// bf16) in LDS once, and the MFMA A fragment of (row t, k-chunk q) is simply the 16 bytes at window byte (t*cg + 8q)*2: consecutive
// k-steps are +128 B immediates.  Only the weights (cg x Kw*cg per group, L2-resident) stream, through a 3-slot LDS-DMA ring.
// Block = (frame chunk of MB = 64*NW rows, group, utterance); wave tile 64 x cg.  Padded / invalid frames enter as zeros (the
// index_put of speech_encoder_plus.py:32), fusing the former pack pass.
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

__device__ __forceinline__ void pc_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int PC_NSTG = 3;

template <int NJ, int NW>   // cg = 16 NJ channels per group, MB = 64 NW frames per block
__global__ __launch_bounds__(NW * 64) void posconv_mfma_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ valid, const bf16_t* __restrict__ wg,
                                                               bf16_t* __restrict__ conv, int Tp_uniform, int D, int G, int Kw,
                                                               const int32_t* __restrict__ row_off) {
    constexpr int CG = 16 * NJ, MB = 64 * NW, NT = NW * 64;
    constexpr int STG = CG * 128;                              // one W stage: [CG rows][64 k] bf16
    constexpr int NCHUNK = CG * 8;                             // 16-byte chunks per W stage
    constexpr int ROUNDS = (NCHUNK + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem_pc[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, g = blockIdx.y, t_start = blockIdx.x * MB;
    // packed batches: utterance b owns rows [row_off[b], row_off[b + 1]) of x; its conv slab is [G][Tp][cg] at element row_off[b] * D
    const int Tp = row_off ? row_off[b + 1] - row_off[b] : Tp_uniform;
    const int64_t row0 = row_off ? (int64_t)row_off[b] : (int64_t)b * Tp;
    if (t_start >= Tp) return;
    const int K = Kw * CG, nk = K / 64;
    const int win_bytes = ((MB + Kw) * CG * 2 + 15) & ~15;
    char* ring = smem_pc + win_bytes;

    int my_cnt = 0;                                            // DMA instructions this wave issues per stage (wave-uniform)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) my_cnt += (r * NT + wave * 64 < NCHUNK) ? 1 : 0;
    const bf16_t* wsrc = wg + (int64_t)g * CG * K;
    auto stage = [&](int st) {
        char* slot = ring + (st % PC_NSTG) * STG;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r * NT + wave * 64 < NCHUNK) {                 // whole waves in or out (NCHUNK is a multiple of 64)
                const int id = r * NT + tid, n = id >> 3, pos = id & 7;
                pc_glds16(wsrc + (int64_t)n * K + st * 64 + ((pos ^ (n & 7)) << 3), slot + (r * NT + wave * 64) * 16);
            }
        }
    };
    stage(0);
    if (nk > 1) stage(1);

    // the unique input window of this block: frames t_start - Kw/2 .. t_start + MB + Kw/2 - 1, channels of group g
    int vlen = valid ? valid[b] : Tp;
    vlen = vlen > Tp ? Tp : vlen;
    {
        constexpr int CPR = CG / 8;                            // 16-byte chunks per window row
        const int total = (MB + Kw) * CPR;
        for (int idx = tid; idx < total; idx += NT) {
            const int r = idx / CPR, cc = idx - r * CPR;
            const int t_in = t_start - Kw / 2 + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t_in >= 0 && t_in < vlen) v = *(const uint4*)(x + (row0 + t_in) * D + g * CG + cc * 8);
            *(uint4*)(smem_pc + ((int64_t)r * CG + cc * 8) * 2) = v;
        }
    }
    const int frow = lane & 15, fk = lane >> 4;
    f32x4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* a_base = smem_pc + ((wave * 64 + frow) * CG + fk * 8) * 2;
    const bool wave_live = t_start + wave * 64 < Tp;            // wave-uniform
    const int w_off0 = frow * 128 + ((fk ^ (frow & 7)) << 4), w_off1 = frow * 128 + (((4 + fk) ^ (frow & 7)) << 4);

    auto wait_mine = [&](bool keep_one) {                      // all of this wave's stages landed, except (keep_one) the newest
        if (!keep_one || my_cnt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (my_cnt == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };
    for (int kt = 0; kt < nk; ++kt) {
        wait_mine(kt + 1 < nk);                                // stage kt (and, on kt = 0, the window stores: lgkmcnt below)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // every wave: done with stage kt-1, stage kt visible
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) stage(kt + 2);                        // into the slot of stage kt-1
        const char* slot = ring + (kt % PC_NSTG) * STG;
        if (!wave_live) continue;                                // (packed batches) this wave's 64 frames lie beyond the utterance: DMA + barriers only
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8_t af[4], wf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8_t*)(a_base + i * 16 * CG * 2 + kt * 128 + h * 64);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *(const bf16x8_t*)(slot + j * 16 * 128 + (h ? w_off1 : w_off0));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    // conv[b][g][t][n]: lane (frow, fk) holds frame frow of its 16-row block, channels 16j + 4fk .. +3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t_start + wave * 64 + i * 16 + frow;
        if (t >= Tp) continue;
        bf16_t* orow = conv + (row0 * G + (int64_t)g * Tp + t) * CG + fk * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            uint2 o;
            o.x = pack2bf(acc[i][j][0], acc[i][j][1]);
            o.y = pack2bf(acc[i][j][2], acc[i][j][3]);
            *(uint2*)(orow + j * 16) = o;
        }
    }
}

template <int NJ, int NW>
int posconv_launch(const void* x, const int32_t* valid, const void* wg, void* conv, int B, int Tp, int D, int G, int Kw, hipStream_t s, const int32_t* row_off) {
    constexpr int CG = 16 * NJ, MB = 64 * NW;
    const int lds = (((MB + Kw) * CG * 2 + 15) & ~15) + PC_NSTG * CG * 128;
    SC_CHECK_ARG(lds <= 160 * 1024, "sc_posconv_conv: Kw=%d too large for LDS", Kw);
    (void)hipFuncSetAttribute((const void*)posconv_mfma_kernel<NJ, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((posconv_mfma_kernel<NJ, NW>), dim3((Tp + MB - 1) / MB, G, B), dim3(NW * 64), lds, s, (const bf16_t*)x, valid, (const bf16_t*)wg,
                       (bf16_t*)conv, Tp, D, G, Kw, row_off);
    SC_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// conv bf16 [B][G][Tp][cg] = grouped conv of x bf16 [B*Tp, D] (frames >= valid[b] read as zero) with wg bf16 [G][cg][Kw*cg]
// (K index = tap*cg + c_in).  Returns 1 (and does nothing) when the shape is not covered (caller falls back to pack + batched GEMM).
static int posconv_conv_impl(const void* x, const int32_t* valid, const void* wg, void* conv, int B, int Tp, int D, int G, int Kw, void* stream,
                             const int32_t* row_off) {
    SC_CHECK_ARG(B > 0 && Tp > 0 && G > 0 && D % G == 0, "sc_posconv_conv: bad shape");
    const int cg = D / G;
    if (cg % 16 != 0 || cg > 64 || cg < 32 || (Kw * cg) % 64 != 0 || D % 8 != 0 || B > 65535 || G > 65535) return 1;
    static const int force_nw = getenv("SC_POSCONV_NW") ? atoi(getenv("SC_POSCONV_NW")) : 0;
    // 8-wave blocks (512 frames) unless that wastes much more of the last chunk than 4-wave blocks (256 frames) would
    const int waste8 = (Tp + 511) / 512 * 512 - Tp, waste4 = (Tp + 255) / 256 * 256 - Tp;
    const int nw = force_nw ? force_nw : (waste8 <= waste4 + Tp / 8 ? 8 : 4);
    hipStream_t s = (hipStream_t)stream;
#define PC_CASE(NJ_) (nw == 8 ? posconv_launch<NJ_, 8>(x, valid, wg, conv, B, Tp, D, G, Kw, s, row_off) : posconv_launch<NJ_, 4>(x, valid, wg, conv, B, Tp, D, G, Kw, s, row_off))
    if (cg == 32) return PC_CASE(2);
    if (cg == 48) return PC_CASE(3);
    return PC_CASE(4);
#undef PC_CASE
}

extern "C" int sc_posconv_conv(const void* x, const int32_t* valid, const void* wg, void* conv, int B, int Tp, int D, int G, int Kw, void* stream) {
    return posconv_conv_impl(x, valid, wg, conv, B, Tp, D, G, Kw, stream, nullptr);
}

// Packed (padding-free) batches: utterance b owns rows [row_off[b], row_off[b + 1]) of x (<= Tmax rows each); conv slab of utterance b =
// [G][rows_b][cg] at element row_off[b] * D.  Same return convention as sc_posconv_conv.
extern "C" int sc_posconv_conv_packed(const void* x, const int32_t* valid, const int32_t* row_off, const void* wg, void* conv, int B, int Tmax, int D, int G,
                                      int Kw, void* stream) {
    SC_CHECK_ARG(row_off != nullptr && valid != nullptr, "sc_posconv_conv_packed: row_off and valid are required");
    return posconv_conv_impl(x, valid, wg, conv, B, Tmax, D, G, Kw, stream, row_off);
}
