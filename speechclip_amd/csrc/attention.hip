// Scaled-dot-product attention kernels for gfx950.
//  (1) attn_fwd_kernel: flash-style forward for head_dim 64 with per-utterance key lengths
//      (HuBERT layers: fairseq MultiheadAttention with key_padding_mask; CLIP ViT blocks: no mask).
//      One wave owns 32 query rows.  S^T = K.Q^T is computed with the operands swapped
//      (mfma(K, Q)) so a lane holds 16 keys of ONE query: the row max/sum is in-lane plus one
//      cross-half shuffle, and the exponentiated registers are already in B-operand order for the
//      O^T += V^T.P^T MFMA (the key->k-slot permutation is chosen to match, so no permlane).
//      K tiles arrive by LDS-DMA into an XOR-swizzled image; V tiles are transposed through
//      registers into a padded V^T image (conflict-free ds_read_b64).  Double-buffered, one
//      barrier per 64-key tile; tiles past the utterance's key length are skipped entirely.
//  (2) cls_attn_kernel: the CLS-rows-only attention of the parallel / cascaded heads: NQ <= 8 learned
//      query tokens against [CLS tokens ; valid frames] for any head_dim <= 1024 (VALU, HBM-bound).
#include <type_traits>
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int KV = 64;          // keys per tile
constexpr int K_TILE_BYTES = KV * 128;
constexpr int STAGE_BYTES = 2 * K_TILE_BYTES;   // K tile [64 keys][128 B] + V tile [64 keys][128 B], both row-major
constexpr int NSTAGE = 3;
// -DSC_ATTN_PP=1 (build option): the two waves a SIMD holds of one 8-wave block (w and w + 4) run half a tile apart -- waves 4-7 enter one
// barrier interval late, two barriers per tile (after S = K.Q^T and after the softmax arithmetic) keep the offset, so one half's softmax (VALU)
// always runs beside the other half's P.V + next K.Q^T (MFMA).  Needs a 4-slot K/V ring (a tile stays in use for four intervals).
// Timing probes and measured-and-rejected variants (SC_ATTN_ABL, SC_ATTN_NOBAR, SC_ATTN_PP, SC_ATTN_VEARLY; env SC_ATTN_NW / SC_ATTN_QB / SC_ATTN_IPB) are
// honoured only in the PROBES build (`make PROBES=1`, -DSC_PROBES=1); the product library compiles the default form alone.
#ifndef SC_PROBES
#define SC_PROBES 0
#endif
#if !SC_PROBES
#undef SC_ATTN_ABL
#undef SC_ATTN_NOBAR
#undef SC_ATTN_PP
#undef SC_ATTN_VEARLY
#define SC_ATTN_ENV_INT(name, dflt) (dflt)
#else
#define SC_ATTN_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#endif
#ifndef SC_ATTN_PP
#define SC_ATTN_PP 0
#endif
#ifndef SC_ATTN_SCALAR             // bit 0: score offset as single v_fma_f32; bit 1: row sums as single v_add_f32 (instead of the packed-f32 forms)
#define SC_ATTN_SCALAR 0
#endif
#ifndef SC_ATTN_LAZY_LOG2          // rescale threshold of the online softmax in log2 units (0: the textbook form, every rise of the maximum rescales)
#define SC_ATTN_LAZY_LOG2 8.0f
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// K chunk swizzle (16-B chunk p of key row r holds d-chunk p ^ (r & 7)); V chunk swizzle ((r >> 1) & 3) << 1 keeps the four
// consecutive key rows touched by one ds_read_b64_tr_b16 group on different bank groups.
__device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 3) << 1; }
// K rows: a ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with (row >> 1) & 7 every group sees
// each swizzle value exactly twice, once on an even and once on an odd row (the two 128-byte bank halves): conflict-free.  (row & 7
// repeats inside a group on rows of equal parity: 2-way conflicts, measured SQ_LDS_BANK_CONFLICT = 48 % of the LDS cycles.)
#ifndef SC_ATTN_KSWZ
#define SC_ATTN_KSWZ 1
#endif
__device__ __forceinline__ int swz_k(int row) { return SC_ATTN_KSWZ ? ((row >> 1) & 7) : (row & 7); }

__device__ unsigned long long* g_attn_trace = nullptr;   // debug: per-block phase cycles (sc_debug_set_attn_trace)

template <int NW, bool TRACE, bool DROP = false, int QB = 1, bool F16 = false>   // F16: q / k / v / out (and P) are IEEE half instead of bf16 (SC_ATTN_F16)   // QB: 32-row query blocks per wave (2: every K / V fragment read feeds two MFMAs, 256 registers, two waves per SIMD)
   // DROP: attention-probability dropout (train-mode frozen encoder, sc_attention_fwd_dropout); waves per block: NW x 32 query rows share one K/V ring (4: 128 rows, 8: 256 rows -- half the K/V traffic, 4 waves per SIMD)
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(QB == 1 ? 4 : 2))) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                       const int32_t* __restrict__ klens, int T_uniform, int64_t ld_qkv,
                                                       int64_t ld_out, float scale_log2e, int causal, int B, int H, int nq, int n_ids, int ipb,
                                                       uint32_t drop_seed, uint32_t drop_thresh_, float drop_keep_scale,
                                                       const int32_t* __restrict__ row_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NSTAGE x (K 8 KiB + V 8 KiB)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int ROWS = NW * 32 * QB;          // query rows per block
    constexpr bool PP = SC_ATTN_PP && NW == 8;
    constexpr int NST = PP ? 4 : NSTAGE;
    const bool gy = PP && wave >= NW / 2;       // the late half (wave-uniform)
    // XCD-aware 1-D grid: blocks id, id+8, id+16.. share an XCD (and its L2); the nq query blocks of one (b, h) unit are made
    // consecutive on ONE XCD so K/V (128 KB per unit) is fetched from HBM once instead of once per query block.
    // A block can walk `ipb` ids (same XCD: ids bid%8 + 8*(...)).  tools/attn_trace.py shows ~25 % of the resident slots empty at any time
    // with one id per block, but chaining ids inside a block (ipb 2..12, or fully persistent blocks) measured 4-6 % SLOWER: the empty
    // time is the per-id prologue (Q load, first two K/V stages) + epilogue, which a fresh block overlaps with its neighbours just as
    // well.  ipb = 1 by default; SC_ATTN_IPB is kept for that experiment.
    for (int it = 0; it < ipb; ++it) {
    const int id = (blockIdx.x >> 3) * (8 * ipb) + it * 8 + (blockIdx.x & 7);
    if (id >= n_ids) break;
    const int unit = ((id >> 3) / nq) * 8 + (id & 7);
    int qblk = (id >> 3) % nq;
    if (unit >= H * B) continue;
    const int b = unit / H, h = unit - b * H;
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i));        // per-id recomputation of the lane geometry: keeps the id loop's invariants out of VGPRs held across it
    const int g = lane_i >> 5, ql = lane_i & 31;
    // packed (padding-free) batches: utterance b owns rows [row_off[b], row_off[b + 1]) of q / k / v / out; T = its own row count
    // (wave-uniform scalar loads).  row_off == nullptr: the uniform layout, T rows per utterance.
    const int T = row_off ? row_off[b + 1] - row_off[b] : T_uniform;
    const int64_t row_base = row_off ? (int64_t)row_off[b] : (int64_t)b * T;
    // Query-row split of the 8-wave form (host: attention_fwd_impl).  A last block with <= 128 valid rows computes 256 -- every wave runs the full key loop on
    // clamped rows: T = 257 (ViT-L/14) and T = 319 (the training crop) paid two blocks for 1.004 / 1.25 blocks of rows, packed utterances half a block each on
    // average.  q_mode 1 (the NW == 8 launch) leaves such a block to a second launch of 4-wave blocks, q_mode 2 (NW == 4: ROWS == 128), one id per (b, h).
    const int q_mode = causal >> 1;
    if (q_mode == 2) {
        const int rem = T & 255;
        if (rem == 0 || rem > 128) continue;
        qblk = (T >> 8) * 2;
    }
    if (qblk * ROWS >= T) continue;
    if (q_mode == 1 && T - qblk * ROWS <= 128) continue;
    const int hoff = h * 64;

    int klen = klens ? klens[b] : T;
    klen = klen < 0 ? 0 : (klen > T ? T : klen);
    int nkv = (klen + KV - 1) / KV;
    if (causal & 1) {  // keys beyond the block's last query are never needed
        const int last_q = min(T, qblk * ROWS + ROWS);
        nkv = min(nkv, (last_q + KV - 1) / KV);
    }

    int qrow[QB], qrow_c[QB];
    uint32_t drop_row[QB];
    const uint32_t drop_pairs = (uint32_t)(((row_off ? T_uniform : T) + 1) >> 1);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = qblk * ROWS + wave * (32 * QB) + qb * 32 + ql;
        qrow_c[qb] = qrow[qb] < T ? qrow[qb] : T - 1;
        drop_row[qb] = DROP ? (row_off ? (uint32_t)((row_base + qrow_c[qb]) * H + h) : (uint32_t)((b * H + h) * T + qrow_c[qb])) : 0u;
    }

    // Q fragments: B operand of S^T (col = query, k-slots = 8 dims)
    bf16x8_t qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            qf[qb][c] = *(const bf16x8_t*)(q + (row_base + qrow_c[qb]) * ld_qkv + hoff + c * 16 + g * 8);

    // K and V tiles both arrive by LDS-DMA (no VGPR staging): 2 + 2 instructions per thread per tile.  Source addressing is kept
    // cheap: a 64-bit per-thread base (row 0 of the unit, this thread's key row / chunk) + a 32-bit row offset per tile.
    constexpr int NLD = 512 / (NW * 64);                            // row groups per thread per operand: 2 (NW = 4) or 1 (NW = 8)
    int tid_i = tid;
    asm volatile("" : "+v"(tid_i));
    const int key0 = tid_i >> 3, pos = tid_i & 7;                   // rows key0 (+ 32) of the tile, 16-byte chunk pos
    const bf16_t* kbase = k + row_base * ld_qkv + hoff;
    const bf16_t* vbase = v + row_base * ld_qkv + hoff;
    const int kchunk0 = (pos ^ swz_k(key0)) << 3, vchunk0 = (pos ^ swz_v(key0)) << 3;          // (key0 + 32) has the same low bits
    auto stage = [&](int tile, int buf) {
        char* kb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int kr = tile * KV + key0 + 32 * i;
            kr = kr < T ? kr : T - 1;
            const int64_t rowoff = (int64_t)kr * ld_qkv;
            glds16(kbase + rowoff + kchunk0, kb + (i * 256 + wave * 64) * 16);
            glds16(vbase + rowoff + vchunk0, kb + K_TILE_BYTES + (i * 256 + wave * 64) * 16);
        }
    };
    auto wait_stage = [&](bool keep_one) {    // all but the newest stage landed (2 * NLD DMA instructions per stage)
        if (!keep_one) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (NLD == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    };

    f32x16_t o[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; }
        m_run[qb] = -INFINITY; l_run[qb] = 0.f;
    }

    // Lane parts of the LDS fragment addresses (ring slot and kb / hb / +8-row terms are compile-time immediates below).
    // K: key = kb*32 + ql, chunk (2c+g) ^ (key & 7) -- the XOR makes the four c distinct lane parts.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned ka[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ka[c] = lds_base + ql * 128 + (((2 * c + g) ^ swz_k(ql)) << 4);
    // V^T (ds_read_b64_tr_b16: within a 16-lane group, lane i points at key row i>>2, 4 consecutive d at (i&3)*4, and receives column
    // (i) of the 4 x 16 block, i.e. V[k0..k0+3][d0 + i]): k0 = kb*32 + hb*16 + 4g + v_row_in; rows k0 and k0+8 share their swizzle.
    const int ti = lane_i & 15;
    const int v_row_in = ti >> 2;                                  // key row within the group of 4
    const int v_chunk_in = (((lane_i >> 4) & 1) << 1) + ((ti & 3) >> 1);   // 16-B chunk within the 64-B d-block
    const int v_byte_in = (ti & 1) * 8;
    const int vk0 = 4 * g + v_row_in;
    unsigned va[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) va[db] = lds_base + K_TILE_BYTES + vk0 * 128 + (((db * 4 + v_chunk_in) ^ swz_v(vk0)) << 4) + v_byte_in;

    if (nkv > 0) {
        stage(0, 0);
        if (nkv > 1) stage(1, 1);
        if (PP) {
            if (nkv > 2) { stage(2, 2); asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }      // NW = 8: two DMA instructions per stage and thread
            else wait_stage(nkv > 1);
        } else wait_stage(nkv > 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (gy) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    // One KV tile; SLOT (ring slot of tile j) is a compile-time constant so every LDS address is lane part + immediate.
    // All LDS reads are inline asm with hand-counted lgkmcnt: for a builtin / plain load the compiler cannot prove that the read does
    // not alias the LDS-DMA writes in flight and guards it with s_waitcnt vmcnt(0), which would drain the prefetch of tile j+2.
    unsigned long long* const trace = TRACE ? g_attn_trace : nullptr;
    unsigned long long tr_qk = 0, tr_sm = 0, tr_pv = 0, tr_bar = 0, tr_t = TRACE ? __builtin_readcyclecounter() : 0;
    const unsigned long long tr_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz, chip-wide
    auto stamp = [&](unsigned long long& acc_) { if (TRACE) { const unsigned long long t = __builtin_readcyclecounter(); acc_ += t - tr_t; tr_t = t; } };
    auto tile_body = [&](int j, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
#if !(defined(SC_ATTN_ABL) && SC_ATTN_ABL == 3)   // perf probe 3 (garbage results): no K / V DMA inside the tile loop
        if (!PP && j + 2 < nkv) stage(j + 2, (SLOT + 2) % NSTAGE);   // that buffer held tile j-1: every wave passed the barrier after reading it
#endif
        // ---- S^T = K . Q^T : all 8 K fragments are requested up front, the MFMAs then run back to back
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        // 6 of the 8 K fragments are requested up front, the last two once the first two MFMAs have consumed theirs (their registers are
        // free again): 24 instead of 32 VGPRs at the peak of the kernel's register pressure -- what keeps it at <= 128 with the
        // persistent-block loop around it.  lgkmcnt bookkeeping: R0..R5 out -> wait 5,4 -> R6,R7 out -> wait 5,4,3,2,1,0.
        f32x16_t s[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[qb][0][i] = 0.f; s[qb][1][i] = 0.f; }
        auto kread = [&](int kb, int c) -> u32x4_t {
            const unsigned addr = ka[c];
            u32x4_t t;
#if defined(SC_ATTN_ABL) && SC_ATTN_ABL == 5      // perf probe 5 (garbage results): no LDS fragment reads
            t = (u32x4_t){addr, addr, addr, addr};
            return t;
#endif
            if (kb == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "i"(SLOT * STAGE_BYTES));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "i"(SLOT * STAGE_BYTES + 4096));
            return t;
        };
        auto kmma = [&](int kb, int c, u32x4_t t) {
            asm volatile("" : "+v"(t));
#if defined(SC_ATTN_ABL) && SC_ATTN_ABL == 4      // perf probe 4 (garbage results): no MFMA
            asm volatile("" :: "v"(t));
            s[0][kb][c] += 1.0f;
#else
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
                s[qb][kb] = mfma_32x32x16<F16>(__builtin_bit_cast(bf16x8_t, t), qf[qb][c], s[qb][kb]);
#endif
        };
        {
            u32x4_t k00 = kread(0, 0), k01 = kread(0, 1), k02 = kread(0, 2), k03 = kread(0, 3), k10 = kread(1, 0), k11 = kread(1, 1);
            asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); kmma(0, 0, k00);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); kmma(0, 1, k01);
            u32x4_t k12 = kread(1, 2), k13 = kread(1, 3);
            asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); kmma(0, 2, k02);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); kmma(0, 3, k03);
            asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); kmma(1, 0, k10);
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); kmma(1, 1, k11);
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); kmma(1, 2, k12);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); kmma(1, 3, k13);
        }
        if (PP) {      // interval boundary a: the late half has its share of tile j+1 landed; tile j-1 is free for everyone after it
            if (gy) wait_stage(j + 2 < nkv);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (gy && j + 3 < nkv) stage(j + 3, (SLOT + 3) % NST);
        }
        // ---- O^T += V^T . P^T : V^T fragments come straight out of the row-major V tile via the transposing LDS read
        u32x2_t vlo[2][2], vhi[2][2];                               // [parity of c4][db]
        auto issue_v = [&](auto c4c) {
            constexpr int c4 = decltype(c4c)::value;
            constexpr int off = SLOT * STAGE_BYTES + (c4 >> 1) * 4096 + (c4 & 1) * 2048;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const unsigned addr = va[db];
                u32x2_t t0, t1;
#if defined(SC_ATTN_ABL) && SC_ATTN_ABL == 5
                t0 = (u32x2_t){addr, addr}; t1 = t0;
#else
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t0) : "v"(addr), "i"(off));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t1) : "v"(addr), "i"(off + 1024));
#endif
                vlo[c4 & 1][db] = t0; vhi[c4 & 1][db] = t1;
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
#ifndef SC_ATTN_VEARLY    // 1: the first V fragments are requested before the softmax arithmetic instead of after it (8 more live VGPRs)
#define SC_ATTN_VEARLY 0
#endif
        if (SC_ATTN_VEARLY && !PP) issue_v(I0{});
        const int kv0 = j * KV;
        const bool partial = (kv0 + KV > klen) || (causal & 1);
        unsigned ppk[QB][2][8];
#if defined(SC_ATTN_ABL) && SC_ATTN_ABL == 2
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) ppk[qb][i >> 3][i & 7] = __float_as_uint(s[qb][i >> 3][i] + s[qb][i >> 3][15 - i]);
#endif
#if !(defined(SC_ATTN_ABL) && SC_ATTN_ABL == 2)   // perf probe 2: no softmax arithmetic at all (MFMA + LDS + DMA skeleton)
        // ---- mask + online softmax (log2 domain).  Only a tile that crosses the key length (or the causal diagonal)
        //      pays for per-element masking; full tiles take the short path.
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
        float mx = -INFINITY;
        if (partial) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    float t = s[qb][kb][r] * scale_log2e;
                    t = (key < klen && (!(causal & 1) || key <= qrow[qb])) ? t : -INFINITY;
                    s[qb][kb][r] = t;
                    mx = fmaxf(mx, t);
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
            mx *= scale_log2e;   // scale > 0: max commutes with the scaling
        }
        {   // the other half-wave's maximum by v_permlane32_swap (a VALU op) instead of an LDS-crossbar shuffle: no lgkmcnt round trip in the tile's serial chain
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        asm volatile("" :: "v"(mx));   // phase boundary (also keeps the scheduler from interleaving the phases into a register-pressure peak)
        if (TRACE && qb == 0) stamp(tr_qk);           // S complete (the max depends on every MFMA result)
        // Lazy rescale (round 3): the kernel runs at the speed of its vector instruction stream (profiles/r03_attention_two_score_tile_experiment.txt), and
        // rescaling O by alpha is 1/6 of that stream while the running maximum of SOME row of a wave moves on nearly every tile.  The reference
        // maximum m_run therefore follows the true running maximum only when that rose by more than 2^SC_ATTN_LAZY_LOG2: otherwise P = exp2(s - m_run)
        // is simply up to 2^8 instead of <= 1 (bf16 and the fp32 sums have the range; l_run uses the same reference, so the quotient is unchanged).
        float alpha = 1.0f;
        {
            const float m_cand = fmaxf(m_run[qb], mx);
            if (m_cand - m_run[qb] > SC_ATTN_LAZY_LOG2) {             // first tile: -inf -> finite
                alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_cand);   // exp2(-inf) = 0 on the first tile
                m_run[qb] = m_cand;
            }
        }
        const float m_new = m_run[qb];
        f32x2_t psum2 = {0.f, 0.f};
        const float sc = partial ? 1.0f : scale_log2e;               // partial tiles were scaled while masking
#if SC_ATTN_SCALAR & 1
        const float neg_m = -m_new;
#endif
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
#if SC_ATTN_SCALAR & 1      // the score offset as two single v_fma_f32 (asm: plain -O3 SLP-packs adjacent f32 ops into v_pk_fma_f32, which is priced above two v_fma_f32 beside MFMAs -- MI355X_MICROARCH.md)
                f32x2_t a2;
                asm("v_fma_f32 %0, %1, %2, %3" : "=v"(a2[0]) : "v"(s[qb][kb][r]), "v"(sc), "v"(neg_m));
                asm("v_fma_f32 %0, %1, %2, %3" : "=v"(a2[1]) : "v"(s[qb][kb][r + 1]), "v"(sc), "v"(neg_m));
#else
                const f32x2_t a2 = (f32x2_t){s[qb][kb][r], s[qb][kb][r + 1]} * sc - m_new;       // one v_pk_fma_f32 per pair
#endif
#if defined(SC_ATTN_ABL) && SC_ATTN_ABL == 1
                const f32x2_t p2 = a2 * 0.001f;                                            // perf probe: no transcendental
#else
                const f32x2_t p2 = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
#endif
                if (DROP) {
                    // torch: attn = dropout(softmax(s)) -- the row sum keeps every probability, the P.V product sees the masked, rescaled ones.
                    // this lane's registers r, r + 1 hold the adjacent keys kv0 + kb*32 + (r & 3) + 8*(r >> 2) + 4*g (even) and + 1: one hash per pair
                    const uint32_t key = (uint32_t)(kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g);
                    const uint32_t hbits = hash_pair(drop_seed, drop_row[qb] * drop_pairs + (key >> 1));
                    const float d0 = (hbits & 0xffffu) >= drop_thresh_ ? p2[0] * drop_keep_scale : 0.f;
                    const float d1 = (hbits >> 16) >= drop_thresh_ ? p2[1] * drop_keep_scale : 0.f;
                    ppk[qb][kb][r >> 1] = pack2x<F16>(d0, d1);
                } else
                ppk[qb][kb][r >> 1] = pack2x<F16>(p2[0], p2[1]);       // P leaves the fp32 registers right here: 16 VGPRs instead of 32 through the PV phase
#if SC_ATTN_SCALAR & 2      // row sums as single f32 adds (two chains)
                asm("v_add_f32 %0, %1, %2" : "=v"(psum2[0]) : "v"(psum2[0]), "v"(p2[0]));
                asm("v_add_f32 %0, %1, %2" : "=v"(psum2[1]) : "v"(psum2[1]), "v"(p2[1]));
#else
                psum2 += p2;
#endif
            }
        l_run[qb] = l_run[qb] * alpha + (psum2[0] + psum2[1]);
        if (alpha != 1.0f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[qb][0][i] *= alpha; o[qb][1][i] *= alpha; }
        }
        }   // qb
#endif
        auto pv = [&](auto c4c, bool more) {
            constexpr int c4 = decltype(c4c)::value;
            constexpr int kb = c4 >> 1, hb = c4 & 1;
            bf16x8_t pf[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const u32x4_t pfu = {ppk[qb][kb][hb * 4 + 0], ppk[qb][kb][hb * 4 + 1], ppk[qb][kb][hb * 4 + 2], ppk[qb][kb][hb * 4 + 3]};
                pf[qb] = __builtin_bit_cast(bf16x8_t, pfu);
            }
            if (more) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                u32x2_t lo = vlo[c4 & 1][db], hi = vhi[c4 & 1][db];
                asm volatile("" : "+v"(lo), "+v"(hi));                // valid only after the wait above: pin the use below it
                const u32x4_t both = {lo[0], lo[1], hi[0], hi[1]};
#if defined(SC_ATTN_ABL) && SC_ATTN_ABL == 4
                asm volatile("" :: "v"(both), "v"(pf[0]));
                o[0][db][c4] += 1.0f;
#else
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    o[qb][db] = mfma_32x32x16<F16>(__builtin_bit_cast(bf16x8_t, both), pf[qb], o[qb][db]);
#endif
            }
        };
        asm volatile("" :: "v"(l_run[0]));
        if (TRACE) stamp(tr_sm);
        if (!(SC_ATTN_VEARLY && !PP)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // none of the compiler's own LDS traffic outstanding
        if (PP) {      // interval boundary b
            if (!gy) wait_stage(j + 2 < nkv);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!gy && j + 3 < nkv) stage(j + 3, (SLOT + 3) % NST);
        }
        if (!(SC_ATTN_VEARLY && !PP)) issue_v(I0{});
        issue_v(I1{}); pv(I0{}, true);
        issue_v(I2{}); pv(I1{}, true);
        issue_v(I3{}); pv(I2{}, true);
        pv(I3{}, false);
        // tile j+1 must have landed before anyone reads it; tile j+2 (just issued) stays in flight across the barrier
        asm volatile("" :: "v"(o[0][0][0]), "v"(o[0][1][0]));
        if (TRACE) stamp(tr_pv);
        if (!PP) {
        wait_stage(j + 2 < nkv);
#if !(defined(SC_ATTN_NOBAR) && SC_ATTN_NOBAR)     // timing probe (garbage results): no per-tile barrier -- what the waves' lock-step costs
        __builtin_amdgcn_s_barrier();
#endif
        }
        asm volatile("" ::: "memory");
        stamp(tr_bar);
    };
    for (int j = 0; j < nkv; j += NST) {
        tile_body(j, std::integral_constant<int, 0>{});
        if (j + 1 < nkv) tile_body(j + 1, std::integral_constant<int, 1>{});
        if (j + 2 < nkv) tile_body(j + 2, std::integral_constant<int, 2>{});
        if (NST == 4 && j + 3 < nkv) tile_body(j + 3, std::integral_constant<int, 3 % NST>{});
    }
    if (PP && nkv > 0) {
        if (!gy) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // the early half's share of the late half's last interval
        if (ipb > 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } // ring free for the next id
    }

    if (TRACE && trace && lane == 0) {     // one row per wave: [id][wave][8]
        unsigned long long* tr = trace + ((size_t)id * NW + wave) * 8;
        tr[0] = tr_qk; tr[1] = tr_sm; tr[2] = tr_pv; tr[3] = tr_bar; tr[4] = nkv; tr[5] = tr_start; tr[6] = __builtin_amdgcn_s_memrealtime();
        tr[7] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
    const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[qb]), __float_as_uint(l_run[qb]), false, false);
    const float l_tot = __uint_as_float(lsw[0]) + __uint_as_float(lsw[1]);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    {
        // A row is split across the half-waves: lane (g, ql) holds columns 8 rq + 4 g .. + 3 of every 8-column group rq.  v_permlane32_swap between
        // the groups rq = 2k (vdst) and 2k + 1 (src) leaves 16 contiguous bytes in every lane (lower half: columns 16k .. 16k+7, upper half: 16k+8 ..
        // 16k+15), so the row goes out as 4 dwordx4 stores per lane instead of 8 dwordx2: the store tail is issue-bound per instruction
        // (MI355X_MICROARCH.md, T21).  The swaps run in every lane (outside the row test: both halves of a row take the same branch anyway).
        bf16_t* orow = out + (row_base + qrow_c[qb]) * ld_out + hoff;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int ra = (2 * kp) * 4, rb = (2 * kp + 1) * 4;
                const unsigned ax = pack2x<F16>(o[qb][db][ra + 0] * inv, o[qb][db][ra + 1] * inv), ay = pack2x<F16>(o[qb][db][ra + 2] * inv, o[qb][db][ra + 3] * inv);
                const unsigned bx = pack2x<F16>(o[qb][db][rb + 0] * inv, o[qb][db][rb + 1] * inv), by = pack2x<F16>(o[qb][db][rb + 2] * inv, o[qb][db][rb + 3] * inv);
                const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (qrow[qb] < T) *(uint4*)(orow + db * 32 + kp * 16 + g * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
    }
    }   // qb
    }   // id loop (every wave passed the last tile's barrier: the ring is free for the next id)
}

// ------------------------------------------------------------------------------------------------
// CLS-rows-only attention.  Keys/values: first the NQ CLS tokens (batch independent, rows of
// cls_qkv = [q | k | v] each D wide), then frames t < lens[b] of kv_x ([k | v], row stride ld_kv).
__global__ __launch_bounds__(256) void cls_attn_kernel(const bf16_t* __restrict__ cls_qkv, const bf16_t* __restrict__ kv_x, int64_t ld_kv,
                                                       const int32_t* __restrict__ lens, bf16_t* __restrict__ out, int T, int NQ, int H,
                                                       int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = H * hd;
    float* qs = (float*)smem;               // [NQ][hd]
    float* sc = qs + NQ * hd;               // [NQ][NQ + T]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const int nkeys = NQ + len;
    const int skeys = NQ + T;

    for (int i = tid; i < NQ * hd; i += 256) {
        int qi = i / hd, d = i - qi * hd;
        qs[i] = bf2f(cls_qkv[(int64_t)qi * 3 * D + h * hd + d]) * scale;
    }
    __syncthreads();
    // scores: one wave per key
    for (int kk = wave; kk < nkeys; kk += 4) {
        const bf16_t* kr = kk < NQ ? cls_qkv + (int64_t)kk * 3 * D + D + h * hd
                                   : kv_x + ((int64_t)b * T + (kk - NQ)) * ld_kv + h * hd;
        float part[8];
#pragma unroll
        for (int qi = 0; qi < 8; ++qi) part[qi] = 0.f;
        for (int e = lane * 4; e < hd; e += 256) {
            const uint2 u = *(const uint2*)(kr + e);
            const float k0 = lo2f(u.x), k1 = hi2f(u.x), k2 = lo2f(u.y), k3 = hi2f(u.y);
#pragma unroll
            for (int qi = 0; qi < 8; ++qi)
                if (qi < NQ) {
                    const float* qq = qs + qi * hd + e;
                    part[qi] += qq[0] * k0 + qq[1] * k1 + qq[2] * k2 + qq[3] * k3;
                }
        }
#pragma unroll
        for (int qi = 0; qi < 8; ++qi)
            if (qi < NQ) {
                float t = wave_sum(part[qi]);
                if (lane == 0) sc[qi * skeys + kk] = t;
            }
    }
    __syncthreads();
    // softmax: wave w handles queries w, w+4
    for (int qi = wave; qi < NQ; qi += 4) {
        float* row = sc + qi * skeys;
        float mx = -INFINITY;
        for (int kk = lane; kk < nkeys; kk += 64) mx = fmaxf(mx, row[kk]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int kk = lane; kk < nkeys; kk += 64) { float e = __expf(row[kk] - mx); row[kk] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int kk = lane; kk < nkeys; kk += 64) row[kk] *= inv;
    }
    __syncthreads();
    // out[q][d] = sum_k p[q][k] V[k][d]; thread handles 2 adjacent d
    for (int d = tid * 2; d < hd; d += 512) {
        float acc0[8], acc1[8];
#pragma unroll
        for (int qi = 0; qi < 8; ++qi) { acc0[qi] = 0.f; acc1[qi] = 0.f; }
        for (int kk = 0; kk < nkeys; ++kk) {
            const bf16_t* vr = kk < NQ ? cls_qkv + (int64_t)kk * 3 * D + 2 * D + h * hd
                                       : kv_x + ((int64_t)b * T + (kk - NQ)) * ld_kv + D + h * hd;
            const uint32_t u = *(const uint32_t*)(vr + d);
            const float x0 = lo2f(u), x1 = hi2f(u);
#pragma unroll
            for (int qi = 0; qi < 8; ++qi)
                if (qi < NQ) {
                    const float pw = sc[qi * skeys + kk];
                    acc0[qi] += pw * x0;
                    acc1[qi] += pw * x1;
                }
        }
#pragma unroll
        for (int qi = 0; qi < 8; ++qi)
            if (qi < NQ) *(uint32_t*)(out + ((int64_t)b * NQ + qi) * D + h * hd + d) = pack2bf(acc0[qi], acc1[qi]);
    }
}

// ------------------------------------------------------------------------------------------------
// Algebraic CLS pooling.  For a learned query q and head h the score of key x is
//   scale * Q_{q,h} . (Wk_h x + bk_h) = x . u_r + beta_r   with u_r = scale * Wk_h^T Q_{q,h}  (r = (q,h), R = NQ*H <= 8)
// and the head output is  Wv_h (sum_keys p_r x) + bv_h.  So neither K nor V of the frames is ever formed.  The frame scores
// x . u_r + beta_r are one skinny GEMM ([B*T, D] x [D, R], done by sc_gemm_bf16 with bias = beta, f32 out); this kernel does the
// softmax over [NQ CLS tokens ; frames t < lens[b]] and the R probability-weighted frame sums xbar_r in one streaming pass over x.
// Block per utterance; wave w owns keys kk = w (mod 4) with 4 keys in flight; per-wave partial sums are combined through LDS.
// SPLIT (nblk = 2 / 3): xbar is written as bf16 [B, R, nblk*D] = (hi | lo [| hi]) with hi + lo = the fp32 sum to ~16 bits, the operand layout of a
// depth-nblk*D GEMM against [W | W] or [W_hi | W_hi | W_lo] (sc_split_hilo_bf16's convention): the pooled vector keeps fp32-grade precision on
// the bf16 MFMA GEMM that follows (round 4: one bf16 rounding of the pooled vector costs 0.013 of centred cosine on the T = 499 white-noise
// batch, where utterances differ by 1e-2 of the norm).
template <int DCH, bool SPLIT>   // DCH = ceil(D / 256) chunks of 4 elements per lane
__global__ __launch_bounds__(256) void cls_pool_kernel(const bf16_t* __restrict__ x, int64_t ld_x, const bf16_t* __restrict__ cls_tok,
                                                       const float* __restrict__ scores, const float* __restrict__ cls_scores,
                                                       const int32_t* __restrict__ lens, bf16_t* __restrict__ xbar, int T, int NQ, int R, int D, int nblk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;                 // [R][NQ + T]  probabilities
    float* red = sc + 8 * (NQ + T);           // [4 waves][R][D] partial sums (only rows < R used)
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
        float mx = -INFINITY;
        for (int kk = lane; kk < nkeys; kk += 64) mx = fmaxf(mx, row[kk]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int kk = lane; kk < nkeys; kk += 64) { const float e = __expf(row[kk] - mx); row[kk] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int kk = lane; kk < nkeys; kk += 64) row[kk] *= inv;
    }
    __syncthreads();
    float acc[8][DCH][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < DCH; ++c) acc[r][c][0] = acc[r][c][1] = acc[r][c][2] = acc[r][c][3] = 0.f;
    for (int k0 = wave; k0 < nkeys; k0 += 16) {
        uint2 xv[4][DCH];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 4 * u;
            const int kc = kk < nkeys ? kk : nkeys - 1;
            const bf16_t* xr = kc < NQ ? cls_tok + (int64_t)kc * D : x + ((int64_t)b * T + (kc - NQ)) * ld_x;
#pragma unroll
            for (int c = 0; c < DCH; ++c) xv[u][c] = (c * 256 + lane * 4 < D) ? *(const uint2*)(xr + c * 256 + lane * 4) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 4 * u;
            if (kk >= nkeys) break;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r >= R) break;
                const float pw = sc[r * skeys + kk];
#pragma unroll
                for (int c = 0; c < DCH; ++c) {
                    acc[r][c][0] += pw * lo2f(xv[u][c].x); acc[r][c][1] += pw * hi2f(xv[u][c].x);
                    acc[r][c][2] += pw * lo2f(xv[u][c].y); acc[r][c][3] += pw * hi2f(xv[u][c].y);
                }
            }
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
    for (int i = tid * 2; i < R * D; i += 512) {
        const int r = i / D, d = i - r * D;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s0 += red[(w * 8 + r) * D + d]; s1 += red[(w * 8 + r) * D + d + 1]; }
        if (SPLIT) {
            const uint32_t hi = pack2bf(s0, s1);
            bf16_t* o = xbar + ((int64_t)b * R + r) * nblk * D + d;
            *(uint32_t*)o = hi;
            *(uint32_t*)(o + D) = pack2bf(s0 - lo2f(hi), s1 - hi2f(hi));
            if (nblk == 3) *(uint32_t*)(o + 2 * D) = hi;
        } else {
            *(uint32_t*)(xbar + ((int64_t)b * R + r) * D + d) = pack2bf(s0, s1);
        }
    }
}

}  // namespace

static bool g_attn_trace_host = false;
extern "C" void sc_debug_set_attn_trace(void* dev_buf) {     // per-block [8] u64: QK / softmax / PV / wait cycles, tiles, start, end (100 MHz)
    unsigned long long* p = (unsigned long long*)dev_buf;
    g_attn_trace_host = SC_PROBES && dev_buf != nullptr;      // the product library instantiates no TRACE variant (`make PROBES=1`)
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &p, sizeof(p));
}

static int attention_fwd_impl(const void* q, const void* k, const void* v, void* out, const int32_t* klens, int B, int H, int T, int head_dim,
                              int64_t ld_qkv, int64_t ld_out, float scale, int flags, float drop_p, uint32_t seed, void* stream,
                              const int32_t* row_off = nullptr, int64_t total_rows = 0) {
    const int causal = flags & SC_ATTN_CAUSAL;
    const bool f16 = (flags & SC_ATTN_F16) != 0;
    SC_CHECK_ARG((flags & ~(SC_ATTN_CAUSAL | SC_ATTN_F16)) == 0, "sc_attention_fwd: unknown flag bits 0x%x", flags);
    SC_CHECK_ARG(!f16 || drop_p == 0.f, "sc_attention_fwd: SC_ATTN_F16 has no dropout form (the frozen pre-LN encoder's rates are 0)");
    SC_CHECK_ARG(head_dim == 64, "sc_attention_fwd: head_dim=%d unsupported (64 only; use sc_cls_attention_fwd for pooling heads)", head_dim);
    SC_CHECK_ARG(ld_qkv % 8 == 0 && ld_out % 8 == 0, "sc_attention_fwd: ld_qkv and ld_out must be multiples of 8 (16-byte rows)");
    SC_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "sc_attention_fwd: misaligned pointers");
    SC_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "sc_attention_fwd_dropout: drop_p=%f must be in [0, 1)", (double)drop_p);
    SC_CHECK_ARG(drop_p == 0.f || (row_off ? total_rows * H * ((T + 1) / 2) : (int64_t)B * H * T * T) < 0xffffffffLL,
                 "sc_attention_fwd_dropout: B*H*T*T must fit 32 bits (mask element index)");
    if (B <= 0 || T <= 0) return 0;
    // 8-wave blocks (256 query rows share the K/V ring) when there are at least 256 queries; SC_ATTN_NW=4 forces the 4-wave form
    static const int force_nw = SC_ATTN_ENV_INT("SC_ATTN_NW", 0);
    // SC_ATTN_QB=2 (A/B): 4 waves x 64 query rows instead of 8 waves x 32 for the long-sequence form
    static const int qb_env = SC_ATTN_ENV_INT("SC_ATTN_QB", 1);
    const bool qb2 = (qb_env == 2 || qb_env == 3) && T > 128 && !force_nw && drop_p == 0.f && !g_attn_trace_host && !f16;
    const int nw = force_nw ? force_nw : (qb2 ? (qb_env == 3 ? 8 : 4) : (T > 128 ? 8 : 4));
    const int64_t units8 = ((int64_t)H * B + 7) / 8;
    static const int ipb_env = SC_ATTN_ENV_INT("SC_ATTN_IPB", 0);
    int ipb = ipb_env > 0 ? ipb_env : 1;
    ipb = ipb < 1 ? 1 : (ipb > 16 ? 16 : ipb);
    const uint32_t th = drop_thresh16(drop_p);         // 16 random bits per probability: see hash_pair (common.h)
    const float ks = 1.0f / (1.0f - drop_p);
    // Query-row split (see q_mode in the kernel): the 8-wave launch keeps the blocks with more than 128 valid rows, a second launch of 4-wave blocks takes every
    // last block with <= 128.  Uniform T only, when T % 256 is in [1, 128] (T = 499 / 500, the headline, is one launch as before): training step (T = 319) 33.41 -> 32.86 ms in
    // 3 of 3 interleaved passes, P-large (ViT-L/14, T = 257) +-0.1 ms.  PACKED batches keep one launch: the per-utterance form of the same split (kernel side kept, q_mode 1 / 2
    // with row_off) measured +0.25 ms on the ragged P-base step (27.55 vs 27.29 ms) -- the 4-wave tail blocks re-read K / V that the 8-wave block shares.
    // profiles/r06_attention_query_split_ab.txt.  SC_ATTN_SPLIT=0: one launch (A/B).
    static const int split_env = getenv("SC_ATTN_SPLIT") ? atoi(getenv("SC_ATTN_SPLIT")) : 1;
    const bool split = split_env && nw == 8 && !causal && !qb2 && !g_attn_trace_host && !force_nw && !row_off && (T & 255) >= 1 && (T & 255) <= 128;
    int rc_launch = 0;
    auto launch = [&](int nw_, int nq, int q_mode) {
        if (nq <= 0) return;
        const int lds = ((SC_ATTN_PP && nw_ == 8) ? 4 : NSTAGE) * STAGE_BYTES;
        if (units8 * 8 * nq >= 0x7fffffff) { sc_set_error("sc_attention_fwd: grid too large"); rc_launch = -1; return; }
        const int n_ids = (int)(units8 * 8 * nq);
        const int groups8 = (n_ids / 8 + ipb - 1) / ipb;           // n_ids is a multiple of 8
        dim3 grid((unsigned)(groups8 * 8));
        const int cflag = causal | (q_mode << 1);
#define ATTN_LAUNCH(NW_, TR_, DR_, ...)                                                                                                     \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<NW_, TR_, DR_, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);            \
        hipLaunchKernelGGL((attn_fwd_kernel<NW_, TR_, DR_, ##__VA_ARGS__>), grid, dim3(NW_ * 64), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, \
                           (const bf16_t*)v, (bf16_t*)out, klens, T, ld_qkv, ld_out, scale * 1.44269504088896341f, cflag, B, H, nq, n_ids, ipb, \
                           seed, th, ks, row_off);                                                                                          \
    } while (0)
        if (f16) { if (nw_ == 8) ATTN_LAUNCH(8, false, false, 1, true); else ATTN_LAUNCH(4, false, false, 1, true); }
        else
#if SC_PROBES
        if (qb2) { if (nw_ == 8) ATTN_LAUNCH(8, false, false, 2); else ATTN_LAUNCH(4, false, false, 2); }
        else
#endif
        if (th) { if (nw_ == 8) ATTN_LAUNCH(8, false, true); else ATTN_LAUNCH(4, false, true); }
#if SC_PROBES
        else if (g_attn_trace_host) { if (nw_ == 8) ATTN_LAUNCH(8, true, false); else ATTN_LAUNCH(4, true, false); }
#endif
        else if (nw_ == 8) ATTN_LAUNCH(8, false, false);
        else ATTN_LAUNCH(4, false, false);
    };
    const int rows = nw * 32 * (qb2 ? 2 : 1);
    if (!split) launch(nw, (T + rows - 1) / rows, 0);
    else {
        launch(8, T / 256, 1);      // the full 256-row blocks
        launch(4, 1, 2);
    }
    if (rc_launch) return rc_launch;
#undef ATTN_LAUNCH
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_attention_fwd(const void* q, const void* k, const void* v, void* out, const int32_t* klens, int B, int H,
                                int T, int head_dim, int64_t ld_qkv, int64_t ld_out, float scale, int causal, void* stream) {
    return attention_fwd_impl(q, k, v, out, klens, B, H, T, head_dim, ld_qkv, ld_out, scale, causal, 0.f, 0u, stream);
}

// Train-mode attention of the FROZEN encoder: the same kernel with dropout on the attention probabilities (fairseq MultiheadAttention
// dropout_module, attention_dropout of the checkpoint) -- Lightning's model.train() switches the frozen HuBERT's dropouts on too.
extern "C" int sc_attention_fwd_dropout(const void* q, const void* k, const void* v, void* out, const int32_t* klens, int B, int H, int T, int head_dim,
                                        int64_t ld_qkv, int64_t ld_out, float scale, int causal, float drop_p, uint32_t seed, void* stream) {
    return attention_fwd_impl(q, k, v, out, klens, B, H, T, head_dim, ld_qkv, ld_out, scale, causal, drop_p, seed, stream);
}

// Packed (padding-free) batches: utterance b owns rows [row_off[b], row_off[b + 1]) of q / k / v / out (row_off: B + 1 device ints,
// rows per utterance <= Tmax); klens[b] <= its row count.  drop_p > 0: the train-mode form.
extern "C" int sc_attention_fwd_packed(const void* q, const void* k, const void* v, void* out, const int32_t* klens, const int32_t* row_off, int B, int H,
                                       int Tmax, int64_t total_rows, int head_dim, int64_t ld_qkv, int64_t ld_out, float scale, float drop_p, uint32_t seed,
                                       int flags, void* stream) {
    SC_CHECK_ARG((flags & SC_ATTN_CAUSAL) == 0, "sc_attention_fwd_packed: no causal form");
    SC_CHECK_ARG(row_off != nullptr && klens != nullptr, "sc_attention_fwd_packed: row_off and klens are required");
    return attention_fwd_impl(q, k, v, out, klens, B, H, Tmax, head_dim, ld_qkv, ld_out, scale, flags, drop_p, seed, stream, row_off, total_rows);
}

extern "C" int sc_cls_attention_fwd(const void* cls_qkv, const void* kv_x, int64_t ld_kv, const int32_t* lens, void* out, int B, int T,
                                    int NQ, int H, int head_dim, float scale, void* stream) {
    SC_CHECK_ARG(NQ >= 1 && NQ <= 8, "sc_cls_attention_fwd: NQ=%d must be in [1,8]", NQ);
    SC_CHECK_ARG(head_dim % 4 == 0 && head_dim <= 1024, "sc_cls_attention_fwd: head_dim=%d must be a multiple of 4, <= 1024", head_dim);
    SC_CHECK_ARG(ld_kv % 4 == 0, "sc_cls_attention_fwd: ld_kv must be a multiple of 4");
    if (B <= 0) return 0;
    const int lds = (NQ * head_dim + NQ * (NQ + T)) * 4;
    SC_CHECK_ARG(lds <= 160 * 1024, "sc_cls_attention_fwd: T=%d too long for the LDS score buffer", T);
    (void)hipFuncSetAttribute((const void*)cls_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(cls_attn_kernel, dim3(H, B), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)cls_qkv, (const bf16_t*)kv_x, ld_kv,
                       lens, (bf16_t*)out, T, NQ, H, head_dim, scale);
    SC_CHECK_LAUNCH();
    return 0;
}

static int cls_pool_launch(const void* x, int64_t ld_x, const void* cls_tok, const float* scores, const float* cls_scores,
                           const int32_t* lens, void* xbar, int B, int T, int NQ, int R, int D, int nblk, void* stream) {
    const bool split = nblk > 1;
    SC_CHECK_ARG(nblk >= 1 && nblk <= 3, "sc_cls_pool_fwd_split: nblk=%d (2: hi|lo, 3: hi|lo|hi)", nblk);
    SC_CHECK_ARG(NQ >= 1 && R >= NQ && R <= 8, "sc_cls_pool_fwd: need 1 <= NQ <= R <= 8 (NQ=%d R=%d)", NQ, R);
    SC_CHECK_ARG(D > 0 && D <= 1024 && D % 4 == 0 && ld_x % 4 == 0, "sc_cls_pool_fwd: D=%d must be a multiple of 4, <= 1024", D);
    if (B <= 0) return 0;
    const int lds = (8 * (NQ + T) + 4 * 8 * D) * 4;
    SC_CHECK_ARG(lds <= 160 * 1024, "sc_cls_pool_fwd: T=%d / D=%d too large for LDS", T, D);
    hipStream_t s = (hipStream_t)stream;
#define SC_POOL_LAUNCH(DCH, SPL)                                                                                                         \
    do {                                                                                                                                 \
        (void)hipFuncSetAttribute((const void*)cls_pool_kernel<DCH, SPL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);              \
        hipLaunchKernelGGL((cls_pool_kernel<DCH, SPL>), dim3(B), dim3(256), lds, s, (const bf16_t*)x, ld_x, (const bf16_t*)cls_tok, scores, \
                           cls_scores, lens, (bf16_t*)xbar, T, NQ, R, D, nblk);                                                          \
    } while (0)
#define SC_POOL_BY_D(SPL)                   \
    do {                                    \
        if (D <= 256) SC_POOL_LAUNCH(1, SPL);      \
        else if (D <= 512) SC_POOL_LAUNCH(2, SPL); \
        else if (D <= 768) SC_POOL_LAUNCH(3, SPL); \
        else SC_POOL_LAUNCH(4, SPL);               \
    } while (0)
    if (split) SC_POOL_BY_D(true);
    else SC_POOL_BY_D(false);
#undef SC_POOL_BY_D
#undef SC_POOL_LAUNCH
    SC_CHECK_LAUNCH();
    return 0;
}

extern "C" int sc_cls_pool_fwd(const void* x, int64_t ld_x, const void* cls_tok, const float* scores, const float* cls_scores,
                               const int32_t* lens, void* xbar, int B, int T, int NQ, int R, int D, void* stream) {
    return cls_pool_launch(x, ld_x, cls_tok, scores, cls_scores, lens, xbar, B, T, NQ, R, D, 1, stream);
}

extern "C" int sc_cls_pool_fwd_split(const void* x, int64_t ld_x, const void* cls_tok, const float* scores, const float* cls_scores,
                                     const int32_t* lens, void* xbar_hilo, int B, int T, int NQ, int R, int D, int nblk, void* stream) {
    SC_CHECK_ARG(nblk == 2 || nblk == 3, "sc_cls_pool_fwd_split: nblk=%d (2: hi|lo, 3: hi|lo|hi)", nblk);
    return cls_pool_launch(x, ld_x, cls_tok, scores, cls_scores, lens, xbar_hilo, B, T, NQ, R, D, nblk, stream);
}
