// bf16 MFMA GEMM with fused epilogues for gfx950 (CDNA4).
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual
// A rows may overlap (lda < K): that is how the HuBERT conv stack (channels-last activations,
// kernel k, stride s => K = k*C, lda = s*C) runs as a plain GEMM without im2col.
//
// Structure (v1): BM x BN x 64 tiles, 4 waves (2x2), 16x16x32 bf16 MFMA, operands staged with
// LDS-DMA (global_load_lds_dwordx4) into a double-buffered, XOR-swizzled LDS image
// (swizzle applied on the SOURCE address, LDS destination stays lane-linear), one barrier per
// K-tile, XCD-aware tile order (all N-tiles of an M-panel run on one XCD so the A panel is an
// L2 hit).  The MFMA is issued with W as the "A operand" so that each lane ends up holding four
// consecutive output columns of one row => 8-byte epilogue stores.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "gemm8p.h"
#include "../../include/speechclip_hip.h"

namespace {

struct GemmParams {
    const bf16_t* A; int64_t lda; int64_t strideA;
    const bf16_t* W; int64_t ldw; int64_t strideW; int w_mod;
    void* C; int64_t ldc; int64_t strideC;
    // two-level batches (sc_gemm_bf16_batched2): z = zo * inner + zi; operand offset = zo * stride + zi * stride2 (inner = 0: one level)
    int inner; int64_t strideA2, strideW2, strideC2;
    int nbatch;                    // gemm256 BATCH variant: independent products z < nbatch (A + z strideA, W + (z % w_mod) strideW, C + z strideC)
    const float* bias;
    const void* residual; int64_t ldr;
    int64_t M; int N; int K;
    int tiles_m; int tiles_n;
    int act; int out_f32;
    int f16;                     // operands (and 16-bit outputs / residuals) are IEEE half instead of bf16: gemm_bf16_kernel<.., true> only (SC_GEMM_F16)
    unsigned long long* trace;   // debug: per-block s_memtime stamps (sc_debug_set_gemm_trace)
    int rot;                     // rotate the K loop per block (L2 channel de-correlation)
    int band;                    // N-tiles per column band of the persistent tile order (0/>=tiles_n: M-panel-major over all of N)
    int epi_mode, epi_mode_res;                // next tile's first two stages: 0 issued before the epilogue, 2 interleaved with its stores (default), 3 after it
    // LayerNorm folded into the GEMMs around it (sc_gemm_bf16_ln; post-LN transformer layers, eval path):
    int stagger;                               // PROBES: block group (b >> 3) & 3 sleeps g * stagger * 4096 cycles before its first tile (de-synchronised epilogues)
    int eprobe;                                // PROBES, fast epilogue only (garbage results): 1 no stores, 2 no next-tile DMA pieces, 4 stores wrap inside 2 MiB of C, 8 no shuffles;
                                               // 64 (valid results) streaming stores
    int kpair;                                 // > 0: stride-2 kernel-3 conv as GEMM (K = 3C, lda = 2C), kpair = C / 64: walk K as (tap0 c, tap2 c) pairs, then tap1
};

constexpr int BK = 64;  // 128 bytes of bf16 per tile row = 8 chunks of 16 B

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// Timing probes (SC_GEMM_ABL ablations, the per-phase s_memtime trace, the 2-slot-ring A/B switch) are compiled only into the PROBES build
// (`make PROBES=1` -> libspeechclip_hip_probes.so, -DSC_PROBES=1); the product library instantiates the kernel without them.  The operand
// cache-policy (nt / sc bits on the LDS-DMA pieces), buffer-form DMA, wave-priority and DMA-placement variants measured in rounds 1-2 are
// recorded in EXPERIMENTS.md (rounds 1-4, old section 3.1) with their numbers and no longer live in this file.
#ifndef SC_PROBES
#define SC_PROBES 0
#endif
#ifndef SC_GEMM_RES_TOUCH         // 1: the residual variants TOUCH their tile's residual lines (4-byte LDS-DMA loads into the idle A slot) during k-step nk - 2, so that the
#define SC_GEMM_RES_TOUCH 1       // epilogue's residual loads hit in L2 instead of paying the HBM / MALL latency with the matrix pipe idle (round 4)
#endif
#ifndef SC_GEMM_FAST_EPI          // 0: every tile takes the general (predicated) epilogue; 1: fast path for the variants without a residual operand;
#define SC_GEMM_FAST_EPI 1        // 2: also for the residual variants (A/B builds)
#endif
// host-side A/B knobs (tile order, epilogue mode, dispatcher thresholds): environment variables in the PROBES build, constants in the product
#if SC_PROBES
#define SC_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define SC_TUNE_SET(name) (getenv(name) != nullptr)
#else
#define SC_TUNE_INT(name, dflt) (dflt)
#define SC_TUNE_SET(name) false
#endif

// Stage a ROWS x 64 bf16 tile: LDS image is [row][8 chunks of 16 B], chunk position p of row r holds
// global k-chunk (p ^ (r & 7)).  256 threads => 32 rows per pass.
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

template <int ROWS>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int64_t ld, int64_t row0, int64_t row_max,
                                           int k0, char* lds, int tid, int wave) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        int c = i * 256 + tid;
        int r = c >> 3, p = c & 7;
        int64_t gr = row0 + r;
        gr = gr < row_max ? gr : row_max;  // clamp: out-of-range rows re-read the last valid row
        const bf16_t* src = g + gr * ld + k0 + ((p ^ (r & 7)) << 3);
        glds16(src, lds + (i * 256 + wave * 64) * 16);
    }
}

template <int BM, int BN, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr int WM = BM / 2, WN = BN / 2;  // wave tile
    constexpr int MI = WM / 16, NI = WN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto sA = [&](int b) -> char* { return smem + b * (BM * 128); };
    auto sB = [&](int b) -> char* { return smem + 2 * BM * 128 + b * (BN * 128); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap (bijective for any grid size): blocks b, b+8, b+16.. share an XCD.
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int v = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int z = blockIdx.y;

    const int zo = p.inner > 0 ? z / p.inner : z, zi = p.inner > 0 ? z - zo * p.inner : 0;
    const bf16_t* A = p.A + (int64_t)zo * p.strideA + (int64_t)zi * p.strideA2;
    const bf16_t* W = p.inner > 0 ? p.W + (int64_t)zo * p.strideW + (int64_t)zi * p.strideW2 : p.W + (int64_t)(z % p.w_mod) * p.strideW;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage_tile<BM>(A, p.lda, m0, p.M - 1, 0, sA(0), tid, wave);
    stage_tile<BN>(W, p.ldw, n0, p.N - 1, 0, sB(0), tid, wave);

    const int frow = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            stage_tile<BM>(A, p.lda, m0, p.M - 1, (kt + 1) * BK, sA(cur ^ 1), tid, wave);
            stage_tile<BN>(W, p.ldw, n0, p.N - 1, (kt + 1) * BK, sB(cur ^ 1), tid, wave);
        }
        const char* a_base = sA(cur) + (wm * WM) * 128;
        const char* b_base = sB(cur) + (wn * WN) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int row = i * 16 + frow;
                af[i] = *(const bf16x8_t*)(a_base + row * 128 + (((kk * 4 + fk) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int row = j * 16 + frow;
                bfr[j] = *(const bf16x8_t*)(b_base + row * 128 + (((kk * 4 + fk) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = mfma_16x16x32<F16>(bfr[j], af[i], acc[i][j]);
        }
    }

    // epilogue: lane holds C[m = .. + (lane & 15)][n = .. + 4*(lane>>4) + r], r = 0..3
    const float* bias = p.bias ? p.bias + (int64_t)(z % p.w_mod) * p.N : nullptr;
    char* Cb = (char*)p.C + ((int64_t)zo * p.strideC + (int64_t)zi * p.strideC2) * (p.out_f32 ? 4 : 2);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wm * WM + i * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 16 + fk * 4;
            if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
            f32x4_t v4 = acc[i][j];
            if (bias) {
                const f32x4_t b4 = *(const f32x4_t*)(bias + n);
                v4 += b4;
            }
            if (p.act == SC_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = (p.out_f32 || F16) ? gelu_erf_precise(v4[r]) : gelu_erf(v4[r]);   // f32 consumers get the 1.5e-7 erf path
            } else if (p.act == SC_ACT_QUICKGELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
            }
            if (p.out_f32) {
                if (p.residual) v4 += *(const f32x4_t*)((const float*)p.residual + m * p.ldr + n);
                *(f32x4_t*)((float*)Cb + m * p.ldc + n) = v4;
            } else {
                if (p.residual) {
                    const uint2 rr = *(const uint2*)((const bf16_t*)p.residual + m * p.ldr + n);
                    v4[0] += lo2fx<F16>(rr.x); v4[1] += hi2fx<F16>(rr.x); v4[2] += lo2fx<F16>(rr.y); v4[3] += hi2fx<F16>(rr.y);
                }
                uint2 o;
                o.x = pack2x<F16>(v4[0], v4[1]);
                o.y = pack2x<F16>(v4[2], v4[3]);
                *(uint2*)((bf16_t*)Cb + m * p.ldc + n) = o;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// v2: 256 x 256 x 64 tiles, 8 waves (2 x 4, wave tile 128 x 64), one resident block per CU.
//  * 128^2 tiles need ~64 B/clk/CU from L2 at full MFMA rate (more than the chip delivers); 256^2 halves that.
//  * operands arrive by LDS-DMA in FULL 128-byte rows (BK = 64; 64-byte rows measured ~25 % slower on the load path)
//    into a 2-slot ring (2 x 64 KiB); chunk position p of row r holds k-chunk p ^ (r & 7) (swizzle on the source
//    address), which makes the ds_read_b128 fragment reads conflict-free.
//  * register software pipeline at half-step (32-deep) granularity: while the 32 MFMAs of one half issue, the
//    fragments of the next half are read ({4 MFMA, 1 ds_read} x 8), so LDS latency never gates the matrix pipe.
//    Slot t is completely in registers once its second half starts, so it is refilled (stage t+2) from the
//    mid-step barrier: ONE raw s_barrier per 64-deep k-step, loads stay in flight across it (counted vmcnt).
//  * epilogue: bias/activation in registers -> bf16 -> wave-private LDS image -> full-row 16-byte stores
//    (the direct fragment-shaped store is 32 x 8-byte stores per lane touching 16 lines each: issue-bound).
constexpr int BK2 = 64;
constexpr int SLOT_BYTES = 2 * 256 * BK2 * 2;  // 64 KiB: A [256][128 B] then B [256][128 B]

// Source addressing of one tile: two WAVE-UNIFORM tile base pointers (SGPRs) + two 32-bit per-lane offsets fixed for the whole
// kernel.  (Per-lane 64-bit pointers kept across the tile were being spilled, and a scratch reload behind freshly issued LDS-DMA
// forces vmcnt(0): it drained the prefetch.)
// Tiles that would cross M (or N) are shifted back so that they END at M (N): every row of a tile is then in range, the row-group
// offsets i*64*ld are uniform (no per-lane clamp), and the epilogue skips the rows / columns that belong to the previous tile.
// Needs M >= 256 and N >= 256, which the dispatcher guarantees for this kernel.
struct StageAddr { const bf16_t* ta; const bf16_t* tw; };

__device__ __forceinline__ void stage256(const StageAddr& sa, int lane_a, int lane_w, int64_t lda64, int64_t ldw64, int k0, char* slot, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(sa.ta + (i * lda64 + k0) + lane_a, slot + (i * 512 + wave * 64) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(sa.tw + (i * ldw64 + k0) + lane_w, slot + 256 * 128 + (i * 512 + wave * 64) * 16);
}

// RING3 (default): the A operand gets a THREE-slot ring, W keeps two (3 x 32 KiB + 2 x 32 KiB = all 160 KiB of LDS), which lets the refill of a
// stage be spread over the WHOLE k-step instead of being packed into its second half: with the two-slot [A|W] ring a slot is free only after the
// mid-step barrier, so its 8 pieces per wave go out during half of the time -- 64 KiB per half k-step = the CU's whole L2 -> LDS path for that
// half, nothing in the other.  Here the A pieces of stage kt+2 go out during the FIRST half of k-step kt (their slot held stage kt-1, free since
// the previous barrier) and the W pieces during the second half, one piece every other MFMA group, and the mid-step wait is `vmcnt(4)` (in-order
// retirement: everything but the A pieces just issued).  Round 2, same box: whole step 46.1-46.3 ms vs 47.5 with the two-slot ring (GEMM 953-955
// vs 919-921 TF/s in the step; QKV +4-6 % isolated).  What did NOT help with the third slot: prefetching A a k-step deeper with all 8 pieces still
// in the second half (-6 ... +1 %: the loop is not latency-bound), and 4 + 4 pieces in two bursts (groups 4..7 of each half: -5 ... 0 %).
// (Rounds 2-5 carried two more epilogue forms here -- LayerNorm folded into the GEMMs on both sides of it, `sc_gemm_bf16_ln` -- opt-in and measured neutral to
//  negative in the step (the HBM-bound LayerNorm launches are where the image tower's side stream overlaps): removed in round 6, EXPERIMENTS.md R6-4.)
// BATCH: the persistent tile list runs over nbatch independent products of the same shape (the split-K partial products of a weight gradient:
// speechclip_amd/train_hubert.py::wgrad); every tile carries its product index, the operands move by a per-product stride.  Plain epilogue only.
template <int ABL, bool TRACE, int ACT, bool RES, bool RING3, bool BATCH = false>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bf16_t* A = p.A;
    const bf16_t* W = p.W;
    const int nk = p.K / BK2;
    // K rotation: the 32 CUs of an XCD would otherwise all stream the SAME W rows (and 2-3 of them the same A rows) at the same
    // instant and pile onto a few L2 channels; block b starts its K loop at chunk rot(b) and wraps (fp32 sum order differs per
    // block, deterministically).  SC_GEMM_NOROT=1 disables it for A/B timing.
    // rot = M-panel index mod nk: the N-tiles of one A panel stay in lock-step (their A lines are fetched once and hit in L2 for
    // the siblings) while different M-panels read different W rows at any instant.
    int rot = 0;
    // Conv-as-GEMM with overlapping rows (k = 3, stride 2: K = 3C, lda = 2C): columns [2C, 3C) of row r ARE columns [0, C) of row r + 1, so
    // chunk c of tap 2 touches the cache lines chunk c of tap 0 touched one row further down.  Walking K as (tap0 c, tap2 c) pairs puts the
    // two touches one k-step apart (an L2 / TCP hit) instead of 16 k-steps apart (evicted: every CU streams 64 KiB per k-step through a 4 MiB
    // L2 shared by 32 CUs) -- the sum over K is order-free.  kpair = C / 64 chunks per tap.
    const int kpair = p.kpair;
    auto kmap = [&](int c) -> int {       // k offset (elements) of the c-th chunk of the walk
        if (kpair > 0) c = c < 2 * kpair ? (c >> 1) + (c & 1) * 2 * kpair : c - kpair;
        return c * BK2;
    };
    auto kofs = [&](int st) -> int {
        int c = st + rot;
        c = c >= nk ? c - nk : c;
        return kmap(c);
    };
    const int frow = lane & 15, fk = lane >> 4;
    const int off_h0 = frow * 128 + ((fk ^ (frow & 7)) << 4);
    const int off_h1 = frow * 128 + (((4 + fk) ^ (frow & 7)) << 4);
    constexpr int HALF_SLOT = 256 * 128;                       // one operand of one stage: 256 rows x 128 B
    const int a_base = wm * 128 * 128;
    const int b_base = (RING3 ? 0 : 256 * 128) + wn * 64 * 128;   // RING3: relative to the W slot; else relative to the combined slot
    // slot addresses: 2-ring = [A|W] x 2 (64 KiB each); 3-ring = A0 A1 A2 W0 W1 (32 KiB each)
    // Ring state: the LDS slots of stages kt, kt+1, kt+2 (A) and kt, kt+1 (W) of the k-step being computed, ROTATED once per k-step and never
    // reset -- the ring does not restart at a tile boundary (stage j of the next tile takes the place stage nk + j of this one would), and the
    // loop carries no modulo arithmetic.  2-ring: A and W share a slot (sW == sA), stage kt+2 goes where stage kt was (sA2 == sA0).
    char* sA0 = smem;
    char* sA1 = smem + (RING3 ? HALF_SLOT : SLOT_BYTES);
    char* sA2 = RING3 ? smem + 2 * HALF_SLOT : sA0;
    char* sW0 = RING3 ? smem + 3 * HALF_SLOT : sA0;
    char* sW1 = RING3 ? smem + 4 * HALF_SLOT : sA1;
    auto rotate_ring = [&]() {
        if (RING3) { char* t = sA0; sA0 = sA1; sA1 = sA2; sA2 = t; t = sW0; sW0 = sW1; sW1 = t; }
        else { char* t = sA0; sA0 = sA1; sA1 = t; sA2 = sA0; sW0 = sA0; sW1 = sA1; }
    };
    auto w_dst = [&](char* wslot) -> char* { return RING3 ? wslot : wslot + HALF_SLOT; };   // where the W pieces of a stage land
    const bool k_counted_wait = p.epi_mode >= 0 && !(p.epi_mode & 0x100);   // SC_GEMM_EPI |= 0x100: always drain at tile start (A/B)
    const bool vec_ok = !p.out_f32 && (p.N % 8 == 0) && (p.ldc % 8 == 0) && (!RES || p.ldr % 8 == 0);
    const bool f32_ok = p.out_f32 && (p.N % 4 == 0) && (p.ldc % 4 == 0) && (!RES || p.ldr % 4 == 0);

    // Persistent: one block per CU walks the tile list (a new 512-thread / 144 KiB block per tile costs several us of
    // dispatch + an exposed prologue).  XCD-aware order: block b runs on XCD b % 8 and takes a contiguous chunk of the
    // tile space, M-panel-major, so the N-tiles of one A panel are L2 hits on the same XCD.
    const int per_product = p.tiles_m * p.tiles_n;
    const int nwg = per_product * (BATCH ? p.nbatch : 1);
    const int G = gridDim.x;
    auto tile_of = [&](int it, int& tm, int& tn, int& tz) -> bool {   // it-th tile of this block (tz: product index, BATCH only)
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot_in_xcd = bid >> 3;
        const int nb_xcd = (G - xcd + 7) >> 3;                  // blocks living on this XCD
        const int q = nwg >> 3, r8 = nwg & 7;
        const int begin = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
        const int cnt = q + (xcd < r8 ? 1 : 0);
        const int idx = it * nb_xcd + slot_in_xcd;
        if (idx >= cnt) return false;
        int v = begin + idx;
        tz = 0;
        if (BATCH) { tz = v / per_product; v -= tz * per_product; }
        // Column bands: the order walks all M-panels of a band of `band` N-tiles before moving to the next band, so the 32 blocks of
        // an XCD work on (32 / band) M-panels x band N-tiles at a time: the band's slice of W (band x 256 x K) stays resident in the
        // XCD's 4 MiB L2 while A streams through once per band.  With all of N in flight (qkv: 9 tiles = 3.5 MiB of W, fc1: 12 = 4.7 MiB)
        // W and the A panels evict each other: measured TCC hit rate 69 %.
        int nb = p.tiles_n, tn0 = 0;
        if (p.band > 0 && p.band < p.tiles_n) {
            const int per_band = p.band * p.tiles_m;
            const int b = v / per_band;
            tn0 = b * p.band;
            v -= b * per_band;
            nb = p.tiles_n - tn0 < p.band ? p.tiles_n - tn0 : p.band;
        }
        tm = v / nb;
        tn = tn0 + (v - tm * nb);
        return true;
    };

    // (no s_setprio anywhere in the k-loop: -0.2 ms per step against priority 1 inside every half-step, 7 of 7 same-box A/B passes, round 2)
    unsigned long long t_begin = TRACE ? __builtin_readcyclecounter() : 0, t_wait = 0, t_loop = 0, t_pre = 0;
    const int64_t lda64 = 64 * p.lda, ldw64 = 64 * p.ldw;
    const int lane_a = (tid >> 3) * (int)p.lda + (((tid & 7) ^ ((tid >> 3) & 7)) << 3);   // row (tid>>3), swizzled k-chunk
    const int lane_w = (tid >> 3) * (int)p.ldw + (((tid & 7) ^ ((tid >> 3) & 7)) << 3);
    auto tile_m0 = [&](int t) -> int64_t { const int64_t m = (int64_t)t * 256; return m + 256 <= p.M ? m : p.M - 256; };
    auto tile_n0 = [&](int t) -> int { const int n = t * 256; return n + 256 <= p.N ? n : p.N - 256; };
    // ABL 8 (timing probe, garbage results): every tile reads its A rows from the first 2048 rows -- distinct lines per k-step, but L2-resident:
    // what the first-touch (MALL / HBM) latency of the A panels costs the loop
    auto a_row0 = [&](int t) -> int64_t { const int64_t m = tile_m0(t); return ABL == 8 ? (m & 2047) : m; };
    int tm, tn, tz;
    bool have = tile_of(0, tm, tn, tz);
    if (SC_PROBES && p.stagger > 0) {
        const int g = (blockIdx.x >> 3) & 3;
        for (int i = 0; i < g * p.stagger; ++i) __builtin_amdgcn_s_sleep(64);
    }
    StageAddr sa{nullptr, nullptr};
    auto set_tile = [&](const bf16_t* ta, const bf16_t* tw) { sa = StageAddr{ta, tw}; };
    // one LDS-DMA piece (64 rows x 128 B... 8 rows per wave): row group g of A / W at k offset k0 (elements) into LDS at dst
    auto piece_a = [&](int g, int k0, char* dst) { glds16(sa.ta + (g * lda64 + k0) + lane_a, dst); };
    auto piece_w = [&](int g, int k0, char* dst) { glds16(sa.tw + (g * ldw64 + k0) + lane_w, dst); };
    int tail_ops = 0;   // vector-memory operations the previous epilogue issued after the next tile's stage-0 pieces (0 = unknown: drain)
    // q-th prologue DMA instruction of a tile, in issue order: A(0) x4, W(0) x4, A(1) x4, W(1) x4
    constexpr int NPRO = 16;
    auto issue_q = [&](int q) {
        const int st = q >> 3, g = q & 7;
        if (st >= nk) return;
        if (g < 4) piece_a(g, kofs(st), (st ? sA1 : sA0) + (g * 512 + wave * 64) * 16);
        else if (st < 2) piece_w(g - 4, kofs(st), w_dst(st ? sW1 : sW0) + ((g - 4) * 512 + wave * 64) * 16);
    };
    if (have) {
        rot = p.rot ? tm % nk : 0;
        set_tile(A + (BATCH ? (int64_t)tz * p.strideA : 0) + a_row0(tm) * p.lda,
                 W + (BATCH ? (int64_t)(tz % p.w_mod) * p.strideW : 0) + (int64_t)tile_n0(tn) * p.ldw);
#pragma unroll
        for (int q = 0; q < NPRO; ++q) issue_q(q);
    }
    for (int it = 0; have; ++it) {
        const int64_t m0 = tile_m0(tm), m_lo = (int64_t)tm * 256;   // rows < m_lo belong to the previous tile
        const int n0 = tile_n0(tn), n_lo = tn * 256;
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        bf16x8_t bfr[4], af[8];
        // stage 0 of this tile (issued before the previous tile's epilogue) must have landed; stage 1 may still fly on
        // the first tile only (afterwards the previous epilogue's stores sit behind it in the queue, so drain).
        // Later tiles: stage 0 (DMA pieces 0..7 of the prologue) was interleaved with the first four row blocks of the previous epilogue and
        // vmcnt retires in order, so everything issued AFTER piece 7 may still fly: the stores (and residual loads) of row blocks 4..7 and the
        // 8 pieces of stage 1 -- `tail_ops`, counted by the epilogue below for full tiles (edge tiles predicate their stores: drain).
        {
            // pieces of the prologue issued after W(0): stage 1 (8) -- they may all still fly
            const int after0 = nk > 1 ? 8 : 0;
            const int allow = it == 0 ? after0 : tail_ops;
            if (allow >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (allow >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (allow >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (TRACE) { unsigned long long t = __builtin_readcyclecounter(); t_wait += t - t_begin; t_begin = t; }
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8_t*)(sW0 + b_base + off_h0 + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8_t*)(sA0 + a_base + off_h0 + i * 16 * 128);

        // One half-step: 32 MFMAs on (bfr, af) while the next half's fragments are read ({4 MFMA, 1 ds_read} x 8).  The B
        // fragments of the next half go into bn right after the first MFMA group, so no LDS wait gates the group head.
        // srcA / srcW: slots the next half's fragments are read from; dma_k0 >= 0: refill stage (2-ring: A and W of one stage into dma_slot;
        // RING3: W of that stage into dma_slot and, dma_ka0 >= 0, A of the stage after it into dma_aslot)
        // hook: scalar bookkeeping of the NEXT k-step, run between two MFMA groups (where it is free) instead of between two k-steps (where
        // ~60 scalar instructions per wave -- k offset with rotation / tap pairing, slot indices -- sat in front of the first MFMA).
        auto no_hook = []() {};
        auto half_step = [&](auto dma_tag, const char* srcA, const char* srcW, int off, bool load_next, int dma_k0, char* dma_slot, int dma_ka0, char* dma_aslot,
                             auto&& hook) {
            constexpr bool DMA = decltype(dma_tag)::value;      // compile-time: no piece of a copy without refill sits behind a run-time test
            bf16x8_t bn[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (ABL == 4) {   // perf probe only (results are garbage): same operand traffic, half as many 32x32x16 MFMAs
                    f32x16_t* a32 = (f32x16_t*)&acc[i][0];
                    *a32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[0], af[i], *a32, 0, 0, 0);
                    *a32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[2], af[i], *a32, 0, 0, 0);
                    asm volatile("" :: "v"(bfr[1]), "v"(bfr[3]));
                } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ABL != 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                    else asm volatile("" :: "v"(bfr[j]), "v"(af[i]));
                }
                }
                if (load_next) {
                    __builtin_amdgcn_sched_barrier(0);
                    // ABL 5 / 6 (timing probes, garbage results): drop half / three quarters of the fragment reads = the LDS read volume of a
                    // 4-wave 128x128-per-wave layout and below
                    if (!((ABL == 5 || ABL == 6) && (i & 1))) af[i] = *(const bf16x8_t*)(srcA + a_base + off + i * 16 * 128);
                    if (i < 4 && !(ABL == 6 && (i & 1))) bn[i] = *(const bf16x8_t*)(srcW + b_base + off + i * 16 * 128);
                    if (ABL == 6 && i < 4 && (i & 1)) bn[i] = bn[i - 1];
                    if (DMA && ABL != 1) {   // one LDS-DMA instruction per MFMA group instead of a burst behind the barrier
                        if (!RING3) {
                            const int dk = (ABL == 7) ? 0 : dma_k0;   // ABL 7 (timing probe, garbage results): the in-loop DMA always re-reads k-chunk 0 = cache hits
                            if (dk < 0) {}
                            else if (i < 4) piece_a(i, dk, dma_slot + (i * 512 + wave * 64) * 16);
                            else piece_w(i - 4, dk, dma_slot + 256 * 128 + ((i - 4) * 512 + wave * 64) * 16);
                        } else {
                            // one piece in MFMA groups 0, 2, 4, 6 of each half-step: the 8 pieces of a k-step spread evenly over its 16 groups
                            // (measured alternatives, EXPERIMENTS.md old section 3.1: odd groups +0.17 ms per step, groups 0-3 +-0.05, groups 4-7 -5 ... 0 %)
                            const bool slot_ = (i & 1) == 0; const int g_ = i >> 1;
                            if (slot_) {
                                if (dma_k0 >= 0) piece_w(g_, dma_k0, dma_slot + (g_ * 512 + wave * 64) * 16);
                                else if (dma_ka0 >= 0) piece_a(g_, dma_ka0, dma_aslot + (g_ * 512 + wave * 64) * 16);
                            }
                        }
                    }
                    if (i == 5) hook();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (load_next) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = bn[j];
            }
        };
        auto mid_sync = [&](int kt, bool refilled, int touched = 0) {
            // slot kt is entirely in registers; stage kt+1 must have landed before anyone reads it.  RING3: the newest 4 operations of this
            // wave are the A pieces of stage kt+2 (or of the next tile's stage 0), issued during the half-step that just ended, whenever
            // that stage exists: let them fly.  `touched`: the newest 2 / 4 operations are the residual touches of k-step nk - 2 (nothing waits
            // for their data; the wait in front of the epilogue retires them).
            (void)kt;
            if (RING3 && refilled) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else if (touched == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else if (touched == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        using T_ = std::integral_constant<bool, true>;
        using F_ = std::integral_constant<bool, false>;
        // steady state: stage kt + 2 exists, every refill piece is issued behind one uniform test.  RING3: the loop includes k-step nk - 2, whose
        // "stage nk" does not exist (k2 = -1: nothing is issued; continuing the ring into the next tile's stage 0 there was measured: +0.7 ms per step).
        int kt = 0;
        int kc = 2 + rot;                      // chunk of the walk that stage kt + 2 holds
        kc = kc >= nk ? kc - nk : kc;
        int k2 = nk > 2 ? kmap(kc) : -1;       // its k offset; -1: no such stage
        for (; kt + 2 < nk || (RING3 && kt + 1 < nk); ++kt) {
            int k2n = -1;
            auto next_k = [&]() {              // stage kt + 3, for the next pass of this loop
                int c = kc + 1;
                kc = c >= nk ? c - nk : c;
                k2n = kt + 3 < nk ? kmap(kc) : -1;
            };
            // Residual variants, k-step nk - 2 (no stage nk to fetch: the A slot sA2 and this half-step's DMA issue slots are idle): touch the lines
            // of this tile's residual -- one 4-byte LDS-DMA load per 128-byte line, 1024 (bf16) / 2048 (f32) lines per tile = 2 / 4 instructions per
            // wave, landing as garbage in sA2, which nobody reads before the next tile's stage 0 overwrites it.  The epilogue's residual loads, ~1.5
            // k-steps later, then come out of L2: in isolation the residual operand cost out-proj +26 % (0.159 -> 0.201 ms) = the HBM time of its
            // 196 MB, paid as exposed latency while the matrix pipe idles in the epilogue.
            int touched = 0;
            auto touch_residual = [&]() {
                const int esz = p.out_f32 ? 4 : 2;
                const int lines_per_row = 256 * esz / 128;                       // 4 (bf16) or 8 (f32)
                const char* rbase = (const char*)p.residual + (m0 * p.ldr + n0) * (int64_t)esz;
                touched = p.out_f32 ? 4 : 2;
                for (int j = 0; j < touched; ++j) {
                    const int idx = (wave * touched + j) * 64 + lane;            // line of the tile: row idx / lines_per_row, line idx % lines_per_row
                    const int row = idx / lines_per_row, ln = idx - row * lines_per_row;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rbase + ((int64_t)row * p.ldr * esz + ln * 128)),
                                                     (__attribute__((address_space(3))) void*)(sA2 + (wave * 4 + j) * 256), 4, 0, 0);
                }
            };
            const bool touch_here = RES && RING3 && SC_GEMM_RES_TOUCH && !BATCH && kt == nk - 2 && k2 < 0 && nk >= 3;
            if (touch_here && SC_GEMM_RES_TOUCH == 1) touch_residual();
            else touched = 0;
            // MFMAs of (kt, h0); read (kt, h1); RING3: A pieces of stage kt+2 into the A slot stage kt-1 left free
            if (!RING3) half_step(F_{}, sA0, sW0, off_h1, true, -1, nullptr, -1, nullptr, no_hook);
            else half_step(T_{}, sA0, sW0, off_h1, true, -1, nullptr, k2, sA2, no_hook);
            mid_sync(kt, k2 >= 0, (touch_here && SC_GEMM_RES_TOUCH == 1) ? touched : 0);
            if (touch_here && SC_GEMM_RES_TOUCH == 2) touch_residual();      // A/B: one half-step later (retired by the wait in front of the epilogue)
            // MFMAs of (kt, h1); read (kt+1, h0); refill: 2-ring = stage kt+2 (A and W) into slot kt; RING3 = W(kt+2) into W slot kt
            half_step(T_{}, sA1, sW1, off_h0, true, k2, RING3 ? sW0 : sA0, -1, nullptr, next_k);
            rotate_ring();
            k2 = k2n;
        }
        if (!RING3 && kt + 1 < nk) {      // 2-ring, k-step nk - 2: nothing left to refill
            half_step(F_{}, sA0, sW0, off_h1, true, -1, nullptr, -1, nullptr, no_hook);
            mid_sync(kt, false);
            half_step(F_{}, sA1, sW1, off_h0, true, -1, nullptr, -1, nullptr, no_hook);
            rotate_ring();
        }
        // bias for this lane's 4 x 4 output columns: loaded BEFORE the next tile's LDS-DMA is issued (an ordinary load
        // issued behind the DMA would have to drain it first: vmcnt is in-order)
        f32x4_t bias4[4];
        auto load_bias = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nn = n0 + wn * 64 + j * 16 + fk * 4;
                bias4[j] = p.bias ? *(const f32x4_t*)(p.bias + nn) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
        };
        {   // last k-step (peeled: nothing left to prefetch after its first half)
            half_step(F_{}, sA0, sW0, off_h1, true, -1, nullptr, -1, nullptr, no_hook);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            half_step(F_{}, nullptr, nullptr, 0, false, -1, nullptr, -1, nullptr, no_hook);
            rotate_ring();      // sA0 / sW0: where stage nk would go = the next tile's stage 0
        }
        load_bias();    // (measured, round 3: loading them before the last half-step instead -- latency under 32 MFMAs -- costs +1.5 ms per step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---- next tile's first two stages fly during this tile's epilogue (the slots are free after this barrier:
        //      the epilogue uses its own 18 KiB image region)
        if (TRACE) { unsigned long long t = __builtin_readcyclecounter(); t_loop += t - t_begin; t_begin = t; }
        int ntm, ntn, ntz;
        const bool nhave = tile_of(it + 1, ntm, ntn, ntz);
        const int64_t c_off = BATCH ? (int64_t)tz * p.strideC : 0;      // this tile's product (its epilogue runs below, after the next tile is set up)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nhave) set_tile(A + (BATCH ? (int64_t)ntz * p.strideA : 0) + a_row0(ntm) * p.lda,
                            W + (BATCH ? (int64_t)(ntz % p.w_mod) * p.strideW : 0) + (int64_t)tile_n0(ntn) * p.ldw);
        rot = (nhave && p.rot) ? ntm % nk : 0;   // the k-loop of THIS tile is over; from here on kofs() addresses the next tile
        const int emode = (RES ? p.epi_mode_res : p.epi_mode) & 0xff;
        if (nhave && emode == 0) {
#pragma unroll
            for (int q = 0; q < NPRO; ++q) issue_q(q);
        }

        if (TRACE) { unsigned long long t = __builtin_readcyclecounter(); t_pre += t - t_begin; t_begin = t; }
        // ------------------------------------------------------------------------------------------ epilogue
        // FAST PATH (round 3): interior tile + bf16 vector stores + a next tile with at least two k-steps.  The general path below predicates
        // every store on (m >= m_lo && n >= n_lo) and recomputes kofs() -- with its rot / kpair branches -- for each of the 16 prologue pieces:
        // ~40 basic blocks per epilogue, which pins every ds_bpermute result wait directly in front of its store (no scheduling across a
        // block boundary) and costs ~250 scalar predicate / address instructions per wave.  Here nothing is predicated, the two k offsets of
        // the next tile's prologue are computed once, stores and residual loads run off one per-lane base pointer + uniform row steps, and
        // row block i + 1's convert / permlane / crossbar shuffles are issued BEFORE row block i's stores, so the shuffle latency hides
        // under the stores and the next tile's LDS-DMA pieces.  sched_barrier between row blocks keeps the issue order of the vector-memory
        // operations what `tail_ops` below counts on.
        const bool fast_epi = SC_GEMM_FAST_EPI && (!RES || SC_GEMM_FAST_EPI >= 2) && vec_ok && nhave && nk >= 2 && emode == 2 && m0 == m_lo && n0 == n_lo;
        if (fast_epi) {
            bf16_t* Cb = (bf16_t*)p.C + c_off;
            const int srow = lane >> 2, schunk = lane & 3;
            const int src_fk = ((schunk & 1) << 1) | (schunk >> 1);
            const int bperm = (src_fk * 16 + srow) << 2;
            const int ncol0 = n0 + wn * 64 + schunk * 8;
            const int64_t mrow0 = m0 + wm * 128 + srow;
            bf16_t* cptr = Cb + mrow0 * p.ldc + ncol0;
            const bf16_t* rptr = RES ? (const bf16_t*)p.residual + mrow0 * p.ldr + ncol0 : nullptr;
            const int64_t cstep = 16 * p.ldc, rstep = 16 * p.ldr;
            const int kq0 = kofs(0), kq1 = kofs(1);                      // `rot` already addresses the next tile
            auto issue_fast = [&](int q) {                                // q-th prologue DMA instruction: A(0) x4, W(0) x4, A(1) x4, W(1) x4
                const int st = q >> 3, g = q & 7;
                const int k0 = st ? kq1 : kq0;
                if (g < 4) piece_a(g, k0, (st ? sA1 : sA0) + (g * 512 + wave * 64) * 16);
                else piece_w(g - 4, k0, w_dst(st ? sW1 : sW0) + ((g - 4) * 512 + wave * 64) * 16);
            };
            uint4 res[RES ? 4 : 1][2];
            auto load_res_fast = [&](int i) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) res[i & 3][jp] = *(const uint4*)(rptr + i * rstep + jp * 32);
            };
            if (RES) { load_res_fast(0); load_res_fast(1); load_res_fast(2); load_res_fast(3); }
            auto shuffled = [&](int i, uint4 (&o)[2]) {                  // row block i: bias / activation -> bf16 -> 16 rows x 64 contiguous bytes per store
                uint2 pk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4_t v4 = acc[i][j] + bias4[j];
                    if (ACT == SC_ACT_GELU) {
                        const f32x2_t g0 = gelu_poly2((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2((f32x2_t){v4[2], v4[3]});
                        v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                    } else if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                    }
                    pk[j].x = pack2bf(v4[0], v4[1]);
                    pk[j].y = pack2bf(v4[2], v4[3]);
                }
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].x, pk[2 * jp + 1].x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].y, pk[2 * jp + 1].y, false, false);
                    if (SC_PROBES && (p.eprobe & 8)) o[jp] = make_uint4(pk[2 * jp].x, pk[2 * jp].y, pk[2 * jp + 1].x, pk[2 * jp + 1].y);
                    else
                    o[jp] = make_uint4(__builtin_amdgcn_ds_bpermute(bperm, r0[0]), __builtin_amdgcn_ds_bpermute(bperm, r1[0]),
                                       __builtin_amdgcn_ds_bpermute(bperm, r0[1]), __builtin_amdgcn_ds_bpermute(bperm, r1[1]));
                }
            };
            uint4 oc[2][2];
            shuffled(0, oc[0]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + 1 < 8) shuffled(i + 1, oc[(i + 1) & 1]);          // next block's shuffles fly under this block's stores
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    uint4 o = oc[i & 1][jp];
                    if (RES) {
                        const uint4 rv = res[i & 3][jp];
                        o.x = pack2bf(lo2f(o.x) + lo2f(rv.x), hi2f(o.x) + hi2f(rv.x));
                        o.y = pack2bf(lo2f(o.y) + lo2f(rv.y), hi2f(o.y) + hi2f(rv.y));
                        o.z = pack2bf(lo2f(o.z) + lo2f(rv.z), hi2f(o.z) + hi2f(rv.z));
                        o.w = pack2bf(lo2f(o.w) + lo2f(rv.w), hi2f(o.w) + hi2f(rv.w));
                    }
                    if (SC_PROBES && (p.eprobe & 1)) asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
                    else if (SC_PROBES && (p.eprobe & 4)) *(uint4*)(Cb + ((cptr - Cb + i * cstep + jp * 32) & ((1 << 20) - 8))) = o;
                    else if (SC_PROBES && (p.eprobe & 64)) {
                        typedef unsigned u32x4_nt_t __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store((u32x4_nt_t){o.x, o.y, o.z, o.w}, (u32x4_nt_t*)(cptr + i * cstep + jp * 32));
                    }
                    else if (SC_PROBES && (p.eprobe >> 8)) {      // cache-policy bits of the output stores: (eprobe >> 8) = sc0 | sc1 << 1 | nt << 2
                        typedef unsigned u32x4_st_t __attribute__((ext_vector_type(4)));
                        const u32x4_st_t ov = {o.x, o.y, o.z, o.w};
                        bf16_t* ap = cptr + i * cstep + jp * 32;
                        switch (p.eprobe >> 8) {
                            case 1: asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(ap), "v"(ov) : "memory"); break;
                            case 2: asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(ap), "v"(ov) : "memory"); break;
                            case 3: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(ap), "v"(ov) : "memory"); break;
                            case 4: asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(ap), "v"(ov) : "memory"); break;
                            case 5: asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(ap), "v"(ov) : "memory"); break;
                            case 6: asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(ap), "v"(ov) : "memory"); break;
                            default: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(ap), "v"(ov) : "memory"); break;
                        }
                    }
                    else *(uint4*)(cptr + i * cstep + jp * 32) = o;
                }
                if (RES && i + 4 < 8) load_res_fast(i + 4);               // slot i & 3 was consumed just above
                if (SC_PROBES && (p.eprobe & 2)) {}
                else { issue_fast(2 * i); issue_fast(2 * i + 1); }   // (all 8 stage-0 pieces before the first store instead: +-0.02 ms per step, round 3)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (vec_ok) {
            // Lane (frow, fk) holds row frow, columns 16j + 4fk .. +3 of each 16-column block j.  v_permlane16_swap between the
            // 16-lane rows fk and fk^1 regroups a block PAIR (j0, j1) so that even-fk lanes own 8 consecutive columns of j0 and
            // odd-fk lanes 8 consecutive columns of j1 (16 bytes per lane).  Stored from there, the four lanes of every lane QUAD
            // would hit four different rows, and the store path handles a wave quad by quad: 13 B/clk/CU measured, against
            // 24-36 B/clk/CU when a quad covers 64 contiguous bytes (tools/probes/store_probe.hip).  So the 16 x 4 (row x chunk)
            // lane matrix is transposed through the LDS crossbar (ds_bpermute, no LDS memory): lane L ends up with row L >> 2,
            // 16-byte chunk L & 3 of the 32-column half, and a store instruction writes 16 rows x 64 contiguous bytes.
            bf16_t* Cb = (bf16_t*)p.C + c_off;
            const int srow = lane >> 2, schunk = lane & 3;
            const int src_fk = ((schunk & 1) << 1) | (schunk >> 1);                          // lane row that holds this chunk after the swap
            const int bperm = (src_fk * 16 + srow) << 2;
            const int ncol0 = n0 + wn * 64 + schunk * 8;                                     // half 0; half 1 adds 32 columns
            const int64_t mrow0 = m0 + wm * 128 + srow;
            auto load_res = [&](int i, uint4 (&dst)[2]) {
                const int64_t m = mrow0 + i * 16;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int n = ncol0 + jp * 32;
                    dst[jp] = (m >= m_lo && n >= n_lo) ? *(const uint4*)((const bf16_t*)p.residual + m * p.ldr + n) : make_uint4(0, 0, 0, 0);
                }
            };
            // Residual rows: a ring of row blocks loaded ahead of their use.  Plain residual variant with the interleaved prologue (emode 2): FOUR
            // blocks ahead, the first four issued before the first prologue DMA piece -- vmcnt retires in order, so a residual load issued BEHIND
            // DMA pieces can only be waited for together with them; four row blocks (8 pieces) later those pieces have long landed.  With two ahead
            // the wait sat right behind fresh pieces, which is why this variant used to issue the whole prologue after the epilogue (emode 3),
            // exposing its latency at the next tile's start.  (All eight up front does not fit: 128 accumulators + 64 residual registers spill.)
            constexpr int NRES = RES ? 4 : 3;
            const int rahead = (NRES == 4 && emode == 2) ? 4 : 2;
            uint4 res[NRES][2];
            if (RES) {
                load_res(0, res[0]); load_res(1, res[1]);
                if (NRES == 4 && rahead == 4) { load_res(2, res[2]); load_res(3, res[3]); }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t m = mrow0 + i * 16;
                if (RES && rahead == 2 && i + 2 < 8) load_res(i + 2, res[(i + 2) % NRES]);
                uint2 pk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4_t v4 = acc[i][j];
                    v4 += bias4[j];
                    if (ACT == SC_ACT_GELU) {
                        const f32x2_t g0 = gelu_poly2((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2((f32x2_t){v4[2], v4[3]});
                        v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                    } else if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                    }
                    pk[j].x = pack2bf(v4[0], v4[1]);
                    pk[j].y = pack2bf(v4[2], v4[3]);
                }
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].x, pk[2 * jp + 1].x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].y, pk[2 * jp + 1].y, false, false);
                    uint4 o = make_uint4(__builtin_amdgcn_ds_bpermute(bperm, r0[0]), __builtin_amdgcn_ds_bpermute(bperm, r1[0]),
                                         __builtin_amdgcn_ds_bpermute(bperm, r0[1]), __builtin_amdgcn_ds_bpermute(bperm, r1[1]));
                    const int n = ncol0 + jp * 32;
                    if (m >= m_lo && n >= n_lo) {
                        if (RES) {
                            const uint4 rv = res[i % NRES][jp];
                            o.x = pack2bf(lo2f(o.x) + lo2f(rv.x), hi2f(o.x) + hi2f(rv.x));
                            o.y = pack2bf(lo2f(o.y) + lo2f(rv.y), hi2f(o.y) + hi2f(rv.y));
                            o.z = pack2bf(lo2f(o.z) + lo2f(rv.z), hi2f(o.z) + hi2f(rv.z));
                            o.w = pack2bf(lo2f(o.w) + lo2f(rv.w), hi2f(o.w) + hi2f(rv.w));
                        }
                        if (ABL != 3) *(uint4*)(Cb + m * p.ldc + n) = o;
                        else asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
                    }
                }
                if (RES && NRES == 4 && rahead == 4 && i + 4 < 8) load_res(i + 4, res[i % NRES]);      // slot i was consumed just above
                if (nhave && emode == 2) { issue_q(2 * i); issue_q(2 * i + 1); }
            }
        } else if (f32_ok) {
            // fp32 outputs (the ViT / pre-LN residual streams): same quad rule as above.  Lane (frow, fk) holds 4 consecutive fp32 =
            // 16 bytes of row frow; the 4 lanes (frow, 0..3) together own one 64-byte segment, so lane L fetches (crossbar) the
            // value of lane (L >> 2) + 16 (L & 3): a store instruction then writes 16 rows x 64 contiguous bytes per 16-column block.
            float* Cf = (float*)p.C + c_off;
            const int srow = lane >> 2, schunk = lane & 3;
            const int bperm = (srow + 16 * schunk) << 2;
            const int64_t mrow0 = m0 + wm * 128 + srow;
            const int ncol0 = n0 + wn * 64 + schunk * 4;
            const bool full_tile = (m0 == m_lo) && (n0 == n_lo);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t m = mrow0 + i * 16;
                f32x4_t rs[4];
                if (RES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = ncol0 + j * 16;
                        rs[j] = (full_tile || (m >= m_lo && n >= n_lo)) ? *(const f32x4_t*)((const float*)p.residual + m * p.ldr + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    }
                }
                f32x4_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4_t v4 = acc[i][j];
                    v4 += bias4[j];
                    if (ACT == SC_ACT_GELU) {       // f32 output: the fp32 polynomial (the packed-half form carries ~11 bits, meant for bf16 results)
                        const f32x2_t g0 = gelu_poly2_f32((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2_f32((f32x2_t){v4[2], v4[3]});
                        v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                    } else if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float t = v4[r]; o[j][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm, __float_as_int(t))); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = ncol0 + j * 16;
                    if (full_tile || (m >= m_lo && n >= n_lo)) {
                        if (RES) o[j] += rs[j];
                        *(f32x4_t*)(Cf + m * p.ldc + n) = o[j];
                    }
                }
            }
        } else {
            char* Cb = (char*)p.C + c_off * (p.out_f32 ? 4 : 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t m = m0 + wm * 128 + i * 16 + frow;
                if (m < m_lo) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + fk * 4;
                    if (n < n_lo) continue;
                    f32x4_t v4 = acc[i][j];
                    v4 += bias4[j];
                    if (ACT == SC_ACT_GELU) {
                        if (p.out_f32) {
                            const f32x2_t g0 = gelu_poly2_f32((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2_f32((f32x2_t){v4[2], v4[3]});
                            v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                        } else {
                            const f32x2_t g0 = gelu_poly2((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2((f32x2_t){v4[2], v4[3]});
                            v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                        }
                    } else if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                    }
                    if (p.out_f32) {
                        if (p.residual) v4 += *(const f32x4_t*)((const float*)p.residual + m * p.ldr + n);
                        *(f32x4_t*)((float*)Cb + m * p.ldc + n) = v4;
                    } else {
                        if (p.residual) {
                            const uint2 rr2 = *(const uint2*)((const bf16_t*)p.residual + m * p.ldr + n);
                            v4[0] += lo2f(rr2.x); v4[1] += hi2f(rr2.x); v4[2] += lo2f(rr2.y); v4[3] += hi2f(rr2.y);
                        }
                        uint2 o;
                        o.x = pack2bf(v4[0], v4[1]);
                        o.y = pack2bf(v4[2], v4[3]);
                        *(uint2*)((bf16_t*)Cb + m * p.ldc + n) = o;
                    }
                }
            }
        }
        if (nhave && (emode == 3 || (emode == 2 && !vec_ok))) {
#pragma unroll
            for (int q = 0; q < NPRO; ++q) issue_q(q);
        }
        {
            // vector-memory operations issued AFTER the next tile's W(0) pieces, all of which may still fly at the next tile-start wait: the
            // rest of its prologue (8 [+4] pieces) and, when the prologue was interleaved with this epilogue's row blocks (W(0) completes with
            // row block 3), the stores [+ residual loads] of row blocks 4..7 -- counted only for full tiles (edge tiles predicate their stores)
            const bool full_tile_now = (m0 == m_lo) && (n0 == n_lo);
            tail_ops = 0;
            if (k_counted_wait && nhave) {
                tail_ops = nk > 1 ? 8 : 0;
                // row blocks 4..7 of an interleaved epilogue: 8 stores (the 4-ahead ring of the residual variant has issued every load by row block 3)
                if (emode == 2 && vec_ok && full_tile_now && nk > 1) tail_ops += 8;
            }
        }
        have = nhave;
        tm = ntm;
        tn = ntn;
        tz = ntz;
        if (TRACE && lane == 0) {     // per block (wave 0's view) in [0, 8), per wave in [8 + 4 wave, +4): wait / loop / next-tile set-up / epilogue
            unsigned long long t = __builtin_readcyclecounter();
            unsigned long long* tr = p.trace + (size_t)blockIdx.x * 40;
            unsigned long long* tw = tr + 8 + wave * 4;
            tw[0] = t_wait; tw[1] = t_loop; tw[2] = t_pre; tw[3] += t - t_begin;
            if (tid == 0) {
                tr[0] = t_wait; tr[1] = t_loop; tr[2] = t_pre; tr[3] += t - t_begin; tr[4] = it + 1;
                tr[5] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
            }
            t_begin = t;
        }
    }
}

template <int ABL, bool TRACE, int ACT, bool RES, bool RING3, bool BATCH = false>
int launch256_one(const GemmParams& p, int grid, hipStream_t s) {
    constexpr int lds = RING3 ? 5 * 256 * 128 : 2 * SLOT_BYTES;   // 160 KiB (A x3 + W x2) or 128 KiB ([A|W] x2)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<ABL, TRACE, ACT, RES, RING3, BATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<ABL, TRACE, ACT, RES, RING3, BATCH>), dim3(grid), dim3(512), lds, s, p);
    SC_CHECK_LAUNCH();
    return 0;
}

template <int ABL, bool TRACE, bool RING3 = false>
int launch256_var(const GemmParams& p, int grid, hipStream_t s) {
    const bool res = p.residual != nullptr;
    switch (p.act) {
        case SC_ACT_GELU: return res ? launch256_one<ABL, TRACE, SC_ACT_GELU, true, RING3>(p, grid, s) : launch256_one<ABL, TRACE, SC_ACT_GELU, false, RING3>(p, grid, s);
        case SC_ACT_QUICKGELU: return res ? launch256_one<ABL, TRACE, SC_ACT_QUICKGELU, true, RING3>(p, grid, s) : launch256_one<ABL, TRACE, SC_ACT_QUICKGELU, false, RING3>(p, grid, s);
        default: return res ? launch256_one<ABL, TRACE, SC_ACT_NONE, true, RING3>(p, grid, s) : launch256_one<ABL, TRACE, SC_ACT_NONE, false, RING3>(p, grid, s);
    }
}

int launch256(const GemmParams& p, hipStream_t s) {
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const int ntiles = p.tiles_m * p.tiles_n;
    int grid = ntiles < n_cu ? ntiles : n_cu;
#if SC_PROBES
    // PROBES build only.  Instrumentation switches, read once: SC_GEMM_GRID caps the number of blocks (per-CU vs chip-wide limits), SC_GEMM_ABL
    // ablates 1 = the LDS-DMA in the main loop, 2 = the MFMAs, 3 = the epilogue stores, 4 = 32x32x16 MFMAs, 5 / 6 = half / a quarter of the
    // fragment reads, 7 = DMA re-reads k-chunk 0, 8 = A rows from an L2-resident window (results are garbage; timing only);
    // SC_GEMM_RING3=0 selects the two-slot [A|W] ring; sc_debug_set_gemm_trace records per-phase s_memtime stamps
    static const int grid_cap = getenv("SC_GEMM_GRID") ? atoi(getenv("SC_GEMM_GRID")) : 0;
    static const char* abl = getenv("SC_GEMM_ABL");
    if (grid_cap > 0 && grid_cap < grid) grid = grid_cap;
    if (p.trace && abl && abl[0] == '3') return launch256_var<3, true>(p, grid, s);
    if (p.trace && abl && abl[0] == '1') return launch256_var<1, true>(p, grid, s);
    if (p.trace && abl && abl[0] == '2') return launch256_var<2, true>(p, grid, s);
    if (p.trace) return launch256_var<0, true>(p, grid, s);
    if (abl && abl[0] == '1') return launch256_var<1, false>(p, grid, s);
    if (abl && abl[0] == '2') return launch256_var<2, false>(p, grid, s);
    if (abl && abl[0] == '4') return launch256_var<4, false>(p, grid, s);
    if (abl && abl[0] == '5') return launch256_var<5, false>(p, grid, s);
    if (abl && abl[0] == '6') return launch256_var<6, false>(p, grid, s);
    if (abl && abl[0] == '7') return launch256_var<7, false>(p, grid, s);
    if (abl && abl[0] == '8') return launch256_var<8, false, true>(p, grid, s);
    static const bool ring3 = !(getenv("SC_GEMM_RING3") && atoi(getenv("SC_GEMM_RING3")) == 0);
    if (!ring3) return launch256_var<0, false>(p, grid, s);
#endif
    // the three-slot A ring with the refill spread over all 16 MFMA groups of a k-step (RING3 in gemm256_kernel)
    return launch256_var<0, false, true>(p, grid, s);
}

template <int BM, int BN, bool F16 = false>
int launch(const GemmParams& p, int batch, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, batch);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, F16>), grid, dim3(256), lds, s, p);
    SC_CHECK_LAUNCH();
    return 0;
}

int gemm_dispatch(GemmParams p, int batch, hipStream_t s) {
    SC_CHECK_ARG(p.K > 0 && p.K % 64 == 0, "sc_gemm: K=%d must be a positive multiple of 64", p.K);
    SC_CHECK_ARG(p.N > 0 && p.N % 4 == 0, "sc_gemm: N=%d must be a positive multiple of 4", p.N);
    SC_CHECK_ARG(p.M > 0 && batch > 0, "sc_gemm: empty problem M=%lld batch=%d", (long long)p.M, batch);
    SC_CHECK_ARG(p.lda % 8 == 0 && p.ldw % 8 == 0 && p.ldc % 4 == 0, "sc_gemm: lda/ldw must be multiples of 8, ldc of 4");
    SC_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && ((uintptr_t)p.C & 15) == 0,
                 "sc_gemm: A/W/C must be 16-byte aligned");
    if (p.f16) {       // half operands: the shapes gemm8p_pers_kernel does not take (few rows, N % 256 != 0) run on the 128-row kernel
        p.tiles_m = (int)((p.M + 127) / 128);
        if (p.N <= 64) { p.tiles_n = (p.N + 63) / 64; return launch<128, 64, true>(p, batch, s); }
        p.tiles_n = (p.N + 127) / 128;
        return launch<128, 128, true>(p, batch, s);
    }
    if (batch == 1 && p.N >= 256 && p.M >= 256 && p.K % 64 == 0) {
        const int64_t t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
        static const int min_tiles = SC_TUNE_INT("SC_GEMM_MIN_TILES", 100);
        static const bool force_v1 = SC_TUNE_SET("SC_GEMM_V1");
        if (t256 >= min_tiles && !force_v1) {
            p.tiles_m = (int)((p.M + 255) / 256); p.tiles_n = (p.N + 255) / 256;
            if (p.band < 0) p.band = p.tiles_n >= 16 ? 4 : 0;   // measured: 8192^3 +20 %; N <= 3072 (the step's shapes) neutral
            return launch256(p, s);
        }
    }
    if (batch > 1 && p.inner == 0 && !p.bias && !p.residual && p.act == SC_ACT_NONE && p.N >= 256 && p.M >= 256) {
        // independent same-shape products (split-K partials of a weight gradient) on the 256-tile kernel: one persistent tile list over all of them
        static const bool no_b256 = SC_TUNE_SET("SC_GEMM_NOBATCH256");
        const int64_t t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256) * (int64_t)batch;
        if (!no_b256 && t256 >= 100 && t256 < 0x7fffffff) {
            p.tiles_m = (int)((p.M + 255) / 256); p.tiles_n = (p.N + 255) / 256;
            p.nbatch = batch; p.band = 0; p.rot = 1; p.epi_mode = 2; p.epi_mode_res = 3; p.kpair = 0;
            static int n_cu = 0;
            if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
            const int grid = (int)(t256 < n_cu ? t256 : n_cu);
            return launch256_one<0, false, SC_ACT_NONE, false, true, true>(p, grid, s);
        }
    }
    if (p.N <= 64) {
        p.tiles_m = (int)((p.M + 127) / 128); p.tiles_n = (p.N + 63) / 64;
        return launch<128, 64>(p, batch, s);
    }
    p.tiles_m = (int)((p.M + 127) / 128); p.tiles_n = (p.N + 127) / 128;
    return launch<128, 128>(p, batch, s);
}

}  // namespace

int sc_vendor_gemm_try(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, const void* residual,
                       int64_t ldr, int64_t M, int N, int K, int out_f32, hipStream_t s);   // vendor_gemm.hip

static int g_last_path = 0;   // 0: gemm256_kernel / gemm_bf16_kernel, 1: vendor library, 3: gemm8p_pers_kernel (instrumentation: which kernel a launch hit)
extern "C" int sc_gemm_last_path(void) { return g_last_path; }
// Kernel choice behind sc_gemm_bf16 (developer / test switch, exported but not part of include/speechclip_hip.h):
//   -1 the dispatcher's rule (default)   0 gemm256_kernel / gemm_bf16_kernel only   16 gemm8p whenever the shape allows   26 gemm8p with the static tile order
// The round-5 A/B variants (17 per-tile launch, 19 / 24 / 25 K rotations, 20 tap-paired K walk, 21-23 forced column bands) exist only in a -DSC_LAB=1 build
// (tools/build_ab.sh); the product library maps them to the default.
#ifndef SC_LAB
#define SC_LAB 0
#endif
#ifndef SC_GEMM_MODE_DEFAULT
#define SC_GEMM_MODE_DEFAULT -1
#endif
static int g_gemm_mode = SC_GEMM_MODE_DEFAULT;
extern "C" void sc_debug_set_gemm_mode(int mode) { g_gemm_mode = (SC_LAB || mode == -1 || mode == 0 || mode == 16 || mode == 26) ? mode : -1; }
static unsigned long long* g_gemm_trace = nullptr;
// per-phase s_memtime stamps of the 256-tile kernel: effective only in the PROBES build (the product library instantiates no TRACE variant)
extern "C" void sc_debug_set_gemm_trace(void* dev_buf) { g_gemm_trace = SC_PROBES ? (unsigned long long*)dev_buf : nullptr; }

extern "C" int sc_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                            const float* bias, const void* residual, int64_t ldr, int64_t M, int N, int K,
                            int flags, void* stream) {
    const int f16 = (flags & SC_GEMM_F16) ? 1 : 0;
    if (!f16 && (flags & SC_GEMM_ACT_MASK) == 0 && lda >= K && !g_gemm_trace && A && W && C && M > 0 && N > 0 && K > 0 && K % 64 == 0) {
        // plain GEMM (+ bias, + residual; bf16 or fp32 out): the vendor library's kernel when a workspace is registered (vendor_gemm.hip)
        const int rc = sc_vendor_gemm_try(A, lda, W, ldw, C, ldc, bias, residual, ldr, M, N, K, (flags & SC_GEMM_OUT_F32) ? 1 : 0, (hipStream_t)stream);
        if (rc <= 0) { g_last_path = 1; return rc; }
    }
    // the argument rules of gemm_dispatch, checked once for every hand-written kernel
    SC_CHECK_ARG(K > 0 && K % 64 == 0, "sc_gemm: K=%d must be a positive multiple of 64", K);
    SC_CHECK_ARG(N > 0 && N % 4 == 0, "sc_gemm: N=%d must be a positive multiple of 4", N);
    SC_CHECK_ARG(M > 0, "sc_gemm: empty problem M=%lld batch=1", (long long)M);
    SC_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "sc_gemm: lda/ldw must be multiples of 8, ldc of 4");
    SC_CHECK_ARG(A && W && C, "sc_gemm: null operand");
    SC_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0, "sc_gemm: A/W/C must be 16-byte aligned");
    if (g_gemm_mode != 0 && (!g_gemm_trace || g_gemm_mode > 0)) {
        // bf16-output GEMMs with N % 256 == 0: the ping-pong kernel (gemm8p.hip), persistent form, from 128 tiles up (ViT-B/32 at 256 images: 150 tiles,
        // +5 ... +19 % over gemm256_kernel; below that the 128 x 128 kernel's finer grid wins); bf16 or fp32 output
        Gemm8pParams d{};
        d.A = (const bf16_t*)A; d.lda = lda; d.W = (const bf16_t*)W; d.ldw = ldw; d.C = C; d.ldc = ldc; d.out_f32 = (flags & SC_GEMM_OUT_F32) ? 1 : 0;
        d.bias = bias; d.residual = residual; d.ldr = ldr; d.M = M; d.N = N; d.K = K;
        d.act = flags & SC_GEMM_ACT_MASK; d.f16 = f16;
        d.esteps = 4; d.trace = g_gemm_trace;
        // column-band tile order: wide outputs (>= 16 N tiles: 8192^3 1 444 vs 1 323 TF/s unbanded, gemm256_kernel 1 406; HuBERT-large fc1 +2 %) walk 4 N tiles at a time
        d.band = (g_gemm_mode == -1 && N / 256 >= 16) ? 4 : 0;
        d.sched = g_gemm_mode == 26 ? -1 : 0;      // 26: static tile order (A/B, and the reference of the dynamic order's bit-identity test)
#if SC_LAB
        d.kpair = (g_gemm_mode == 20 && lda < K && K == 3 * (lda / 2) && (lda / 2) % 64 == 0) ? (int)(lda / 2 / 64) : 0;      // tap-paired K walk
        if (g_gemm_mode == 17) d.esteps = 1;                                                                                     // per-tile kernel
        d.rows = g_gemm_mode == 19 ? 1 : g_gemm_mode == 24 ? 2 : g_gemm_mode == 25 ? 3 : 0;                                      // K rotations
        if (g_gemm_mode >= 21 && g_gemm_mode <= 23) d.band = g_gemm_mode == 21 ? 3 : g_gemm_mode == 22 ? 4 : 6;                  // forced column bands
#endif
        // (the kernel copies the bias vector with 16-byte loads and reads the residual in 16-byte pieces)
        const bool aligned = (!residual || ((uintptr_t)residual & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0);
        const bool dflt_ok = N % 256 == 0 && N <= 8192 && ((M + 255) / 256) * (int64_t)(N / 256) >= 128;
        if (aligned && (g_gemm_mode > 0 || dflt_ok)) {
            const int rc = sc_gemm8p_try(d, (hipStream_t)stream);
            if (rc <= 0) { g_last_path = 3; return rc; }
        }
    }
    g_last_path = 0;
    GemmParams p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.strideA = 0;
    p.W = (const bf16_t*)W; p.ldw = ldw; p.strideW = 0; p.w_mod = 1;
    p.C = C; p.ldc = ldc; p.strideC = 0;
    p.bias = bias; p.residual = residual; p.ldr = ldr;
    p.M = M; p.N = N; p.K = K;
    p.act = flags & SC_GEMM_ACT_MASK; p.out_f32 = (flags & SC_GEMM_OUT_F32) ? 1 : 0; p.f16 = f16;
    p.trace = g_gemm_trace;
    // the measured best (A/B knobs of the PROBES build: SC_GEMM_NOROT, SC_GEMM_BAND, SC_GEMM_EPI, SC_GEMM_EPI_RES, SC_GEMM_NOKPAIR)
    static const int k_rot = SC_TUNE_SET("SC_GEMM_NOROT") ? 0 : 1;
    static const int k_band = SC_TUNE_INT("SC_GEMM_BAND", -1);   // -1: chosen by the dispatcher
    static const int k_epi = SC_TUNE_INT("SC_GEMM_EPI", 2);
    static const int k_epi_res = SC_TUNE_INT("SC_GEMM_EPI_RES", 2);   // 2 since late round 2 (four-ahead residual ring): -0.13 ms per step vs 3 in 7 of 7 A/B passes
    p.rot = k_rot; p.band = k_band; p.epi_mode = k_epi; p.epi_mode_res = k_epi_res;
    static const int k_stagger = SC_TUNE_INT("SC_GEMM_STAGGER", 0);
    static const int k_eprobe = SC_TUNE_INT("SC_GEMM_EPROBE", 0);
    static const int k_nt_n = SC_TUNE_INT("SC_GEMM_NT_N", 0);       // PROBES: streaming (nt) stores for the GEMMs with this N and no activation (by consumer)
    static const int k_nt_act = SC_TUNE_INT("SC_GEMM_NT_ACT", 0);
    static const int k_st_policy = SC_TUNE_INT("SC_GEMM_ST_POLICY", 0);   // PROBES: sc0 | sc1 << 1 | nt << 2 on the fast epilogue's output stores
    p.stagger = k_stagger; p.eprobe = k_eprobe | (k_st_policy << 8) | ((k_nt_n && N == k_nt_n && (int)(flags & SC_GEMM_ACT_MASK) == k_nt_act && !residual) ? 64 : 0);
    static const int k_pair = SC_TUNE_SET("SC_GEMM_NOKPAIR") ? 0 : 1;
    if (k_pair && lda < K && K == 3 * (lda / 2) && (lda / 2) % 64 == 0) p.kpair = (int)(lda / 2 / 64);   // k = 3, stride-2 conv layers of the extractor
    return gemm_dispatch(p, 1, (hipStream_t)stream);
}

// Two-level batch: product z = zo * inner + zi (zo < outer) reads A at zo*strideA + zi*strideA2, W at zo*strideW + zi*strideW2 and writes C at
// zo*strideC + zi*strideC2 (elements) -- e.g. (utterance, head) pairs of packed q|k|v rows: outer stride = rows per utterance, inner stride = 64.
extern "C" int sc_gemm_bf16_batched2(const void* A, int64_t lda, int64_t strideA, int64_t strideA2, const void* W, int64_t ldw, int64_t strideW,
                                     int64_t strideW2, void* C, int64_t ldc, int64_t strideC, int64_t strideC2, int64_t M, int N, int K, int outer,
                                     int inner, int flags, void* stream) {
    GemmParams p{};
    SC_CHECK_ARG(outer > 0 && inner > 0 && (int64_t)outer * inner <= 65535, "sc_gemm_bf16_batched2: outer*inner=%lld must be in [1, 65535]",
                 (long long)outer * inner);
    p.A = (const bf16_t*)A; p.lda = lda; p.strideA = strideA; p.strideA2 = strideA2;
    p.W = (const bf16_t*)W; p.ldw = ldw; p.strideW = strideW; p.strideW2 = strideW2; p.w_mod = 1;
    p.C = C; p.ldc = ldc; p.strideC = strideC; p.strideC2 = strideC2; p.inner = inner;
    p.bias = nullptr; p.residual = nullptr; p.ldr = 0;
    p.M = M; p.N = N; p.K = K;
    p.act = flags & SC_GEMM_ACT_MASK; p.out_f32 = (flags & SC_GEMM_OUT_F32) ? 1 : 0;
    SC_CHECK_ARG((strideA % 8 == 0) && (strideA2 % 8 == 0) && (strideW % 8 == 0) && (strideW2 % 8 == 0) && (strideC % 4 == 0) && (strideC2 % 4 == 0),
                 "sc_gemm_bf16_batched2: operand strides must keep 16-byte alignment");
    return gemm_dispatch(p, outer * inner, (hipStream_t)stream);
}

extern "C" int sc_gemm_bf16_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw,
                                    int64_t strideW, int w_mod, void* C, int64_t ldc, int64_t strideC,
                                    const float* bias, int64_t M, int N, int K, int batch, int flags, void* stream) {
    GemmParams p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.strideA = strideA;
    p.W = (const bf16_t*)W; p.ldw = ldw; p.strideW = strideW; p.w_mod = w_mod > 0 ? w_mod : 1;
    p.C = C; p.ldc = ldc; p.strideC = strideC;
    p.bias = bias; p.residual = nullptr; p.ldr = 0;
    p.M = M; p.N = N; p.K = K;
    p.act = flags & SC_GEMM_ACT_MASK; p.out_f32 = (flags & SC_GEMM_OUT_F32) ? 1 : 0;
    SC_CHECK_ARG(batch <= 65535, "sc_gemm_batched: batch=%d exceeds grid.y limit", batch);
    return gemm_dispatch(p, batch, (hipStream_t)stream);
}
