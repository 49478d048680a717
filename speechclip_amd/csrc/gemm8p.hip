// bf16 MFMA GEMM, ping-pong ("8-phase") form for gfx950 (CDNA4):  C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual, bf16 out.
//
// 256 x 256 x 64 tiles, 8 waves = 2 groups (waves 0-3 / 4-7: one wave of each group on every SIMD) x 4 column strips; wave tile 128 x 64.
// Each 64-deep k-step is cut into 4 PHASES per wave: one 64 x 32 quadrant of the wave tile = 16 MFMAs, preceded by the LDS fragment reads it needs
// and by the LDS-DMA issue of one 16 KiB half tile of a later k-step.  A phase is two barrier intervals -- memory (reads + DMA issue), then matrix
// (16 back-to-back MFMAs at s_setprio 1) -- and the two groups run ONE INTERVAL APART: while a SIMD's group-0 wave is in its MFMA cluster the
// group-1 wave issues its reads and DMA, and vice versa.  The matrix pipe always has exactly one wave streaming MFMAs and never waits behind a
// vector-memory / LDS issue stall of the same wave (gemm256_kernel interleaves 4 MFMA : 1 read : 1 DMA in every wave, all 8 waves in lock step,
// and measures as if MFMA, DMA issue and LDS reads did not overlap at all: EXPERIMENTS.md R5-1).
//
// LDS: 2 buffers x {A0, A1, B0, B1} half tiles of 128 rows x 128 B (A_g: the 128 rows of group g; B_h: W rows h*128 ..), swizzled on the source
// side like gemm.hip (chunk position c of row r holds k-chunk c ^ (r & 7)).  Refill: phase (t,0) A0(t+1), (t,1) A1(t+1), (t,2) B0(t+2), (t,3) B1(t+2):
// every slot is re-staged at least one full phase after the last wave finished reading it (reads are waited for BEFORE the barrier that ends a
// memory interval), and at least 3.5 phases before its first reader.  One counted wait per k-step (phase 3: vmcnt(4)).
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "common.h"
#include "gemm8p.h"
#include "../../include/speechclip_hip.h"

#ifndef SC_PROBES
#define SC_PROBES 0
#endif
#ifndef SC_8P_RES_LATE        // 1: residual variants issue the next tile's k-step-1 refill from the epilogue, behind the first residual loads (A/B: in-step -2 %)
#define SC_8P_RES_LATE 0
#endif
#ifndef SC_8P_ABL             // timing ablations of the persistent kernel (garbage results): 1 no fragment reads, 2 no LDS-DMA, 3 no MFMA, 4 no barrier after the MFMA cluster
#define SC_8P_ABL 0
#endif
#ifndef SC_8P_PRIO
#define SC_8P_PRIO 1
#endif
#ifndef SC_8P_RES_ASM         // 1: residual epilogues load the residual tile with explicit loads + counted waits and store after the last add (see the epilogue)
#define SC_8P_RES_ASM 1
#endif
#ifndef SC_8P_EPI_PAIR        // 1: the two groups' epilogues share ONE barrier interval (group 1 drops the barrier behind its last cluster and re-enters every tile one barrier late)
#define SC_8P_EPI_PAIR 1
#endif
#ifndef SC_8P_BIAS_INIT       // 1: a tile's accumulators start from the bias (read from LDS in the first k-step's memory intervals) instead of zero: no bias add in the epilogue
#define SC_8P_BIAS_INIT 1
#endif
#ifndef SC_8P_DYN             // 1: the persistent kernel takes its tiles from per-XCD counters (Gemm8pParams::sched >= 0) instead of a fixed stride
#define SC_8P_DYN 1
#endif

namespace {

constexpr int HT = 128 * 128;            // one half tile: 128 rows x 128 B = 16 KiB
constexpr int BUF = 4 * HT;              // A0 A1 B0 B1

// Tile counters of the persistent kernel's dynamic order: one 64-byte slot per launch in flight (the host hands out slots round robin): eight 64-bit words,
// word x = (launch generation << 32) | next tile of XCD x's chunk (beyond the one tile every block owns by its position).  The generation (launch sequence
// number / ring size + 1, handed in by the host, monotonic) makes the ring FAULT-TOLERANT: every block raises its XCD's word to (generation << 32) with one
// atomic max before its first fetch, so whatever an earlier user of the slot left behind -- including a launch that faulted or was torn down mid-flight --
// is an older generation and is overwritten; a word of the current generation is left alone.  Nothing has to be cleaned up at the end of a launch.
// (tests/test_gemm8p_gpu.py::test_dynamic_tile_order_survives_a_poisoned_counter_ring fills the ring with "every tile already taken" and compares bits.)
constexpr int SCHED_RING = 4096;
constexpr int SCHED_LDS = 2 * BUF + 32752;      // where a block's waves exchange the fetched index (behind the bias vector: dynamic order needs N < 8192)
__device__ unsigned long long g_sched[SCHED_RING * 8];
// Where the residual epilogues send the stores of rows that belong to the previous tile (ragged last M panel, tile shifted back): their stores all follow
// the last residual add, and 8-32 per-lane store predicates held until then cost more registers than the kernel has (spills inside the k-loop);
// an address select per row block costs none.  Contents are never read.
__device__ __attribute__((aligned(16))) char g_sink[64 * 16 + 256];

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int ACT, bool RES>
__global__ __launch_bounds__(512) void gemm8p_kernel(Gemm8pParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, w4 = wave & 3;
    const int nk = p.nk;

    // XCD-aware tile order (bijective): blocks b, b + 8, .. share an XCD and take a contiguous, M-panel-major chunk of the tile space
    const int tiles_m = (int)(p.M / 256), tiles_n = p.tn;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = v / tiles_n, tn = v - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * 256;
    const int n0 = tn * 256;
    const bf16_t* ta = p.A + m0 * p.lda;
    const bf16_t* tw = p.W + (int64_t)n0 * p.ldw;

    const int frow = lane & 15, fk = lane >> 4;
    const int off_h0 = frow * 128 + ((fk ^ (frow & 7)) << 4);
    const int off_h1 = frow * 128 + (((4 + fk) ^ (frow & 7)) << 4);
    // fragment bases inside a buffer: A half of my group; my 64-row strip of W inside its half tile
    const int a_base = g * HT;
    const int b_base = 2 * HT + (w4 >> 1) * HT + (w4 & 1) * 64 * 128;

    // LDS-DMA: a half tile = 2 instructions per wave: rows q * 64 + wave * 8 + (lane >> 3)
    const int lr = lane >> 3, lc = lane & 7;
    const int64_t lane_a = (int64_t)(wave * 8 + lr) * p.lda + ((lc ^ lr) << 3);
    const int64_t lane_w = (int64_t)(wave * 8 + lr) * p.ldw + ((lc ^ lr) << 3);
    const int64_t lda64 = 64 * p.lda, ldw64 = 64 * p.ldw;
    auto stage_a = [&](int h, int kt, char* buf) {     // A half tile h of k-step kt
        const bf16_t* src = ta + (int64_t)h * 128 * p.lda + (int64_t)kt * 64 + lane_a;
        char* dst = buf + h * HT + wave * 1024;
        glds16(src, dst);
        glds16(src + lda64, dst + 8192);
    };
    auto stage_b = [&](int h, int kt, char* buf) {     // W half tile h of k-step kt
        const bf16_t* src = tw + (int64_t)h * 128 * p.ldw + (int64_t)kt * 64 + lane_w;
        char* dst = buf + (2 + h) * HT + wave * 1024;
        glds16(src, dst);
        glds16(src + ldw64, dst + 8192);
    };

    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (SC_PROBES && p.trace) tr0 = __builtin_readcyclecounter();
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t af[4][2], bf_[2][2][2];          // af[i][h]: row block i of the current A sub tile, k half h; bf_[b][j][h]: column block j of B sub tile b

    // prologue: k-step 0 complete, W of k-step 1
    {
        char* b0 = smem; char* b1 = smem + BUF;
        stage_a(0, 0, b0); stage_a(1, 0, b0); stage_b(0, 0, b0); stage_b(1, 0, b0);
        if (nk > 1) { stage_b(0, 1, b1); stage_b(1, 1, b1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (g == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // group 1 runs one barrier interval behind group 0
    if (SC_PROBES && p.trace) tr1 = __builtin_readcyclecounter();

    auto read_a = [&](const char* buf, int a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const bf16x8_t*)(buf + a_base + (a * 64 + i * 16) * 128 + off_h0);
            af[i][1] = *(const bf16x8_t*)(buf + a_base + (a * 64 + i * 16) * 128 + off_h1);
        }
    };
    auto read_b = [&](const char* buf, int b) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf_[b][j][0] = *(const bf16x8_t*)(buf + b_base + (b * 32 + j * 16) * 128 + off_h0);
            bf_[b][j][1] = *(const bf16x8_t*)(buf + b_base + (b * 32 + j * 16) * 128 + off_h1);
        }
    };
    auto quadrant = [&](auto atag, auto btag) {
        constexpr int a = decltype(atag)::value, b = decltype(btag)::value;
        if (SC_8P_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[a * 4 + i][b * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[b][j][h], af[i][h], acc[a * 4 + i][b * 2 + j], 0, 0, 0);
        if (SC_8P_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto mem_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mat_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    char* bx = smem;            // buffer of k-step kt
    char* by = smem + BUF;      // the other one
    for (int kt = 0; kt < nk; ++kt) {
        // ---- phase 0: quadrant (a0, b0); stage A0(kt + 1)
        read_a(bx, 0); read_b(bx, 0);
        if (kt + 1 < nk) stage_a(0, kt + 1, by);
        mem_end();
        quadrant(I0{}, I0{});
        mat_end();
        // ---- phase 1: quadrant (a0, b1); stage A1(kt + 1)
        read_b(bx, 1);
        if (kt + 1 < nk) stage_a(1, kt + 1, by);
        mem_end();
        quadrant(I0{}, I1{});
        mat_end();
        // ---- phase 2: quadrant (a1, b1); stage B0(kt + 2) into the slot B0(kt) left after phase 1
        read_a(bx, 1);
        if (kt + 2 < nk) stage_b(0, kt + 2, bx);
        mem_end();
        quadrant(I1{}, I1{});
        mat_end();
        // ---- phase 3: quadrant (a1, b0); stage B1(kt + 2); everything of k-step kt + 1 must have landed (in-order: all but the newest 4 / 2 / 0)
        if (kt + 2 < nk) { stage_b(1, kt + 2, bx); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mem_end();
        quadrant(I1{}, I0{});
        mat_end();
        { char* x = bx; bx = by; by = x; }
    }
    if (g == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    if (SC_PROBES && p.trace) tr2 = __builtin_readcyclecounter();

    // ---- epilogue (same store shape as gemm.hip's fast path: 16 rows x 64 contiguous bytes per store)
    f32x4_t bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + w4 * 64 + j * 16 + fk * 4;
        bias4[j] = p.bias ? *(const f32x4_t*)(p.bias + nn) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int srow = lane >> 2, schunk = lane & 3;
    const int src_fk = ((schunk & 1) << 1) | (schunk >> 1);
    const int bperm = (src_fk * 16 + srow) << 2;
    const int ncol0 = n0 + w4 * 64 + schunk * 8;
    const int64_t mrow0 = m0 + g * 128 + srow;
    bf16_t* cptr = (bf16_t*)p.C + mrow0 * p.ldc + ncol0;
    const bf16_t* rptr = RES ? (const bf16_t*)p.residual + mrow0 * p.ldr + ncol0 : nullptr;
    const int64_t cstep = 16 * p.ldc, rstep = 16 * p.ldr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 rv[2];
        if (RES) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) rv[jp] = *(const uint4*)(rptr + i * rstep + jp * 32);
        }
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
            uint4 o = make_uint4(__builtin_amdgcn_ds_bpermute(bperm, r0[0]), __builtin_amdgcn_ds_bpermute(bperm, r1[0]),
                                 __builtin_amdgcn_ds_bpermute(bperm, r0[1]), __builtin_amdgcn_ds_bpermute(bperm, r1[1]));
            if (RES) {
                o.x = pack2bf(lo2f(o.x) + lo2f(rv[jp].x), hi2f(o.x) + hi2f(rv[jp].x));
                o.y = pack2bf(lo2f(o.y) + lo2f(rv[jp].y), hi2f(o.y) + hi2f(rv[jp].y));
                o.z = pack2bf(lo2f(o.z) + lo2f(rv[jp].z), hi2f(o.z) + hi2f(rv[jp].z));
                o.w = pack2bf(lo2f(o.w) + lo2f(rv[jp].w), hi2f(o.w) + hi2f(rv[jp].w));
            }
            *(uint4*)(cptr + i * cstep + jp * 32) = o;
        }
    }
    if (SC_PROBES && p.trace && lane == 0) {     // per wave: prologue (launch -> first k-step), k-loop, epilogue issue; absolute start / end stamps
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tr3 = __builtin_readcyclecounter();
        unsigned long long* tr = p.trace + ((size_t)(blockIdx.x % 4096) * 8 + wave) * 8;
        tr[0] = tr1 - tr0; tr[1] = tr2 - tr1; tr[2] = tr3 - tr2; tr[3] = tr0; tr[4] = tr3;
        tr[5] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// Persistent form.  One block per CU walks its tile list; the k-step sequence runs on across tile boundaries: the DMA refill issued from the last
// two k-steps of a tile already belongs to the next tile (its whole first k-step and the W halves of the second), so that tile starts without a
// prologue, and its loads land while the epilogue runs.  The accumulators are re-initialised quadrant by quadrant inside the memory intervals of the
// first k-step (32 moves beside the other group's MFMA cluster instead of 128 in front of the first one).
// (A ROLLING epilogue -- quadrant Q_p of the finished tile converted and stored inside the memory interval that precedes the first MFMA cluster of
// quadrant Q_p of the next tile -- was built and measured in round 5: 3-7 % SLOWER than even the per-tile kernel on the K = 768 shapes.  The CU's
// store path moves 24-36 B/clk; 16 KiB of stores per group and interval stretch each of the 8 intervals of that k-step far beyond the 256 cycles
// of the MFMA cluster they were meant to hide under.  EXPERIMENTS.md, round 5.)
// F16 (round 6): A, W and the 16-bit outputs / residuals are IEEE half instead of bf16 (SC_GEMM_F16; the pre-LN encoder layers of HuBERT-large): the f16 MFMA opcode of the
// same shape, RNE conversion on the way out; everything else -- LDS image, schedule, fp32 accumulators, fp32 outputs -- is format-blind.
template <int ACT, bool RES, bool F32, bool F16 = false>
__global__ __launch_bounds__(512) void gemm8p_pers_kernel(Gemm8pParams p) {
    // (fp32 outputs keep the plain pairing: their epilogues move 512 KiB per tile and are bound by the CU's load / store path -- side by side they
    //  measured 6 % slower (P-large out-proj 740 -> 697 TF/s), one after the other group 0's stores overlap group 1's residual loads)
    // (round 6: starting every second block of an XCD 25 / 50 / 75 % of a tile late in the ~2-round fp32-epilogue launches of P-large, to take the CUs out of lock step --
    //  k-loops with HBM idle, then 512 KiB epilogues with the matrix pipes idle, all at once: out-proj 745 -> 740 / 697 / 662 TF/s, fc2 1 113 -> 1 085 / 1 025 / 992, P-large step
    //  40.3 -> 40.4 / 40.8 / 41.1 ms.  The delay simply adds: the epilogues are not slowed by each other.  profiles/r06_fp32_epilogue_stagger_ab.txt.  Removed.)
    constexpr bool PAIR = SC_8P_EPI_PAIR && !F32;
    static_assert(!(SC_8P_RES_LATE && RES && F32), "SC_8P_RES_LATE (A/B build) re-issues the skipped refill only from the bf16 residual epilogue: not with fp32 outputs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, w4 = wave & 3;
    const int nk = p.nk;

    // persistent XCD-aware tile list (as gemm256_kernel): block b lives on XCD b % 8 and takes every nb_xcd-th tile of that XCD's contiguous,
    // M-panel-major chunk of the tile space
    const int tiles_m = (int)((p.M + 255) / 256), tiles_n = p.tn;
    const int nwg = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int nb_xcd = (G - xcd + 7) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    // Tile order inside the chunk.  Static: every nb_xcd-th tile.  Dynamic (p.sched >= 0): the first tile by position, every further one from the XCD's
    // counter -- a block that starts late (the image tower's kernels on the side stream hold CUs when this launch begins: 144 KiB of LDS, one block per
    // CU) or runs slower takes fewer tiles instead of stretching the launch by its whole fixed share.  The fetch is issued by wave 0 in k-step 0 of a
    // tile (behind phase 3's counted wait, so it is the oldest entry of the in-order queue for one k-step only), is covered by k-step 1's wait, goes
    // through LDS to the other waves and is used from k-step nk - 2 on (the refill that crosses into the next tile): nk >= 6, host check.
    const bool dyn = SC_8P_DYN && p.sched >= 0;
    unsigned long long* const sched = g_sched + (dyn ? p.sched : 0) * 8;
    if (slot_in_xcd >= cnt) return;
    // arm my XCD's counter for this launch's generation (see g_sched); the wait behind the bias staging below covers it, ahead of this block's first fetch
    if (dyn && tid == 0) (void)__hip_atomic_fetch_max(sched + xcd, (unsigned long long)p.sched_gen << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Column bands (p.band N tiles, 0 = all of N): the order walks every M panel of a band before the next band, so an XCD has only the band's slice of
    // W (band x 256 x K) in flight: with all of N in flight (QKV 3.5 MiB, fc1 4.7 MiB of W against a 4 MiB L2) W is re-fetched once per tile round.
    const int band = p.band;
    auto tile_mn = [&](int idx, int& tm, int& tn) {      // idx: position inside the XCD's chunk
        int v = begin + idx;
        int nb = tiles_n, tn0 = 0;
        if (band > 0 && band < tiles_n) {
            const int per_band = band * tiles_m;
            const int b = v / per_band;
            tn0 = b * band;
            v -= b * per_band;
            nb = tiles_n - tn0 < band ? tiles_n - tn0 : band;
        }
        tm = v / nb;
        tn = tn0 + (v - tm * nb);
    };

    const int frow = lane & 15, fk = lane >> 4;
    const int off_h0 = frow * 128 + ((fk ^ (frow & 7)) << 4);
    const int off_h1 = frow * 128 + (((4 + fk) ^ (frow & 7)) << 4);
    const int a_base = g * HT;
    const int b_base = 2 * HT + (w4 >> 1) * HT + (w4 & 1) * 64 * 128;

    const int lr = lane >> 3, lc = lane & 7;
    const int64_t lane_a = (int64_t)(wave * 8 + lr) * p.lda + ((lc ^ lr) << 3);
    const int64_t lane_w = (int64_t)(wave * 8 + lr) * p.ldw + ((lc ^ lr) << 3);
    const int64_t lda64 = 64 * p.lda, ldw64 = 64 * p.ldw;
    auto stage_a = [&](const bf16_t* ta, int h, int k0, char* buf) {      // k0: element offset of the k-chunk
        if (SC_8P_ABL == 2) return;
        const bf16_t* src = ta + (int64_t)h * 128 * p.lda + k0 + lane_a;
        char* dst = buf + h * HT + wave * 1024;
        glds16(src, dst);
        glds16(src + lda64, dst + 8192);
    };
    auto stage_b = [&](const bf16_t* tw, int h, int k0, char* buf) {
        if (SC_8P_ABL == 2) return;
        const bf16_t* src = tw + (int64_t)h * 128 * p.ldw + k0 + lane_w;
        char* dst = buf + (2 + h) * HT + wave * 1024;
        glds16(src, dst);
        glds16(src + ldw64, dst + 8192);
    };

    // bias: the whole vector sits in the 32 KiB of LDS the two operand buffers leave free (N <= 8192, host check) and is read per tile with ds_read
    // (lgkmcnt): an ordinary global load inside the persistent loop makes hipcc wait vmcnt(0) in front of its first use in EVERY pass of the loop,
    // which drains the LDS-DMA pipeline
    for (int n = tid * 4; n < p.N; n += 512 * 4)
        *(f32x4_t*)(smem + 2 * BUF + n * 4) = p.bias ? *(const f32x4_t*)(p.bias + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const char* lds_bias = smem + 2 * BUF;

    f32x4_t acc[8][4];
    bf16x8_t af[4][2], bf_[2][2][2];

    // k walk of a tile: step j reads chunk (j + rot) mod nk of the walk, rot = M-panel index mod nk (the blocks of an XCD would otherwise all pull the
    // SAME W rows at the same instant: L2 channel de-correlation, as gemm256_kernel); kpair > 0 (stride-2 kernel-3 conv as GEMM, K = 3C, lda = 2C):
    // the walk visits (tap 0 chunk c, tap 2 chunk c) pairs, then tap 1 -- tap 2 of row r IS tap 0 of row r + 1, so the two touches of those lines are one
    // k-step apart (an L2 hit) instead of 2C / 64 k-steps apart.  A sum over K is order-free.
    const int kpair = p.kpair;
    auto kofs = [&](int j, int rot) -> int {
        int c = j + rot; c = c >= nk ? c - nk : c;
        if (kpair > 0) c = c < 2 * kpair ? (c >> 1) + (c & 1) * 2 * kpair : c - kpair;
        return c * 64;
    };
    // a tile that would cross M is shifted back to END at M (every row in range); its stores skip the rows that belong to the previous tile
    auto tile_a = [&](int tm_) -> const bf16_t* { const int64_t m = (int64_t)tm_ * 256; return p.A + (m + 256 <= p.M ? m : p.M - 256) * p.lda; };

    int tm, tn;
    int cur = slot_in_xcd;
    tile_mn(cur, tm, tn);
    const bf16_t* ta = tile_a(tm);
    const bf16_t* tw = p.W + (int64_t)tn * 256 * p.ldw;
    int rot = p.rows == 1 ? tm % nk : p.rows == 2 ? tn % nk : p.rows == 3 ? (2 * tn) % nk : 0;      // (p.rows reused as the rotation switch: 0 = off)
    {
        char* b0 = smem; char* b1 = smem + BUF;
        const int k0 = kofs(0, rot), k1 = kofs(1, rot);
        stage_a(ta, 0, k0, b0); stage_a(ta, 1, k0, b0); stage_b(tw, 0, k0, b0); stage_b(tw, 1, k0, b0);
        stage_b(tw, 0, k1, b1); stage_b(tw, 1, k1, b1); stage_a(ta, 0, k1, b1); stage_a(ta, 1, k1, b1);          // nk >= 2 (host check)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (!PAIR && g == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // group 1 runs one barrier interval behind group 0

    // epilogue lane mapping (gemm.hip): lane L stores row L >> 2, 16-byte chunk L & 3 of a 32-column half
    const int srow = lane >> 2, schunk = lane & 3;
    const int src_fk = ((schunk & 1) << 1) | (schunk >> 1);
    const int bperm = (src_fk * 16 + srow) << 2;
    const int64_t cstep = 16 * p.ldc, rstep = 16 * p.ldr;

    auto read_a = [&](const char* buf, int a) {
        if (SC_8P_ABL == 1) { for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(af[i][0])); asm volatile("" : "+v"(af[i][1])); } return; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *(const bf16x8_t*)(buf + a_base + (a * 64 + i * 16) * 128 + off_h0);
            af[i][1] = *(const bf16x8_t*)(buf + a_base + (a * 64 + i * 16) * 128 + off_h1);
        }
    };
    auto read_b = [&](const char* buf, int b) {
        if (SC_8P_ABL == 1) { for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(bf_[b][j][0])); asm volatile("" : "+v"(bf_[b][j][1])); } return; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf_[b][j][0] = *(const bf16x8_t*)(buf + b_base + (b * 32 + j * 16) * 128 + off_h0);
            bf_[b][j][1] = *(const bf16x8_t*)(buf + b_base + (b * 32 + j * 16) * 128 + off_h1);
        }
    };
    auto quadrant = [&](auto atag, auto btag) {
        constexpr int a = decltype(atag)::value, b = decltype(btag)::value;
        if (SC_8P_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SC_8P_ABL != 3) acc[a * 4 + i][b * 2 + j] = mfma_16x16x32<F16>(bf_[b][j][h], af[i][h], acc[a * 4 + i][b * 2 + j]);
                    else asm volatile("" : "+v"(acc[a * 4 + i][b * 2 + j]) : "v"(bf_[b][j][h]), "v"(af[i][h]));
                }
        if (SC_8P_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto init_q = [&](auto atag, auto btag) {
        constexpr int a = decltype(atag)::value, b = decltype(btag)::value;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[a * 4 + ii][b * 2 + jj] = SC_8P_BIAS_INIT ? *(const f32x4_t*)(lds_bias + (tn * 256 + w4 * 64 + (b * 2 + jj) * 16 + fk * 4) * 4)
                                                              : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    auto mem_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mat_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if (SC_8P_ABL != 4) __builtin_amdgcn_s_barrier();      // ABL 4 (timing probe, racy): one barrier per phase only
        asm volatile("" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    char* bx = smem;
    char* by = smem + BUF;
    unsigned long long tr_loop = 0, tr_epi = 0, tr_first = 0, tr_t = 0;
    const bool tracing = SC_PROBES && p.trace;
    if (tracing) tr_t = __builtin_readcyclecounter();
    const unsigned long long tr_begin = tr_t;
    unsigned long long fetched = 0;
    int n_tiles = 0;
    for (;;) {
        bool have_next = false;
        int nxt = cur, ntm = tm, ntn = tn, rot_n = 0;
        const bf16_t* ta_n = ta;
        const bf16_t* tw_n = tw;
        auto set_next = [&](int idx) {
            nxt = idx;
            have_next = idx < cnt;
            if (have_next) tile_mn(idx, ntm, ntn);
            ta_n = tile_a(ntm);
            tw_n = p.W + (int64_t)ntn * 256 * p.ldw;
            rot_n = p.rows == 1 ? ntm % nk : p.rows == 2 ? ntn % nk : p.rows == 3 ? (2 * ntn) % nk : 0;
        };
        if (!dyn) set_next(cur + nb_xcd);
        const bool no_wait0 = RES && SC_8P_RES_ASM && !SC_8P_RES_LATE && n_tiles > 0 && nk >= 3;      // (the same for the plain variants: +-0, twice)
        ++n_tiles;
        // Barrier pairing at a tile boundary.  Group 1 runs one barrier interval behind group 0 inside a tile.  With the same pairing across the boundary
        // the two epilogues land in DIFFERENT intervals (group 0's beside group 1's last cluster, group 1's beside group 0's first cluster of the next
        // tile) and run one after the other: the trace showed group 0 waiting a whole epilogue at its first barrier (first k-step 6.5 k cycles plain,
        // 10 k residual / GELU, against 3 k).  So group 1 drops the barrier behind its last cluster (nothing after it touches an operand buffer before the
        // barriers of the next tile's k-step 0) and enters every tile with one extra barrier: both epilogues sit between the same two barriers.
        if (PAIR && g == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        // (Measured alternatives of this schedule, round 5, same box, TF/s qkv / out / fc2: this one 1128 / 1100 / 1306; TWO phases of 32 MFMAs per k-step --
        //  half as many barriers -- 1075 / 997 / 1275; ONE barrier per phase with a leader / follower order of the two groups inside the interval
        //  1100 / 1015 / 1235.  EXPERIMENTS.md.)
        for (int kt = 0; kt < nk; ++kt) {
            const bool first = kt == 0;
            if (tracing && kt == 1) { const unsigned long long c = __builtin_readcyclecounter(); tr_first += c - tr_t; tr_t = c; }
            // Refill: the whole of k-step + 2 (possibly the next tile's) is issued from phases 2 and 3 of this k-step into THIS k-step's buffer: its
            // W halves are free after phase 1 (every wave holds its W fragments in registers), its A halves after phase 2.  Everything is requested a
            // full k-step before its first reader, and -- tile boundary -- the next tile's first TWO k-steps are requested before the epilogue's
            // stores enter the queue, so they do not wait behind them (in-order retirement).
            // Residual variants, last k-step of a tile: the refill (the next tile's k-step 1) is issued from the EPILOGUE instead, behind its first residual
            // loads -- vector-memory results return in order, so residual loads queued behind 8 freshly issued DMA pieces (first-touch A lines from HBM)
            // cannot be consumed before those have landed, and the epilogue opens with the matrix pipe idle for that long (SC_8P_RES_LATE=0: old order).
            const bool late = RES && SC_8P_RES_LATE && kt == nk - 1;
            const bool s_ok = (kt + 2 < nk || have_next) && !late;
            const bf16_t* sa = kt + 2 < nk ? ta : ta_n;
            const bf16_t* sb = kt + 2 < nk ? tw : tw_n;
            const int s_k = kt + 2 < nk ? kofs(kt + 2, rot) : kofs(kt + 2 - nk, rot_n);
            // (A "balanced" fragment-read schedule -- quadrant order (a0,b0) (a0,b1) (a1,b0) (a1,b1), the next k-step's b0 read in phase 3's memory interval, the
            //  k-half-1 fragments of a0 requested at the head of phase 0's cluster: 4 / 4 / 8 / 4 reads per memory interval instead of 12 / 4 / 8 / 0 -- measured
            //  3 % SLOWER on every shape (fc2 1235 vs 1270, conv1 1125 vs 1158 TF/s, same box), round 5.  EXPERIMENTS.md.)
            // (Fragment reads of phases 1 and 2 issued at the tail of the previous matrix interval -- behind the cluster's last MFMA, in front of the barrier, so
            //  that their LDS latency runs across the barrier: measured 1-3 % SLOWER on every shape, round 5.  EXPERIMENTS.md R5-3.)
            // (Deferred output stores -- 10 of the 16 stores per wave of the finished tile kept in registers / 2 KiB of LDS and issued one per phase 0 and one
            //  behind phase 3's wait of the next tile's first k-steps: 2-14 % SLOWER (QKV 946 vs 1097, fc1 880 vs 999 TF/s): the stores share the in-order
            //  vmcnt queue with the LDS-DMA refills, so the counted waits also wait for store acknowledgements.  Round 5, EXPERIMENTS.md R5-3.)
            if (dyn && kt == 2) set_next(nb_xcd + __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)(smem + SCHED_LDS)));
            // ---- phase 0: quadrant (a0, b0)
            if (first) init_q(I0{}, I0{});
            read_a(bx, 0); read_b(bx, 0);
            mem_end();      // (waiting for phase 0's reads only in front of the MFMAs that need them: +-1 %, round 5)
            quadrant(I0{}, I0{});
            mat_end();
            // ---- phase 1: quadrant (a0, b1)
            if (first) init_q(I0{}, I1{});
            read_b(bx, 1);
            mem_end();
            quadrant(I0{}, I1{});
            mat_end();
            // ---- phase 2: quadrant (a1, b1); both W halves of k-step + 2
            if (first) init_q(I1{}, I1{});
            read_a(bx, 1);
            if (s_ok) { stage_b(sb, 0, s_k, bx); stage_b(sb, 1, s_k, bx); }
            mem_end();
            quadrant(I1{}, I1{});
            mat_end();
            // ---- phase 3: quadrant (a1, b0), operands already in registers; both A halves of k-step + 2; k-step + 1 must have landed: in-order,
            //      everything but the 8 pieces of phases 2 and 3
            if (first) init_q(I1{}, I0{});
            // (residual variants, k-step 0 of every tile but the block's first: the previous epilogue waited vmcnt(0) -- this tile's k-step 1 included -- before
            //  its first store, so no counted wait here: it would wait for that epilogue's whole store burst to be acknowledged)
            if (s_ok) { stage_a(sa, 0, s_k, bx); stage_a(sa, 1, s_k, bx); if (!(first && no_wait0)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (dyn && wave == 0 && kt < 2) {
                if (kt == 0) {      // lane 0 only (exec switched inside the asm: no per-lane select on `fetched`, which is in flight until k-step 1's wait)
                    unsigned long long ex;
                    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add_x2 %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                                 : "+v"(fetched), "=&s"(ex) : "v"(sched + xcd), "v"(1ull) : "memory");
                } else if (lane == 0) {
                    *(volatile __attribute__((address_space(3))) unsigned int*)(smem + SCHED_LDS) = (unsigned int)fetched;      // low word = the count (k-step 1's wait above covered the fetch)
                }
            }
            // (residual variants: touching the tile's residual lines two k-steps ahead -- 4-byte LDS-DMA loads into a dummy LDS area, gemm256_kernel's trick --
            //  measured in the step, round 5: out-proj 789 -> 761, fc2 1170 -> 1112 TF/s.  Not kept.)
            mem_end();
            quadrant(I1{}, I0{});
            if (PAIR && g == 1 && kt == nk - 1) __builtin_amdgcn_sched_barrier(0); else mat_end();
            { char* x = bx; bx = by; by = x; }
        }
        if (tracing) { const unsigned long long c = __builtin_readcyclecounter(); tr_loop += c - tr_t; tr_t = c; }
        // ---- epilogue: 16 rows x 64 contiguous bytes per store (gemm.hip's fast path); the next tile's first k-step is landing meanwhile
        {
            // (residual variants with the explicit residual pipeline re-read the bias from LDS at every use: 16 VGPRs the epilogue needs elsewhere)
            constexpr bool BIAS_LDS = (RES && SC_8P_RES_ASM) || SC_8P_BIAS_INIT;
            f32x4_t bias4[BIAS_LDS ? 1 : 4];
            const char* bias_at = lds_bias + (tn * 256 + w4 * 64 + fk * 4) * 4;
            if constexpr (!BIAS_LDS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bias4[j] = *(const f32x4_t*)(bias_at + j * 64);
            }
            auto bias_of = [&](int j) -> f32x4_t {
                if constexpr (BIAS_LDS) return *(volatile const __attribute__((address_space(3))) f32x4_t*)(bias_at + j * 64);
                else return bias4[j];
            };
            auto biased = [&](const f32x4_t& v, int j) -> f32x4_t {
                if constexpr (SC_8P_BIAS_INIT) return v;      // the accumulators started from the bias
                else return v + bias_of(j);
            };
            const int64_t m_lo = (int64_t)tm * 256, m0 = m_lo + 256 <= p.M ? m_lo : p.M - 256;
            const int skip = (int)(m_lo - m0) - g * 128 - srow;            // rows i * 16 + .. below this belong to the previous tile (ragged last M panel)
            const int64_t mrow0 = m0 + g * 128 + srow;
            const int ncol0 = tn * 256 + w4 * 64 + schunk * 8;
            if (F32) {
                // fp32 outputs (pre-LN / ViT residual streams: C and the residual are f32): lane (frow, fk) holds 4 consecutive fp32 = 16 bytes of row
                // frow; lane L fetches (LDS crossbar) the value of lane (L >> 2) + 16 (L & 3), so that the 4 lanes of a quad own one 64-byte segment and a
                // store instruction writes 16 rows x 64 contiguous bytes per 16-column block (the quad rule of the bf16 path).
                const int bperm32 = (srow + 16 * schunk) << 2;
                const int ncol32 = tn * 256 + w4 * 64 + schunk * 4;
                float* cf = (float*)p.C + mrow0 * p.ldc + ncol32;
                const float* rf = RES ? (const float*)p.residual + mrow0 * p.ldr + ncol32 : nullptr;
                if constexpr (RES && SC_8P_RES_ASM) {
                    // Residual tile through EXPLICIT loads and counted waits.  Left to the compiler, a residual row block read ahead of its use sits in the
                    // queue with the output stores of the blocks before it, and hipcc (loads and stores pending on the one vmcnt counter) waits vmcnt(0) in
                    // front of every use: each row block paid a full HBM round trip plus the acknowledgement of all earlier stores.  Here the queue holds only
                    // loads until the last residual has been added: two row blocks in flight (32 VGPRs, 8 KiB per wave), refilled as they are used, waits
                    // counted exactly (loads return in order); the outputs replace the accumulators and all stores go out at the end, after vmcnt(0).
                    f32x4_t rs[2][4];
                    auto load_rs = [&](int i) {
                        const float* a = rf + i * rstep;
                        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
                                     "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                                     : "=&v"(rs[i & 1][0]), "=&v"(rs[i & 1][1]), "=&v"(rs[i & 1][2]), "=&v"(rs[i & 1][3]) : "v"(a) : "memory");
                    };
                    load_rs(0); load_rs(1);
                    auto conv = [&](int i) {      // bias, activation, lane exchange of row block i, in place
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4_t v4 = biased(acc[i][j], j);
                            if (ACT == SC_ACT_GELU) {
                                const f32x2_t g0 = gelu_poly2_f32((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2_f32((f32x2_t){v4[2], v4[3]});
                                v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                            } else if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm32, __float_as_int(v4[r])));
                        }
                    };
                    conv(0); conv(1);      // two row blocks ahead of the add: work under the first residual's round trip
                    auto blk = [&](auto itag) {
                        constexpr int i = decltype(itag)::value;
                        if (i + 2 < 8) conv(i + 2 < 8 ? i + 2 : 7);
                        constexpr int newer = (i + 1 < 8 ? 1 : 0) * 4;      // loads issued behind row block i's
                        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rs[i & 1][0]), "+v"(rs[i & 1][1]), "+v"(rs[i & 1][2]), "+v"(rs[i & 1][3]) : "n"(newer) : "memory");
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] += rs[i & 1][j];
                        if (i + 2 < 8) load_rs(i + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    blk(std::integral_constant<int, 0>{}); blk(std::integral_constant<int, 1>{}); blk(std::integral_constant<int, 2>{}); blk(std::integral_constant<int, 3>{});
                    blk(std::integral_constant<int, 4>{}); blk(std::integral_constant<int, 5>{}); blk(std::integral_constant<int, 6>{}); blk(std::integral_constant<int, 7>{});
                    // (vmcnt is 0 here: the next tile's k-step 1 has landed too -- no_wait0 above)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float* dst = i * 16 >= skip ? cf + i * cstep : (float*)(g_sink + lane * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) *(f32x4_t*)(dst + j * 16) = acc[i][j];
                    }
                } else {
                f32x4_t rs[RES ? 2 : 1][4];
                auto load_rs = [&](int i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rs[i & 1][j] = *(const f32x4_t*)(rf + i * rstep + j * 16);
                };
                if (RES) { load_rs(0); load_rs(1); }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4_t o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4_t v4 = biased(acc[i][j], j);
                        if (ACT == SC_ACT_GELU) {       // f32 output: the fp32 polynomial (the packed-half form carries ~11 bits, meant for bf16 results)
                            const f32x2_t g0 = gelu_poly2_f32((f32x2_t){v4[0], v4[1]}), g1 = gelu_poly2_f32((f32x2_t){v4[2], v4[3]});
                            v4 = (f32x4_t){g0[0], g0[1], g1[0], g1[1]};
                        } else if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[j][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm32, __float_as_int(v4[r])));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (RES) o[j] += rs[i & 1][j];
                        if (i * 16 >= skip) *(f32x4_t*)(cf + i * cstep + j * 16) = o[j];
                    }
                    if (RES && i + 2 < 8) load_rs(i + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            } else {
            bf16_t* cptr = (bf16_t*)p.C + mrow0 * p.ldc + ncol0;
            const bf16_t* rptr = RES ? (const bf16_t*)p.residual + mrow0 * p.ldr + ncol0 : nullptr;
            uint4 res[RES ? 4 : 1][2];
            auto load_res = [&](int i) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) res[i & 3][jp] = *(const uint4*)(rptr + i * rstep + jp * 32);
            };
            if (RES && !SC_8P_RES_ASM) { load_res(0); load_res(1); load_res(2); load_res(3); }
            if (RES && SC_8P_RES_LATE && have_next) {      // the refill the last k-step left out (bx / by are swapped by now: `by` held k-step nk - 1)
                const int k1 = kofs(1, rot_n);
                stage_b(tw_n, 0, k1, by); stage_b(tw_n, 1, k1, by); stage_a(ta_n, 0, k1, by); stage_a(ta_n, 1, k1, by);
            }
            auto shuffled = [&](int i, uint4 (&o)[2]) {
                uint2 pk[4];
                if (ACT == SC_ACT_GELU) {          // all 8 value pairs of the row block side by side: gelu_poly2_x8 (common.h)
                    f32x2_t xs[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4_t v4 = biased(acc[i][j], j);
                        xs[2 * j] = (f32x2_t){v4[0], v4[1]}; xs[2 * j + 1] = (f32x2_t){v4[2], v4[3]};
                    }
                    gelu_poly2_x8<F16>(xs);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pk[j].x = pack2x<F16>(xs[2 * j][0], xs[2 * j][1]); pk[j].y = pack2x<F16>(xs[2 * j + 1][0], xs[2 * j + 1][1]); }
                } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4_t v4 = biased(acc[i][j], j);
                    if (ACT == SC_ACT_QUICKGELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = quick_gelu(v4[r]);
                    }
                    pk[j].x = pack2x<F16>(v4[0], v4[1]);
                    pk[j].y = pack2x<F16>(v4[2], v4[3]);
                }
                }
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].x, pk[2 * jp + 1].x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].y, pk[2 * jp + 1].y, false, false);
                    o[jp] = make_uint4(__builtin_amdgcn_ds_bpermute(bperm, r0[0]), __builtin_amdgcn_ds_bpermute(bperm, r1[0]),
                                       __builtin_amdgcn_ds_bpermute(bperm, r0[1]), __builtin_amdgcn_ds_bpermute(bperm, r1[1]));
                }
            };
            if constexpr (RES && SC_8P_RES_ASM) {
                // (explicit residual loads, as the fp32 path above: four row blocks in flight -- 32 VGPRs, 8 KiB per wave -- refilled as they are used, counted
                //  waits, results kept, every store behind the last add)
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                u32x4_t rv[4][2];
                uint4 ov[8][2];
                auto load_rv = [&](int i) {
                    const bf16_t* a = rptr + i * rstep;
                    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64" : "=&v"(rv[i & 3][0]), "=&v"(rv[i & 3][1]) : "v"(a) : "memory");
                };
                load_rv(0); load_rv(1); load_rv(2); load_rv(3);
                shuffled(0, ov[0]); shuffled(1, ov[1]);      // the conversion runs two row blocks ahead of the add: work under the first residual's round trip
                auto blk = [&](auto itag) {
                    constexpr int i = decltype(itag)::value;
                    if (i + 2 < 8) shuffled(i + 2, ov[i + 2 < 8 ? i + 2 : 7]);
                    constexpr int newer = (i + 3 < 8 ? 3 : 7 - i) * 2;      // loads issued behind row block i's
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rv[i & 3][0]), "+v"(rv[i & 3][1]) : "n"(newer) : "memory");
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        uint4 o = ov[i][jp];
                        const uint4 r4 = make_uint4(rv[i & 3][jp][0], rv[i & 3][jp][1], rv[i & 3][jp][2], rv[i & 3][jp][3]);
                        o.x = pack2x<F16>(lo2fx<F16>(o.x) + lo2fx<F16>(r4.x), hi2fx<F16>(o.x) + hi2fx<F16>(r4.x));
                        o.y = pack2x<F16>(lo2fx<F16>(o.y) + lo2fx<F16>(r4.y), hi2fx<F16>(o.y) + hi2fx<F16>(r4.y));
                        o.z = pack2x<F16>(lo2fx<F16>(o.z) + lo2fx<F16>(r4.z), hi2fx<F16>(o.z) + hi2fx<F16>(r4.z));
                        o.w = pack2x<F16>(lo2fx<F16>(o.w) + lo2fx<F16>(r4.w), hi2fx<F16>(o.w) + hi2fx<F16>(r4.w));
                        ov[i][jp] = o;
                    }
                    if (i + 4 < 8) load_rv(i + 4);
                    __builtin_amdgcn_sched_barrier(0);
                };
                blk(std::integral_constant<int, 0>{}); blk(std::integral_constant<int, 1>{}); blk(std::integral_constant<int, 2>{}); blk(std::integral_constant<int, 3>{});
                blk(std::integral_constant<int, 4>{}); blk(std::integral_constant<int, 5>{}); blk(std::integral_constant<int, 6>{}); blk(std::integral_constant<int, 7>{});
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    bf16_t* dst = i * 16 >= skip ? cptr + i * cstep : (bf16_t*)(g_sink + lane * 16);
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) *(uint4*)(dst + jp * 32) = ov[i][jp];
                }
            } else {
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
                        o.x = pack2x<F16>(lo2fx<F16>(o.x) + lo2fx<F16>(rv.x), hi2fx<F16>(o.x) + hi2fx<F16>(rv.x));
                        o.y = pack2x<F16>(lo2fx<F16>(o.y) + lo2fx<F16>(rv.y), hi2fx<F16>(o.y) + hi2fx<F16>(rv.y));
                        o.z = pack2x<F16>(lo2fx<F16>(o.z) + lo2fx<F16>(rv.z), hi2fx<F16>(o.z) + hi2fx<F16>(rv.z));
                        o.w = pack2x<F16>(lo2fx<F16>(o.w) + lo2fx<F16>(rv.w), hi2fx<F16>(o.w) + hi2fx<F16>(rv.w));
                    }
                    if (i * 16 >= skip) *(uint4*)(cptr + i * cstep + jp * 32) = o;
                }
                if (RES && i + 4 < 8) load_res(i + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
                    }
        }
        if (tracing) { const unsigned long long c = __builtin_readcyclecounter(); tr_epi += c - tr_t; tr_t = c; }
        if (!have_next) break;
        cur = nxt; tm = ntm; tn = ntn; ta = ta_n; tw = tw_n; rot = rot_n;
    }
    if (!PAIR && g == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    if (tracing && lane == 0) {      // per wave: first k-step of every tile / the other k-steps / epilogue issue, block lifetime, tiles
        unsigned long long* tr = p.trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        tr[0] = tr_first; tr[1] = tr_loop; tr[2] = tr_epi; tr[3] = __builtin_readcyclecounter() - tr_begin; tr[4] = n_tiles;
    }
}

template <int ACT, bool RES, bool F32, bool F16 = false>
int launch_pers(const Gemm8pParams& p, int grid, hipStream_t s) {
    constexpr int lds = 2 * BUF + 32768;   // 128 KiB of operands + the bias vector
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm8p_pers_kernel<ACT, RES, F32, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm8p_pers_kernel<ACT, RES, F32, F16>), dim3(grid), dim3(512), lds, s, p);
    SC_CHECK_LAUNCH();
    return 0;
}

template <int ACT, bool RES>
int launch_one(const Gemm8pParams& p, int grid, hipStream_t s) {
    constexpr int lds = 2 * BUF;   // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm8p_kernel<ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm8p_kernel<ACT, RES>), dim3(grid), dim3(512), lds, s, p);
    SC_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Test hook (exported, not part of include/speechclip_hip.h): overwrite the whole counter ring with what a launch that died mid-flight could leave behind --
// an OLD generation with every tile "already taken".  Synchronous.
extern "C" int sc_debug_poison_gemm_sched(void) {
    static unsigned long long junk[SCHED_RING * 8];
    for (int i = 0; i < SCHED_RING * 8; ++i) junk[i] = 0x000000007fffff00ull + (unsigned)i;      // generation 0, count ~2^31
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sched), junk, sizeof(junk)) == hipSuccess ? 0 : -1;
}

int sc_gemm8p_try(const Gemm8pParams& pin, hipStream_t s) {
    Gemm8pParams p = pin;
    if (p.N % 256 || p.K % 64 || p.M < 256) return 1;
    if (p.M % 256 && (p.esteps == 1 || p.N > 8192)) return 1;      // the per-tile A/B kernel takes full M panels only
    p.tn = p.N / 256; p.nk = p.K / 64;
    if (p.nk < 2) return 1;
    if (p.ldc % (p.out_f32 ? 4 : 8) || (p.residual && p.ldr % (p.out_f32 ? 4 : 8)) || p.lda % 8 || p.ldw % 8) return 1;
    const int64_t tiles = ((p.M + 255) / 256) * p.tn;
    if (tiles > 0x7fffffff) return 1;
    const int grid = (int)tiles;
    const bool res = p.residual != nullptr;
    if (p.esteps != 1 && p.N <= 8192) {       // persistent form (esteps == 1: the plain per-tile kernel, A/B)
        static int n_cu = 0;
        if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
        const int pg = grid < n_cu ? grid : n_cu;
        // dynamic tile order: a slot of the counter ring per launch (SCHED_RING launches would have to be in flight at once for two to meet).  A launch
        // that is being CAPTURED into a HIP graph keeps the static order: its slot would be frozen into the graph node, and a replay could then run
        // beside an eager launch that the ring has handed the same slot.
        static std::atomic<unsigned long long> sched_seq{0};
        bool dyn = SC_8P_DYN && p.sched >= 0 && p.nk >= 6 && p.N < 8192 && pg >= 8;
        if (dyn) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) dyn = false;
        }
        if (dyn) { const unsigned long long seq = sched_seq.fetch_add(1); p.sched = (int)(seq % SCHED_RING); p.sched_gen = (unsigned)(seq / SCHED_RING + 1); }
        else p.sched = -1;
        // (ACT x RES x F32 x F16 as template arguments: one switch over the packed key)
#define SC_8P_CASE(A_, R_, F_, H_) case ((A_) | ((R_) << 2) | ((F_) << 3) | ((H_) << 4)): return launch_pers<A_, R_ != 0, F_ != 0, H_ != 0>(p, pg, s);
#define SC_8P_CASES(R_, F_, H_) SC_8P_CASE(SC_ACT_NONE, R_, F_, H_) SC_8P_CASE(SC_ACT_GELU, R_, F_, H_) SC_8P_CASE(SC_ACT_QUICKGELU, R_, F_, H_)
        switch ((p.act & 3) | ((res ? 1 : 0) << 2) | ((p.out_f32 ? 1 : 0) << 3) | ((p.f16 ? 1 : 0) << 4)) {
            SC_8P_CASES(0, 0, 0) SC_8P_CASES(1, 0, 0) SC_8P_CASES(0, 1, 0) SC_8P_CASES(1, 1, 0)
            SC_8P_CASES(0, 0, 1) SC_8P_CASES(1, 0, 1) SC_8P_CASES(0, 1, 1) SC_8P_CASES(1, 1, 1)
            default: return 1;
        }
#undef SC_8P_CASES
#undef SC_8P_CASE
    }
    if (p.out_f32 || p.f16) return 1;
    switch (p.act) {
        case SC_ACT_GELU: return res ? launch_one<SC_ACT_GELU, true>(p, grid, s) : launch_one<SC_ACT_GELU, false>(p, grid, s);
        case SC_ACT_QUICKGELU: return res ? launch_one<SC_ACT_QUICKGELU, true>(p, grid, s) : launch_one<SC_ACT_QUICKGELU, false>(p, grid, s);
        default: return res ? launch_one<SC_ACT_NONE, true>(p, grid, s) : launch_one<SC_ACT_NONE, false>(p, grid, s);
    }
}
