// ROUND-5 EXPERIMENT, NOT PART OF THE PRODUCT (moved out of speechclip_amd/csrc at the end of round 5; builds against csrc/gemm8p.h of that revision with the
// rows / units / esteps fields in their original meaning).  Result: NEGATIVE, EXPERIMENTS.md R5-2.
// bf16 MFMA GEMM, "duet" form for gfx950 (CDNA4):  C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual, bf16 out.
//
// Why a second kernel.  gemm256_kernel (gemm.hip) runs ONE 256x256 tile per CU at a time: all 8 waves reach the tile's epilogue together, so the
// matrix pipe idles for ~12 k cycles per tile (epilogue + tile-start wait + next-tile prologue) -- 25 % of a K = 768 tile.  Here the two wave
// groups of the block (waves 0-3 / 4-7: one wave of each group on every SIMD) work on DIFFERENT 128-row half panels of A against the SAME 256
// columns of W, E k-steps out of phase.  A block keeps its N tile for the whole launch and the k walk is cyclic (stage t holds k-chunk t mod nk
// for everybody: a sum over K is order-free), so the W stage in LDS is shared by the two groups no matter where each is in its own tile, and the
// LDS / L2 traffic per k-step is what the 256x256 tile had (2 x 16 KiB of A + 32 KiB of W).  While one group converts and stores its 128x256
// half tile (E "solo" k-steps, no MFMA), the other group's waves have the SIMDs' matrix pipes to themselves and also issue the whole LDS-DMA
// refill; in the remaining nk - E "joint" k-steps both groups compute and split the refill.
//
// Work split.  Block b lives on XCD b % 8.  Blocks are ranked XCD-major and cut into ROWS of tn = N / 256 blocks: the blocks of a row hold the
// tn different N tiles and walk the SAME list of half panels in lock-step (nothing enforces it: they start together and run the same code),
// so an A half panel is fetched into the XCD's L2 by one of them and hit by the others.  256 mod tn blocks stay idle (tn = 9, 12: 4 of 256).
//
// Pipeline per k-step t (one raw s_barrier per step, counted vmcnt; same register pipeline as gemm256_kernel):
//   first half   32 MFMA on (t, h0) | read fragments (t, h1)   | LDS-DMA: A pieces of stage t+2 -> A slot (t+2) % 3
//   wait own pieces of stage t+1, s_barrier
//   second half  32 MFMA on (t, h1) | read fragments (t+1, h0) | LDS-DMA: W pieces of stage t+2 -> W slot t % 2
// Epilogue steps execute their share of the half tile's 8 row blocks and the same barrier; they issue no DMA and read no LDS memory.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "gemm8p.h"
#include "../../include/speechclip_hip.h"

#ifndef SC_PROBES
#define SC_PROBES 0
#endif
#ifndef SC_DUET_ABL           // timing ablations (garbage results): 1 no in-loop LDS-DMA, 2 no MFMA, 3 no epilogue stores, 4 empty epilogue, 5 no fragment reads
#define SC_DUET_ABL 0
#endif
#ifndef SC_DUET_PRIO          // 1: compute-phase waves run at s_setprio 1, epilogue-phase waves at 0
#define SC_DUET_PRIO 0
#endif

namespace {

constexpr int HALF_SLOT = 256 * 128;   // 32 KiB: one operand of one stage (A: two groups x 128 rows x 128 B; W: 256 rows x 128 B)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int ACT, bool RES, int E>
__global__ __launch_bounds__(512) void gemm_duet_kernel(Gemm8pParams p) {
    static_assert(E == 4 || E == 8, "epilogue steps");
    constexpr int RB = 8 / E;                      // row blocks (16 rows each) per epilogue step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, w4 = wave & 3;        // group, wave within the group (its 64-column strip)
    const int nk = p.nk, P = nk + E;

    // ---- block -> (row of blocks, N tile)
    const int G8 = gridDim.x >> 3;
    const int rank = (blockIdx.x & 7) * G8 + (blockIdx.x >> 3);
    const int row = rank / p.tn, tn_i = rank - row * p.tn;
    if (row >= p.rows) return;
    const int64_t u_begin = (int64_t)row * p.units / p.rows, u_end = (int64_t)(row + 1) * p.units / p.rows;
    const int cnt = (int)(u_end - u_begin);
    const int nt0 = (cnt + 1) >> 1, nt1 = cnt >> 1;                 // half tiles of group 0 (units u_begin + 0, 2, ..) / group 1 (+ 1, 3, ..)
    const int n0 = tn_i * 256;
    const int T = nt0 > nt1 ? nt0 * P : E + nt1 * P;                // k-steps of this block (group 1 runs E steps behind group 0)

    const bf16_t* tw = p.W + (int64_t)n0 * p.ldw;
    auto unit_m0 = [&](int64_t u) -> int64_t { const int64_t m = u * 128; return m + 128 <= p.M ? m : p.M - 128; };
    auto tile_ptr = [&](int h, int q) -> const bf16_t* { return p.A + unit_m0(u_begin + 2 * (int64_t)q + h) * p.lda; };

    // ---- k walk: stage j holds chunk (j + rot0) mod nk of the walk (rows of blocks start at different chunks: L2 channel de-correlation)
    const int kpair = p.kpair;
    auto kmap = [&](int c) -> int {
        if (kpair > 0) c = c < 2 * kpair ? (c >> 1) + (c & 1) * 2 * kpair : c - kpair;
        return c * 64;
    };
    const int rot0 = row % nk;

    // ---- fragment read offsets (same LDS image as gemm256_kernel: [row][8 x 16 B], chunk position c of row r holds k-chunk c ^ (r & 7))
    const int frow = lane & 15, fk = lane >> 4;
    const int off_h0 = frow * 128 + ((fk ^ (frow & 7)) << 4);
    const int off_h1 = frow * 128 + (((4 + fk) ^ (frow & 7)) << 4);
    const int a_base = g * 128 * 128;
    const int b_base = w4 * 64 * 128;

    // ---- LDS-DMA pieces: 4 waves x 64 lanes x 16 B = 32 rows x 128 B
    const int lr = lane >> 3, lc = lane & 7;
    const int lane_a = (w4 * 8 + lr) * (int)p.lda + ((lc ^ lr) << 3);
    const int lane_w = (w4 * 8 + lr) * (int)p.ldw + ((lc ^ lr) << 3);
    const int64_t lda32 = 32 * p.lda, ldw32 = 32 * p.ldw;
    auto piece_a = [&](const bf16_t* ta, int h, int pc, int k0, char* slot) {
        if (SC_DUET_ABL == 1 || SC_DUET_ABL == 7) return;
        glds16(ta + (pc * lda32 + k0) + lane_a, slot + (h * 4 + pc) * 4096 + w4 * 1024);
    };
    auto piece_w = [&](int pc, int k0, char* slot) {
        if (SC_DUET_ABL == 1 || SC_DUET_ABL == 7) return;
        glds16(tw + (pc * ldw32 + k0) + lane_w, slot + pc * 4096 + w4 * 1024);
    };

    // ---- ring: A0 A1 A2 W0 W1 (32 KiB each), rotated once per k-step
    char* sA0 = smem; char* sA1 = smem + HALF_SLOT; char* sA2 = smem + 2 * HALF_SLOT;
    char* sW0 = smem + 3 * HALF_SLOT; char* sW1 = smem + 4 * HALF_SLOT;

    // ---- bias of this lane's 4 x 4 output columns (accumulator layout): loaded once, the N tile never changes
    f32x4_t bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n0 + w4 * 64 + j * 16 + fk * 4;
        bias4[j] = p.bias ? *(const f32x4_t*)(p.bias + nn) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bias4[j]));

    // ---- prologue: stages 0 and 1 (group 0 computes at t = 0, 1; group 1 starts at t = E >= 4)
    {
        const bf16_t* ta = tile_ptr(0, 0);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            int c = st + rot0; c = c >= nk ? c - nk : c;
            const int k0 = kmap(c);
            char* as = st ? sA1 : sA0; char* ws = st ? sW1 : sW0;
            if (g == 0) {
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) piece_a(ta, 0, pc, k0, as);
            }
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) piece_w(4 * g + pc, k0, ws);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    f32x4_t acc[8][4];
    bf16x8_t bfr[4], af[8];

    // per-group state: s = position inside the period (0 .. nk-1 compute, nk .. P-1 epilogue; < 0: not started), q = tile index
    int s0 = 0, q0 = 0, s1 = -E, q1 = 0;
    int kc2 = 2 + rot0; kc2 = kc2 >= nk ? kc2 - nk : kc2;          // chunk of the walk that stage t + 2 holds

    // epilogue addressing (lane L stores row L >> 2, 16-byte chunk L & 3 of a 32-column half: see gemm.hip)
    const int srow = lane >> 2, schunk = lane & 3;
    const int src_fk = ((schunk & 1) << 1) | (schunk >> 1);
    const int bperm = (src_fk * 16 + srow) << 2;
    const int ncol0 = n0 + w4 * 64 + schunk * 8;

    // PROBES: per-wave cycles: [0] solo work [1] solo barrier wait [2] joint work [3] joint barrier wait [4] epilogue work [5] epilogue barrier wait
    // [6] null steps [7] epilogue chunk 0: wait for this wave's DMA pieces
    unsigned long long trc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t0 = 0;
    const bool tracing = SC_PROBES && p.trace;
    auto TR = [&](int idx) { if (tracing) { const unsigned long long c = __builtin_readcyclecounter(); trc[idx] += c - tr_t0; tr_t0 = c; } };
    if (tracing) tr_t0 = __builtin_readcyclecounter();

    // counted wait: at most n of this wave's vector-memory operations may still be in flight (in-order retirement); n is even, <= 24
    auto wait_vm = [&](int n) {
        if (SC_DUET_ABL == 6) return;          // timing probe (garbage results): nobody waits for LDS-DMA data
        switch (n >> 1) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        }
    };

    for (int t = 0; t < T; ++t) {
        const int ph0 = (s0 < 0 || q0 >= nt0) ? 0 : (s0 < nk ? 1 : 2);
        const int ph1 = (s1 < 0 || q1 >= nt1) ? 0 : (s1 < nk ? 1 : 2);
        const int my_ph = g ? ph1 : ph0, other_ph = g ? ph0 : ph1;
        const int s_me = g ? s1 : s0, q_me = g ? q1 : q0;
        // who computes at t + 2, and on which tile: stage t + 2 is issued during this step
        int q0n = q0, q1n = q1;
        bool need0, need1;
        { int s2 = s0 + 2; if (s2 >= P) { s2 -= P; ++q0n; } need0 = s2 >= 0 && q0n < nt0 && s2 < nk; }
        { int s2 = s1 + 2; if (s2 >= P) { s2 -= P; ++q1n; } need1 = s2 >= 0 && q1n < nt1 && s2 < nk; }
        const bool needw = need0 || need1;
        const bf16_t* ta0 = tile_ptr(0, need0 ? q0n : 0);
        const bf16_t* ta1 = tile_ptr(1, need1 ? q1n : 0);
        const int k2 = kmap(kc2);

        if (my_ph == 1) {
            // ------------------------------------------------------------------------------------------ compute step
            // LDS-DMA duty: both groups computing: own A half panel + half of W each; partner in its epilogue: the partner issues everything (this
            // wave's MFMA stream never stalls on a full vector-memory queue); partner idle (first / last steps of the block): this group issues all.
            if (SC_DUET_PRIO) __builtin_amdgcn_s_setprio(1);
            const bool joint = other_ph == 1, all = other_ph == 0;
            const bool need_me = g ? need1 : need0;
            const bf16_t* ta_me = g ? ta1 : ta0;

            if (s_me == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8_t*)(sW0 + b_base + off_h0 + j * 16 * 128);
#pragma unroll
                for (int i = 0; i < 8; ++i) af[i] = *(const bf16x8_t*)(sA0 + a_base + off_h0 + i * 16 * 128);
            }

            // ---- first half: MFMAs of (t, h0); read (t, h1); A pieces of stage t + 2
            {
                bf16x8_t bn[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (SC_DUET_ABL != 2 && SC_DUET_ABL != 7) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                        else asm volatile("" :: "v"(bfr[j]), "v"(af[i]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (SC_DUET_ABL != 5 && SC_DUET_ABL != 7) af[i] = *(const bf16x8_t*)(sA0 + a_base + off_h1 + i * 16 * 128);
                    if (i < 4) bn[i] = (SC_DUET_ABL != 5 && SC_DUET_ABL != 7) ? *(const bf16x8_t*)(sW0 + b_base + off_h1 + i * 16 * 128) : bfr[i];
                    if (joint) {
                        if ((i & 1) == 0 && need_me) piece_a(ta_me, g, i >> 1, k2, sA2);
                    } else if (all) {
                        if (i < 4) { if (need0) piece_a(ta0, 0, i, k2, sA2); }
                        else { if (need1) piece_a(ta1, 1, i - 4, k2, sA2); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = bn[j];
            }
            // ---- stage t + 1 must have landed before anyone reads it.  In-order retirement: everything this wave issued for stage t + 1 is older than
            //      the A pieces of this half step and -- first step of a tile -- the 2 RB stores that ended its previous epilogue.
            {
                const int newest = (joint ? 4 * (int)need_me : all ? 4 * ((int)need0 + (int)need1) : 0) + (s_me == 0 ? 2 * RB : 0);
                wait_vm(newest);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                TR(joint ? 2 : 0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                TR(joint ? 3 : 1);
            }
            // ---- second half: MFMAs of (t, h1); read (t + 1, h0); W pieces of stage t + 2 into the slot stage t leaves
            {
                bf16x8_t bn[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (SC_DUET_ABL != 2 && SC_DUET_ABL != 7) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                        else asm volatile("" :: "v"(bfr[j]), "v"(af[i]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (SC_DUET_ABL != 5 && SC_DUET_ABL != 7) af[i] = *(const bf16x8_t*)(sA1 + a_base + off_h0 + i * 16 * 128);
                    if (i < 4) bn[i] = (SC_DUET_ABL != 5 && SC_DUET_ABL != 7) ? *(const bf16x8_t*)(sW1 + b_base + off_h0 + i * 16 * 128) : bfr[i];
                    if (needw) {
                        if (joint) { if ((i & 1) == 0) piece_w(4 * g + (i >> 1), k2, sW0); }
                        else if (all) piece_w(i, k2, sW0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = bn[j];
            }
            TR(joint ? 2 : 0);
        } else if (my_ph == 2) {
            // ------------------------------------------------------------------------------------------ epilogue step (+ the whole LDS-DMA refill)
            if (SC_DUET_PRIO) __builtin_amdgcn_s_setprio(0);
            const int64_t u = u_begin + 2 * (int64_t)q_me + g;
            const int64_t mrow0 = u * 128 + srow;                          // M % 128 == 0 (host check): every unit is a full half panel
            bf16_t* cptr = (bf16_t*)p.C + mrow0 * p.ldc + ncol0;
            const bf16_t* rptr = RES ? (const bf16_t*)p.residual + mrow0 * p.ldr + ncol0 : nullptr;
            const int64_t cstep = 16 * p.ldc, rstep = 16 * p.ldr;
            // A pieces of stage t + 2 (the slot stage t - 1 left at the previous barrier)
            int n_a = 0;
            if (need0) {
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) piece_a(ta0, 0, pc, k2, sA2);
                n_a += 4;
            }
            if (need1) {
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) piece_a(ta1, 1, pc, k2, sA2);
                n_a += 4;
            }
            auto shuffled = [&](int i, uint4 (&o)[2]) {                    // row block i: bias / activation -> bf16 -> 16 rows x 64 contiguous bytes per store
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
                    o[jp] = make_uint4(__builtin_amdgcn_ds_bpermute(bperm, r0[0]), __builtin_amdgcn_ds_bpermute(bperm, r1[0]),
                                       __builtin_amdgcn_ds_bpermute(bperm, r0[1]), __builtin_amdgcn_ds_bpermute(bperm, r1[1]));
                }
            };
            uint4 oc[RB][2], res[RES ? RB : 1][2];
            auto chunk_head = [&](auto ctag) {                             // before the barrier: residual loads, conversion, shuffles
                constexpr int c = decltype(ctag)::value;
                if (RES) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int jp = 0; jp < 2; ++jp) res[rb][jp] = *(const uint4*)(rptr + (c * RB + rb) * rstep + jp * 32);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) shuffled(c * RB + rb, oc[rb]);
            };
            auto chunk_tail = [&](auto ctag) {                             // after the barrier: residual add, stores
                constexpr int c = decltype(ctag)::value;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int i = c * RB + rb;
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        uint4 o = oc[rb][jp];
                        if (RES) {
                            const uint4 rv = res[rb][jp];
                            o.x = pack2bf(lo2f(o.x) + lo2f(rv.x), hi2f(o.x) + hi2f(rv.x));
                            o.y = pack2bf(lo2f(o.y) + lo2f(rv.y), hi2f(o.y) + hi2f(rv.y));
                            o.z = pack2bf(lo2f(o.z) + lo2f(rv.z), hi2f(o.z) + hi2f(rv.z));
                            o.w = pack2bf(lo2f(o.w) + lo2f(rv.w), hi2f(o.w) + hi2f(rv.w));
                        }
                        if (SC_DUET_ABL == 3) asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
                        else *(uint4*)(cptr + i * cstep + jp * 32) = o;
                    }
                }
            };
            const int cidx = s_me - nk;
            if (SC_DUET_ABL == 8) {          // timing probe (garbage results): the accumulators stay live, no conversion, no stores
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
            } else
            switch (cidx) {
                case 0: chunk_head(std::integral_constant<int, 0>{}); break;
                case 1: chunk_head(std::integral_constant<int, 1>{}); break;
                case 2: chunk_head(std::integral_constant<int, 2>{}); break;
                case 3: chunk_head(std::integral_constant<int, 3>{}); break;
                case 4: if (E > 4) chunk_head(std::integral_constant<int, (E > 4 ? 4 : 0)>{}); break;
                case 5: if (E > 4) chunk_head(std::integral_constant<int, (E > 4 ? 5 : 0)>{}); break;
                case 6: if (E > 4) chunk_head(std::integral_constant<int, (E > 4 ? 6 : 0)>{}); break;
                default: if (E > 4) chunk_head(std::integral_constant<int, (E > 4 ? 7 : 0)>{}); break;
            }
            // This wave's pieces of stage t + 1 (W issued right after the previous barrier; A before it; or, chunk 0, issued in its last compute step)
            // are older than: the previous chunk's 2 RB stores (chunks > 0), the A pieces above and this chunk's residual loads.
            wait_vm(SC_DUET_ABL == 8 ? n_a : (cidx > 0 ? 2 * RB : 0) + n_a + (RES ? 2 * RB : 0));
            TR(4);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TR(5);
            if (needw) {
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) piece_w(pc, k2, sW0);
            }
            if (SC_DUET_ABL != 8)
            switch (cidx) {
                case 0: chunk_tail(std::integral_constant<int, 0>{}); break;
                case 1: chunk_tail(std::integral_constant<int, 1>{}); break;
                case 2: chunk_tail(std::integral_constant<int, 2>{}); break;
                case 3: chunk_tail(std::integral_constant<int, 3>{}); break;
                case 4: if (E > 4) chunk_tail(std::integral_constant<int, (E > 4 ? 4 : 0)>{}); break;
                case 5: if (E > 4) chunk_tail(std::integral_constant<int, (E > 4 ? 5 : 0)>{}); break;
                case 6: if (E > 4) chunk_tail(std::integral_constant<int, (E > 4 ? 6 : 0)>{}); break;
                default: if (E > 4) chunk_tail(std::integral_constant<int, (E > 4 ? 7 : 0)>{}); break;
            }
            TR(4);
        } else {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TR(6);
        }
        // ---- advance
        { char* x = sA0; sA0 = sA1; sA1 = sA2; sA2 = x; x = sW0; sW0 = sW1; sW1 = x; }
        if (++s0 == P) { s0 = 0; ++q0; }
        if (++s1 == P) { s1 = 0; ++q1; }
        if (++kc2 == nk) kc2 = 0;
    }
    if (SC_PROBES && p.trace && lane == 0) {
        unsigned long long* tr = p.trace + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) tr[i] = trc[i];
    }
}

template <int ACT, bool RES, int E>
int launch_one(const Gemm8pParams& p, int grid, hipStream_t s) {
    constexpr int lds = 5 * HALF_SLOT;   // 160 KiB
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_duet_kernel<ACT, RES, E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_duet_kernel<ACT, RES, E>), dim3(grid), dim3(512), lds, s, p);
    SC_CHECK_LAUNCH();
    return 0;
}

template <int E>
int launch_var(const Gemm8pParams& p, int grid, hipStream_t s) {
    const bool res = p.residual != nullptr;
    switch (p.act) {
        case SC_ACT_GELU: return res ? launch_one<SC_ACT_GELU, true, E>(p, grid, s) : launch_one<SC_ACT_GELU, false, E>(p, grid, s);
        case SC_ACT_QUICKGELU: return res ? launch_one<SC_ACT_QUICKGELU, true, E>(p, grid, s) : launch_one<SC_ACT_QUICKGELU, false, E>(p, grid, s);
        default: return res ? launch_one<SC_ACT_NONE, true, E>(p, grid, s) : launch_one<SC_ACT_NONE, false, E>(p, grid, s);
    }
}

}  // namespace

int sc_gemm_duet_try(const Gemm8pParams& pin, hipStream_t s) {
    Gemm8pParams p = pin;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, dev); n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const int grid = n_cu & ~7;                       // blocks b, b + 8, .. share an XCD; one block per CU (160 KiB of LDS each)
    if (grid < 8) return 1;
    if (p.out_f32 || p.N % 256 || p.K % 64 || p.M < 256 || p.M % 128) return 1;      // full 128-row half panels only: the epilogue stores unpredicated (counted vmcnt)
    p.tn = p.N / 256; p.nk = p.K / 64;
    if (p.tn > grid || p.tn > 32 || p.nk < 8) return 1;
    if (p.ldc % 8 || (p.residual && p.ldr % 8) || p.lda % 8 || p.ldw % 8) return 1;
    if (p.lda > (1 << 22) || p.ldw > (1 << 22)) return 1;          // per-lane offsets inside a half panel are 32-bit
    p.rows = grid / p.tn;
    p.units = (p.M + 127) / 128;
    if (p.units < (int64_t)p.rows * 6) return 1;                    // short launches: the start / end skew of the two groups is not amortised
    if (p.esteps != 8) p.esteps = 4;
    if (p.esteps == 8) return launch_var<8>(p, grid, s);
    return launch_var<4>(p, grid, s);
}
