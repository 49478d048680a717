// K-LOOP PROTOTYPE of the "one wave per SIMD" GEMM structure VERDICT r5 asked for (item 1c), measured BEFORE building the full kernel:
// 256 x 256 x 64 tiles, 4 waves per block (one per SIMD, 512 registers each), 128 x 128 wave tile = 256 accumulator registers,
// operands by LDS-DMA (global_load_lds_dwordx4) into two 64 KiB stages, ONE barrier per k-step, fragment reads software-pipelined half a k-step
// (MF = 16) / a quarter k-step (MF = 32) ahead of their MFMAs.  Template MF: 16 = v_mfma_f32_16x16x32_bf16 (8 x 8 tiles), 32 = v_mfma_f32_32x32x16_bf16
// (4 x 4 tiles).  The epilogue is the simplest correct one (8-byte stores straight from the MFMA layout): what is measured is the k-loop, by s_memtime stamps
// around it (cycles per k-step; 2 048 = MFMA-bound) -- and the whole-launch TF/s, next to gemm8p's numbers from tools/gemm_bench.py on the same box.
// Results are CHECKED (sampled entries against a host fp64 dot product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm4w_probe_bin gemm4w_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <type_traits>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef DMA_H1          // LDS-DMA pieces (of 16 per wave and k-step) issued in the SECOND half of the k-step, right behind the barrier; the rest in the first half of the next
#define DMA_H1 16
#endif
#ifndef ABL             // 1 no fragment reads, 2 no LDS-DMA, 3 no MFMA, 4 no barrier (timing ablations: garbage results)
#define ABL 0
#endif

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned pack2bf(float a, float b) { bf16x2_t v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }

constexpr int STAGE = 65536, WOFF = 32768;

template <int MF, int NW>
__global__ __launch_bounds__(NW * 64) void gemm4w(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NI = MF == 16 ? 8 : 4;          // accumulator tiles per wave: NI (A row blocks) x NJ (W row blocks)
    constexpr int NJ = NW == 4 ? NI : NI / 2;     // NW = 4: wave tile 128 x 128; NW = 8 (two waves per SIMD, free-running): 128 x 64
    constexpr int WN = NW == 4 ? 128 : 64;        // W rows (output columns) per wave
    constexpr int PP = 32 / NW;                   // DMA pieces of A (and of W) per wave and k-step
    constexpr int NS = MF == 16 ? 2 : 4;          // sub-steps of a 64-deep k-step (k-halves of 32 / k-quarters of 16)
    constexpr int RB = MF;                        // rows per fragment
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;
    const int nk = K / 64;
    const int tiles_m = M / 256, tiles_n = N / 256, nwg = tiles_m * tiles_n;
    // XCD-aware persistent tile list: block b lives on XCD b % 8 and takes every nb_xcd-th tile of that XCD's contiguous, M-panel-major chunk
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nb_xcd = (G - xcd + 7) >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, cnt = q8 + (xcd < r8 ? 1 : 0);

    // fragment read offsets inside a stage: sub-step s, my lane; + i * RB * 128 per row block
    int a_off[NS], w_off[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int row, chunk, swz;
        if (MF == 16) { row = lane & 15; chunk = s * 4 + (lane >> 4); swz = row & 7; }
        else          { row = lane & 31; chunk = s * 2 + (lane >> 5); swz = (row >> 1) & 7; }
        a_off[s] = (wm * 128 + row) * 128 + ((chunk ^ swz) << 4);
        w_off[s] = WOFF + (wn * WN + row) * 128 + ((chunk ^ swz) << 4);
    }
    // LDS-DMA: piece = 8 rows x 128 B; wave w stages rows w * 64 .. + 64 of the A tile and of the W tile (8 + 8 pieces per k-step)
    const int lr = lane >> 3, lc = lane & 7;
    unsigned dma_a[2], dma_w[2];                  // per-lane byte offset of piece parity 0 / 1 (the swizzle of MF = 32 depends on it)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int swz = MF == 16 ? lr : ((4 * par + (lane >> 4)) & 7);
        dma_a[par] = (unsigned)(lr * K + ((lc ^ swz) << 3)) * 2u;
        dma_w[par] = dma_a[par];
    }
    auto stage_piece = [&](const bf16_t* ta, const bf16_t* tw, int q, int kt, char* st) {      // q: 0..7 A pieces, 8..15 W pieces of this wave
        if (ABL == 2) return;
        const int p = wave * PP + (q % PP);
        const bf16_t* base = (q < PP ? ta : tw) + (size_t)p * 8 * K + (size_t)kt * 64;
        glds16((const char*)base + (q < PP ? dma_a[q & 1] : dma_w[q & 1]), st + (q < PP ? 0 : WOFF) + p * 1024);
    };

    typedef typename std::conditional<MF == 16, f32x4_t, f32x16_t>::type acc_t;
    acc_t acc[NI][NJ];
    bf16x8_t fa[2][NI], fw[2][NJ];
    auto read_frags = [&](const char* st, int s, int set) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (ABL == 1) { asm volatile("" : "+v"(fa[set][i])); if (i < NJ) asm volatile("" : "+v"(fw[set][i])); continue; }
            fa[set][i] = *(const bf16x8_t*)(st + a_off[s] + i * RB * 128);
            if (i < NJ) fw[set][i] = *(const bf16x8_t*)(st + w_off[s] + i * RB * 128);
        }
    };
    auto mfma = [&](int i, int j, int set) {
        if (ABL == 3) { asm volatile("" : "+v"(acc[i][j]) : "v"(fw[set][j]), "v"(fa[set][i])); return; }
        if constexpr (MF == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[set][j], fa[set][i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };

    unsigned long long cyc_loop = 0; int steps = 0;
    for (int idx = slot; idx < cnt; idx += nb_xcd) {
        const int v = begin + idx, tm = v / tiles_n, tn = v - tm * tiles_n;
        const bf16_t* ta = A + (size_t)tm * 256 * K;
        const bf16_t* tw = W + (size_t)tn * 256 * K;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < (MF == 16 ? 4 : 16); ++e) acc[i][j][e] = 0.f;
        // prologue: k-steps 0 and 1 requested, k-step 0 landed, first fragments read
        __builtin_amdgcn_s_barrier();      // (every wave has left the previous tile's k-loop: both stages are free)
#pragma unroll
        for (int q = 0; q < 2 * PP; ++q) stage_piece(ta, tw, q, 0, smem);
#pragma unroll
        for (int q = 0; q < 2 * PP; ++q) stage_piece(ta, tw, q, 1, smem + STAGE);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PP) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(smem, 0, 0);
        const unsigned long long c0 = __builtin_readcyclecounter();
        char* sx = smem; char* sy = smem + STAGE;
        for (int kt = 0; kt < nk; ++kt) {
            const int kn2 = kt + 2 < nk ? kt + 2 : nk - 1, kn1 = kt + 1 < nk ? kt + 1 : nk - 1;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int set = s & 1;
                const bool last = s == NS - 1;
                if (last) {
                    // every read of stage sx has been issued (sub-step NS - 2); they are complete once the fragments of this sub-step are waited for (compiler's lgkmcnt).
                    // k-step kt + 1 (requested one k-step ago) must have landed in sy before anybody reads it.
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if (ABL != 4) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                // MFMAs of sub-step s interleaved with: the fragment reads of the next sub-step, the DMA pieces of this part of the k-step
                constexpr int NM = NI * NJ;                       // MFMAs per sub-step
                constexpr int GRP = MF == 16 ? 4 : 1;             // MFMAs per filler slot
                constexpr int SLOTS = NM / GRP;                   // 16 (NW 4) / 8 (NW 8) slots per sub-step
                constexpr int NR = NI + NJ;                       // fragment reads of the next sub-step
                constexpr int RPS = (NR + SLOTS - 1) / SLOTS;     // reads per slot
                constexpr int NP = 2 * PP;                        // DMA pieces per wave and k-step
                constexpr int H1 = DMA_H1 < NP ? DMA_H1 : NP;     // pieces issued in the last sub-step (behind the barrier); the rest in sub-step 0 of the next k-step
                constexpr int DPS = (H1 + SLOTS - 1) / SLOTS, DPS0 = (NP - H1 + SLOTS - 1) / SLOTS;
                const char* rd_st = last ? sy : sx;
                const int rd_s = last ? 0 : s + 1;
#pragma unroll
                for (int slot_i = 0; slot_i < SLOTS; ++slot_i) {
#pragma unroll
                    for (int rr = 0; rr < RPS; ++rr) {      // (the last k-step of a tile reads junk of the other stage: the persistent form reads the next tile's first fragments here)
                        const int r = slot_i * RPS + rr;
                        if (r < NR) {
                            if (ABL == 1) { if (r < NI) asm volatile("" : "+v"(fa[set ^ 1][r < NI ? r : 0])); else asm volatile("" : "+v"(fw[set ^ 1][r >= NI ? r - NI : 0])); }
                            else if (r < NI) fa[set ^ 1][r < NI ? r : 0] = *(const bf16x8_t*)(rd_st + a_off[rd_s] + r * RB * 128);
                            else fw[set ^ 1][r >= NI ? r - NI : 0] = *(const bf16x8_t*)(rd_st + w_off[rd_s] + (r - NI) * RB * 128);
                        }
                    }
                    // DMA: k-step kt + 2 into sx (tile tail: re-stages the last k-step, unused -- the persistent form requests the next tile here)
#pragma unroll
                    for (int dd = 0; dd < DPS; ++dd) { const int q = slot_i * DPS + dd; if (last && q < H1) stage_piece(ta, tw, q, kn2, sx); }
#pragma unroll
                    for (int dd = 0; dd < DPS0; ++dd) { const int q = H1 + slot_i * DPS0 + dd; if (s == 0 && q < NP) stage_piece(ta, tw, q, kn1, sy); }
#pragma unroll
                    for (int g = 0; g < GRP; ++g) { const int m = slot_i * GRP + g; mfma(m / NJ, m % NJ, set); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            { char* x = sx; sx = sy; sy = x; }
        }
        cyc_loop += __builtin_readcyclecounter() - c0; steps += nk;
        // epilogue (simple): 8-byte stores from the MFMA layout, D[n][m] orientation (W is the first MFMA operand)
        bf16_t* ct = C + (size_t)(tm * 256 + wm * 128) * N + tn * 256 + wn * WN;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (MF == 16) {
                    const int m = i * 16 + (lane & 15), n = j * 16 + (lane >> 4) * 4;
                    uint2 o = make_uint2(pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3]));
                    *(uint2*)(ct + (size_t)m * N + n) = o;
                } else {
                    const int m = i * 32 + (lane & 31);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = j * 32 + 8 * g + 4 * (lane >> 5);
                        uint2 o = make_uint2(pack2bf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                        *(uint2*)(ct + (size_t)m * N + n) = o;
                    }
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (trace && lane == 0) { trace[(blockIdx.x * NW + wave) * 2] = cyc_loop; trace[(blockIdx.x * NW + wave) * 2 + 1] = steps; }
}

static unsigned short f2bf_host(float f) { unsigned b; memcpy(&b, &f, 4); return (unsigned short)((b + 0x7fff + ((b >> 16) & 1)) >> 16); }
static float bf2f_host(unsigned short h) { unsigned b = (unsigned)h << 16; float f; memcpy(&f, &b, 4); return f; }

template <int MF, int NW>
static void run(int M, int N, int K, const bf16_t* dA, const bf16_t* dW, bf16_t* dC, unsigned long long* dtrace, const std::vector<unsigned short>& hA, const std::vector<unsigned short>& hW, double seconds) {
    const int lds = 2 * STAGE;
    HIPCHECK(hipFuncSetAttribute((const void*)gemm4w<MF, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = 256;
    HIPCHECK(hipMemset(dC, 0, (size_t)M * N * 2));
    hipLaunchKernelGGL((gemm4w<MF, NW>), dim3(grid), dim3(NW * 64), lds, 0, dA, dW, dC, M, N, K, dtrace);
    HIPCHECK(hipDeviceSynchronize());
    // check sampled entries
    std::vector<unsigned short> hC((size_t)M * N);
    HIPCHECK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double max_err = 0; int bad = 0; unsigned long long s = 12345;
    for (int t = 0; t < 20000; ++t) {
        s = s * 6364136223846793005ull + 1442695040888963407ull; const int m = (int)((s >> 33) % M);
        s = s * 6364136223846793005ull + 1442695040888963407ull; const int n = (int)((s >> 33) % N);
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f_host(hA[(size_t)m * K + k]) * bf2f_host(hW[(size_t)n * K + k]);
        const double got = bf2f_host(hC[(size_t)m * N + n]), err = fabs(got - ref);
        if (err > 0.02 + 0.01 * fabs(ref)) ++bad;
        if (err > max_err) max_err = err;
    }
    hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    // sustained: launches for `seconds`, timing the last ones
    int n = 0; float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        n = 0; HIPCHECK(hipEventRecord(e0));
        const double budget = rep == 0 ? seconds : seconds * 0.5;
        const double one = 2.0 * M * N * K / 1.0e15;      // ~ seconds per launch at 1 PF/s
        const int launches = (int)(budget / one) + 1;
        for (int i = 0; i < launches; ++i) { hipLaunchKernelGGL((gemm4w<MF, NW>), dim3(grid), dim3(NW * 64), lds, 0, dA, dW, dC, M, N, K, dtrace); ++n; }
        HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> tr(grid * NW * 2);
    HIPCHECK(hipMemcpy(tr.data(), dtrace, tr.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, st = 0; for (int b = 0; b < grid * NW; ++b) { cyc += tr[b * 2]; st += tr[b * 2 + 1]; }
    printf("NW %d MF %2d  DMA_H1 %2d ABL %d | M %7d N %5d K %5d | %7.1f TF/s | k-step %6.0f cycles (2048 = MFMA-bound) | check: %d bad of 20000, max abs err %.4f\n",
           NW, MF, DMA_H1, ABL, M, N, K, 2.0 * M * N * K * n / (ms * 1e-3) / 1e12, st > 0 ? cyc / st : 0.0, bad, max_err);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
    struct Shape { int M, N, K; } shapes[] = {{128000, 2304, 768}, {128000, 768, 3072}, {128000, 768, 768}, {8192, 8192, 8192}};
    const int mask = argc > 2 ? atoi(argv[2]) : 15;      // bit i: run shape i
    const int mfs = argc > 3 ? atoi(argv[3]) : 15;         // bit 0: 4 waves MF 16, bit 1: 4 waves MF 32, bit 2: 8 waves MF 16, bit 3: 8 waves MF 32
    int si = -1;
    for (const Shape& sh : shapes) {
        if (!((mask >> ++si) & 1)) continue;
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        unsigned long long s = 0x9E3779B97F4A7C15ull;
        auto u = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
        for (auto& x : hA) x = f2bf_host((float)((u() + u() + u() + u() - 2.0) * 0.866));
        const float ws = 1.0f / sqrtf((float)K);
        for (auto& x : hW) x = f2bf_host((float)((u() + u() + u() + u() - 2.0) * 1.732 * ws));
        bf16_t *dA, *dW, *dC; unsigned long long* dtrace;
        HIPCHECK(hipMalloc(&dA, hA.size() * 2)); HIPCHECK(hipMalloc(&dW, hW.size() * 2)); HIPCHECK(hipMalloc(&dC, (size_t)M * N * 2)); HIPCHECK(hipMalloc(&dtrace, 256 * 8 * 16));
        HIPCHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        if (mfs & 1) run<16, 4>(M, N, K, dA, dW, dC, dtrace, hA, hW, seconds);
        if (mfs & 2) run<32, 4>(M, N, K, dA, dW, dC, dtrace, hA, hW, seconds);
        if (mfs & 4) run<16, 8>(M, N, K, dA, dW, dC, dtrace, hA, hW, seconds);
        if (mfs & 8) run<32, 8>(M, N, K, dA, dW, dC, dtrace, hA, hW, seconds);
        HIPCHECK(hipFree(dA)); HIPCHECK(hipFree(dW)); HIPCHECK(hipFree(dC)); HIPCHECK(hipFree(dtrace));
    }
    return 0;
}
