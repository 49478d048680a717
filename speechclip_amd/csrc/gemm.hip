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
#include "common.h"
#include "../../include/speechclip_hip.h"

namespace {

struct GemmParams {
    const bf16_t* A; int64_t lda; int64_t strideA;
    const bf16_t* W; int64_t ldw; int64_t strideW; int w_mod;
    void* C; int64_t ldc; int64_t strideC;
    const float* bias;
    const void* residual; int64_t ldr;
    int64_t M; int N; int K;
    int tiles_m; int tiles_n;
    int act; int out_f32;
};

constexpr int BK = 64;  // 128 bytes of bf16 per tile row = 8 chunks of 16 B

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage a ROWS x 64 bf16 tile: LDS image is [row][8 chunks of 16 B], chunk position p of row r holds
// global k-chunk (p ^ (r & 7)).  256 threads => 32 rows per pass.
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

template <int BM, int BN>
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

    const bf16_t* A = p.A + (int64_t)z * p.strideA;
    const bf16_t* W = p.W + (int64_t)(z % p.w_mod) * p.strideW;
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: lane holds C[m = .. + (lane & 15)][n = .. + 4*(lane>>4) + r], r = 0..3
    const float* bias = p.bias ? p.bias + (int64_t)(z % p.w_mod) * p.N : nullptr;
    char* Cb = (char*)p.C + (int64_t)z * p.strideC * (p.out_f32 ? 4 : 2);
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
                for (int r = 0; r < 4; ++r) v4[r] = gelu_erf(v4[r]);
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
                    v4[0] += lo2f(rr.x); v4[1] += hi2f(rr.x); v4[2] += lo2f(rr.y); v4[3] += hi2f(rr.y);
                }
                uint2 o;
                o.x = pack2bf(v4[0], v4[1]);
                o.y = pack2bf(v4[2], v4[3]);
                *(uint2*)((bf16_t*)Cb + m * p.ldc + n) = o;
            }
        }
    }
}

template <int BM, int BN>
int launch(const GemmParams& p, int batch, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    dim3 grid(p.tiles_m * p.tiles_n, batch);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN>), grid, dim3(256), lds, s, p);
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
    if (p.N <= 64) {
        p.tiles_m = (int)((p.M + 127) / 128); p.tiles_n = (p.N + 63) / 64;
        return launch<128, 64>(p, batch, s);
    }
    p.tiles_m = (int)((p.M + 127) / 128); p.tiles_n = (p.N + 127) / 128;
    return launch<128, 128>(p, batch, s);
}

}  // namespace

extern "C" int sc_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                            const float* bias, const void* residual, int64_t ldr, int64_t M, int N, int K,
                            int flags, void* stream) {
    GemmParams p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.strideA = 0;
    p.W = (const bf16_t*)W; p.ldw = ldw; p.strideW = 0; p.w_mod = 1;
    p.C = C; p.ldc = ldc; p.strideC = 0;
    p.bias = bias; p.residual = residual; p.ldr = ldr;
    p.M = M; p.N = N; p.K = K;
    p.act = flags & SC_GEMM_ACT_MASK; p.out_f32 = (flags & SC_GEMM_OUT_F32) ? 1 : 0;
    return gemm_dispatch(p, 1, (hipStream_t)stream);
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
