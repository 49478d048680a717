// Microbenchmark: do the matrix pipe and the vector ALU of ONE SIMD run concurrently when the work comes from two different waves?
// One 512-thread block per CU (waves w and w + 4 share a SIMD).  Per wave and iteration: M = 16 independent-accumulator
// v_mfma_f32_32x32x16_bf16 (512 cycles of matrix pipe) or V = 32 v_exp_f32 + 96 v_fma_f32 (the attention kernel's per-tile mix).
//   mode 0: all 8 waves matrix      mode 1: all 8 waves vector      mode 2: waves 0-3 matrix, 4-7 vector (one of each per SIMD)
//   mode 3: every wave alternates a matrix and a vector phase, the two waves of a SIMD in opposite phases
//   mode 4: as 3, both waves in the SAME phase          mode 5: one wave per SIMD (waves 4-7 idle), alternating phases
//   mode 6 / 7: one wave per SIMD, vector only / matrix only
// Build: hipcc --offload-arch=gfx950 -O3 -o simd_overlap_probe_bin simd_overlap_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void matrix_phase(f32x16_t (&acc)[4], bf16x8_t a, bf16x8_t b) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
}
__device__ __forceinline__ void vector_phase(float (&v)[32], float c) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        float t = __builtin_fmaf(v[i], c, -1.0f);
        t = __builtin_amdgcn_exp2f(t);
        t = __builtin_fmaf(t, c, 0.25f);
        v[i] = __builtin_fmaf(t, 0.5f, c);
    }
}

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, float c) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16_t acc[4];
    float v[32];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 32; ++i) v[i] = 0.001f * (lane + i);
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * e); }
    const bool hi = wave >= 4;
    if ((mode == 5 || mode == 6 || mode == 7) && hi) return;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0 || mode == 7 || (mode == 2 && !hi)) matrix_phase(acc, a, b);
        else if (mode == 1 || mode == 6 || (mode == 2 && hi)) vector_phase(v, c);
        else if (mode == 3) { if (hi) { vector_phase(v, c); matrix_phase(acc, a, b); } else { matrix_phase(acc, a, b); vector_phase(v, c); } }
        else { matrix_phase(acc, a, b); vector_phase(v, c); }
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    for (int i = 0; i < 32; ++i) r += v[i];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int grid : {256, 16})     // 16 blocks: 16 CUs busy, far from the socket's power cap -- separates issue limits from the power controller
    for (int mode = 0; mode < 8; ++mode) {
        k<<<grid, 512>>>(out, 1000, mode, 0.37f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 512>>>(out, iters, mode, 0.37f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %3d mode %d: %.3f ms  = %.0f ns per iteration\n", grid, mode, ms, ms * 1e6 / iters);
    }
    return 0;
}
