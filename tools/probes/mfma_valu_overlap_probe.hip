// Does a wave streaming back-to-back MFMAs (a ping-pong MFMA cluster) share its SIMD gracefully with a wave doing packed-half VALU work (a GELU epilogue)?
// One 512-thread block per CU; waves w and w + 4 share a SIMD.  mode bit 0: waves 0-3 run MFMA clusters; bit 1: waves 4-7 run VALU chains.
// Prints cycles per MFMA cluster (16 x v_mfma_f32_16x16x32_bf16, independent accumulators) and per 64 VALU instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap_probe.hip -o /tmp/mvprobe && /tmp/mvprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(int mode, int iters, unsigned long long* out, float* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_m = wave < 4;
    unsigned long long t0 = 0, t1 = 0;
    if (is_m) {
        if (!(mode & 1)) return;
        f32x4_t acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        bf16x8_t a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i - 3); }
        __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        half2_t x[8], g[8];
        for (int i = 0; i < 8; ++i) { x[i] = (half2_t){(_Float16)(0.01f * (threadIdx.x + i)), (_Float16)(0.02f * i)}; g[i] = x[i]; }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = g[i] * x[i] + (_Float16)0.25f;       // 64 v_pk_fma_f16 per iteration, 8 independent chains
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += (float)g[i][0] + (float)g[i][1];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    unsigned long long h[256 * 8];
    const int iters = 2000;
    for (int grid : {16, 256}) {
        for (int mode = 1; mode <= 3; ++mode) {
            hipMemset(out, 0, 256 * 8 * 8);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, mode, iters, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, out, 256 * 8 * 8, hipMemcpyDeviceToHost);
            double m = 0, v = 0; int nm = 0, nv = 0;
            for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) { if (!h[b * 8 + w]) continue; if (w < 4) { m += h[b * 8 + w]; ++nm; } else { v += h[b * 8 + w]; ++nv; } }
            printf("grid %3d mode %d (%s): cycles per 16-MFMA cluster %7.1f   cycles per 64 v_pk_fma_f16 %7.1f\n", grid, mode,
                   mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", nm ? m / nm / iters : 0.0, nv ? v / nv / iters : 0.0);
        }
    }
    return 0;
}
