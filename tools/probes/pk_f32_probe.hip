// Packed-fp32 arithmetic beside other kernels (round 6, EXPERIMENTS.md R6-14): every lane evaluates v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (incl. the op_sel
// cross-half form) and the same operations as single v_fma_f32 / v_mul_f32 / v_add_f32 on identical pseudo-random operands and counts results that differ in any bit,
// separately for the low and the high half.  Launched in a loop on a side stream while the main stream runs a load (tools/pk_f32_probe.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd(uint32_t s) { return __uint_as_float(0x3f800000u | (h32(s) >> 9)) - 1.5f; }      // [-0.5, 0.5)

extern "C" __global__ __launch_bounds__(256) void pk_probe_kernel(unsigned long long* counts, int iters, uint32_t seed) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    unsigned bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x2 acc = {0.f, 0.f};
    float acc0 = 0.f, acc1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t s = seed + gid * 977u + it * 4u;
        const f32x2 a = {rnd(s), rnd(s + 1)}, b = {rnd(s + 2) * 3.f, rnd(s + 3) * 3.f}, c = {rnd(s + 0x9999u), rnd(s + 0x7777u)};
        f32x2 pf, pm, pa, px;
        float f0, f1, m0, m1, a0, a1, x0, x1;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(a), "v"(c));
        asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(px) : "v"(b));      // (lo + hi, hi + lo)
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a[1]), "v"(b[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(a[0]), "v"(c[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(a[1]), "v"(c[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(x0) : "v"(b[0]), "v"(b[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(x1) : "v"(b[1]), "v"(b[0]));
        bad[0] += __float_as_uint(pf[0]) != __float_as_uint(f0); bad[1] += __float_as_uint(pf[1]) != __float_as_uint(f1);
        bad[2] += __float_as_uint(pm[0]) != __float_as_uint(m0); bad[3] += __float_as_uint(pm[1]) != __float_as_uint(m1);
        bad[4] += __float_as_uint(pa[0]) != __float_as_uint(a0); bad[5] += __float_as_uint(pa[1]) != __float_as_uint(a1);
        bad[6] += __float_as_uint(px[0]) != __float_as_uint(x0); bad[7] += __float_as_uint(px[1]) != __float_as_uint(x1);
        // a dependent chain as well (accumulators), packed against single: compared once at the end
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a[1]), "v"(b[1]));
    }
    for (int k = 0; k < 8; ++k)
        if (bad[k]) atomicAdd(&counts[k], (unsigned long long)bad[k]);
    if (__float_as_uint(acc[0]) != __float_as_uint(acc0)) atomicAdd(&counts[8], 1ull);
    if (__float_as_uint(acc[1]) != __float_as_uint(acc1)) atomicAdd(&counts[9], 1ull);
}

extern "C" int pk_probe_launch(void* counts, int blocks, int iters, uint32_t seed, void* stream) {
    hipLaunchKernelGGL(pk_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)counts, iters, seed);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
