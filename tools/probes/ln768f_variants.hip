// layernorm768f_kernel as it shipped until the reproducibility fix (hipcc's SLP vectoriser packs the two rows a wave processes into v_pk_*_f32 code), in variants,
// for tools/vit_race_probe.py (variant ln_v<k>): which ingredient makes the packed form's result timing-dependent beside the speech front end?
//   0  the original source (wave sums by __shfl_xor = ds_bpermute_b32, rows side by side: packed code)
//   1  the same with the wave sums by DPP (row-local quad_perm / mirror steps, row_bcast15 / 31, v_readlane): no LDS-crossbar instruction in the kernel, still packed
//   2  the original with an empty asm barrier between the two rows' statistics (the rows cannot be packed: what -fno-slp-vectorize does, per kernel)
//   3  the original with s_waitcnt lgkmcnt(0) + vmcnt(0) in front of every cross-lane step (packed, bpermute, but nothing in flight around them)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float a, float b) { bf16x2_t v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(uint32_t, v); }

template <int V> __device__ __forceinline__ float wsum(float v) {
    if constexpr (V == 1) {
#define SC_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xf, false))
        v += SC_DPP(v, 0xB1, 0xf);       // quad_perm [1,0,3,2]
        v += SC_DPP(v, 0x4E, 0xf);       // quad_perm [2,3,0,1]
        v += SC_DPP(v, 0x141, 0xf);      // row_half_mirror
        v += SC_DPP(v, 0x140, 0xf);      // row_mirror: every lane holds its row-of-16's sum
        v += SC_DPP(v, 0x142, 0xa);      // row_bcast15 into rows 1 and 3
        v += SC_DPP(v, 0x143, 0xc);      // row_bcast31 into rows 2 and 3: lane 63 holds the total
#undef SC_DPP
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            if constexpr (V == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(v) :: "memory");
            v += __shfl_xor(v, o, 64);
        }
        return v;
    }
}

template <int V>
__global__ __launch_bounds__(256) void ln768f_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     unsigned short* __restrict__ out, int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;
    f32x4_t v[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float* xr = x + (row0 + (two ? r : 0)) * 768 + lane * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) v[r][q] = *(const f32x4_t*)(xr + q * 256);
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) s += (v[r][q][0] + v[r][q][1]) + (v[r][q][2] + v[r][q][3]);
        mean[r] = wsum<V>(s) * (1.0f / 768.0f);
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = v[r][q][i] - mean[r]; qq += d * d; }
        rstd[r] = rsqrtf(wsum<V>(qq) * (1.0f / 768.0f) + eps);
        if constexpr (V == 2) asm volatile("" : "+v"(mean[r]), "+v"(rstd[r]) :: "memory");
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int col = q * 256 + lane * 4;
        const f32x4_t g4 = *(const f32x4_t*)(gamma + col), b4 = *(const f32x4_t*)(beta + col);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r && !two) continue;
            const f32x4_t o = (v[r][q] - mean[r]) * rstd[r] * g4 + b4;
            uint2 u;
            u.x = pack2bf(o[0], o[1]); u.y = pack2bf(o[2], o[3]);
            *(uint2*)(out + (row0 + r) * 768 + col) = u;
        }
    }
}

extern "C" int ln768f_variant(int variant, const void* x, const void* gamma, const void* beta, void* out, int64_t rows, float eps, void* stream) {
    const dim3 grid((unsigned)((rows + 7) / 8)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define L(V_) hipLaunchKernelGGL(ln768f_kernel<V_>, grid, block, 0, s, (const float*)x, (const float*)gamma, (const float*)beta, (unsigned short*)out, rows, eps)
    if (variant == 1) L(1); else if (variant == 2) L(2); else if (variant == 3) L(3); else L(0);
#undef L
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
