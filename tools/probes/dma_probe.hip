// Microbenchmark: achievable L2 -> LDS LDS-DMA bandwidth per CU for the GEMM's staging pattern (512 threads, 64 KiB per "stage").
// mode 0: every block streams the SAME 64 KiB x NSTG region (W-like, hot in L2)
// mode 1: blocks of one XCD share a region, XCDs differ
// mode 2: every block has its own region (fits L2/MALL depending on size)
// depth: stages kept in flight (1 = wait for each stage; 2 = issue next before waiting)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// BUF: the same traffic through buffer_load_dwordx4 ... offen lds (SGPR resource + 32-bit lane offset + SGPR stage offset) instead of
// global_load_lds_dwordx4 with a 64-bit per-lane address.
template <bool BUF>
__global__ __launch_bounds__(512) void k(const char* buf, size_t region, int mode, int iters, int nstg, int depth, int rowbytes, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    size_t base = mode == 0 ? 0 : (mode == 1 ? (size_t)(blockIdx.x & 7) * region : (size_t)blockIdx.x * region);
    const char* p = buf + base;
    // rows of `rowbytes` bytes (128 or 64) at a stride of 1536 B (K = 768 bf16), 16 B per lane
    const int per_row = rowbytes / 16;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const int stg = it % nstg;
        const char* sp = p + (size_t)stg * rowbytes;      // k-offset within the rows
        char* slot = smem + (it & 1) * 65536;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 512 + tid;
            const int r = c / per_row, q = c % per_row;
            if (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(slot + (i * 512 + wave * 64) * 16), 16,
                                                              r * 1536 + q * 16, stg * rowbytes, 0, 0);
            else glds16(sp + (size_t)r * 1536 + q * 16, slot + (i * 512 + wave * 64) * 16);
        }
        if (depth == 1 || it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && smem[5] == 77) *sink = 1;
}
int main(int argc, char** argv) {
    int iters = 2000;
    size_t region = 8 << 20;
    char* buf; hipMalloc(&buf, region * 256 + (1 << 20)); hipMemset(buf, 1, region * 256 + (1 << 20));
    int* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bufv = 0; bufv < 2; ++bufv)
    for (int rowbytes : {128, 64})
    for (int mode = 0; mode < 3; ++mode)
    for (int depth = 1; depth <= 2; ++depth)
    for (int nstg : {12, 4096}) {
        if ((size_t)nstg * rowbytes > 1536 && nstg == 12 && rowbytes == 128) {}
        int ns = nstg == 12 ? 1536 / rowbytes : nstg;      // 12 x 128 B = one K=768 row; 4096 -> walks 4096 rows blocks (bigger footprint)
        // footprint per block-region: rows 4096 x 1536 B = 6 MB (mode 2: distinct per block -> 1.5 GB total, beyond L2/MALL)
        auto kern = bufv ? k<true> : k<false>;
        kern<<<256, 512, 131072>>>(buf, region, mode, 50, ns > 12 ? 12 : ns, depth, rowbytes, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, 512, 131072>>>(buf, region, mode, iters, ns > 12 ? 12 : ns, depth, rowbytes, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = 256.0 * iters * 65536.0 * (rowbytes / 128.0);   // rowbytes=64 moves half the bytes per 8 instr
        printf("%s row=%3dB mode=%d depth=%d: %.3f ms  %.2f TB/s aggregate  (%.1f B/clk/CU @2.2GHz)\n", bufv ? "buffer" : "global", rowbytes, mode, depth, ms,
               bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.2e9));
        if (nstg == 12) continue;
    }
    return 0;
}
