// Microbenchmark: per-CU throughput of the GEMM epilogue's store patterns.  Each block (512 threads, 8 waves as 2 x 4) writes 256 x 256
// bf16 tiles into a row-major matrix of pitch `ldc` elements, walking down M; wave (wm, wn) owns rows wm*128.., cols wn*64..
// pattern 0: per instruction 16 rows x 64 B  (lane (frow, fk): row frow, 16 B at column-bytes fk*16)         -- permlane epilogue today
// pattern 1: per instruction  8 rows x 128 B (full lines)
// pattern 2: per instruction 16 rows x 64 B but the two halves of a line issued back to back (same as 0, order check)
// pattern 3: per instruction 4 rows x 256 B (if the wave owned 128 columns)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(512) void k(char* C, long ldc_bytes, int tiles_per_block, int pattern, int nt, long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles_per_block; ++t) {
        char* tile = C + ((long)(blockIdx.x * tiles_per_block + t) * 256 + wm * 128) * ldc_bytes + wn * 128;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            char* dst;
            if (pattern == 0 || pattern == 2) {
                const int frow = lane & 15, fk = lane >> 4;
                dst = tile + (long)((i >> 1) * 16 + frow) * ldc_bytes + (i & 1) * 64 + fk * 16;
            } else if (pattern == 1) {
                dst = tile + (long)(i * 8 + (lane >> 3)) * ldc_bytes + (lane & 7) * 16;
            } else if (pattern == 4) {   // full lines, but the 8 lanes of a line scattered over the wave (frow = lane & 15, fk = lane >> 4)
                const int frow = lane & 15, fk = lane >> 4, chunk = ((fk & 1) << 1) | (fk >> 1);
                dst = tile + (long)((i >> 1) * 16 + (i & 1) * 8 + (frow & 7)) * ldc_bytes + (frow < 8 ? 0 : 64) + chunk * 16;
            } else if (pattern == 5) {   // half lines, lane-adjacent quads: lanes 4r..4r+3 -> row r
                dst = tile + (long)((i >> 1) * 16 + (lane >> 2)) * ldc_bytes + (i & 1) * 64 + (lane & 3) * 16;
            } else if (pattern == 6) {   // full lines from lane PAIRS of quads: lanes 8r..8r+7 -> row r, but chunk order 0 2 1 3 | 4 6 5 7
                const int c = lane & 7, cc = (c & 4) | ((c & 1) << 1) | ((c >> 1) & 1);
                dst = tile + (long)(i * 8 + (lane >> 3)) * ldc_bytes + cc * 16;
            } else {
                // 4 rows x 256 B: pretend wave tile is 64 rows x 128 cols (wm in 0..3 by wave>>1, wn = wave&1)
                char* tile2 = C + ((long)(blockIdx.x * tiles_per_block + t) * 256 + (wave >> 1) * 64) * ldc_bytes + (wave & 1) * 256;
                dst = tile2 + (long)(i * 4 + (lane >> 4)) * ldc_bytes + (lane & 15) * 16;
            }
            if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst), "v"(v) : "memory");
            else *(u32x4*)dst = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (tid == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
int main(int argc, char** argv) {
    const long N = 2304, ldc_bytes = N * 2;
    const int tiles = 64;
    char* C; hipMalloc(&C, (size_t)256 * tiles * 256 * ldc_bytes + (1 << 20));
    long long* cyc; hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 64})
    for (int nt = 0; nt < 1; ++nt)
    for (int pattern = 0; pattern < 7; ++pattern) {
        k<<<grid, 512>>>(C, ldc_bytes, 4, pattern, nt, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 512>>>(C, ldc_bytes, tiles, pattern, nt, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
        const double bytes = (double)grid * tiles * 131072.0;
        printf("grid=%3d nt=%d pattern=%d: %.3f ms  %.2f TB/s  counter ticks/tile %.0f  -> %.1f B/clk/CU @2.4GHz (wall)\n", grid, nt, pattern, ms, bytes / ms / 1e9,
               avg / tiles, 131072.0 * tiles / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
