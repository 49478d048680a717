// Internal interface between gemm.hip (the sc_gemm_bf16 dispatcher) and gemm8p.hip (the ping-pong kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

struct Gemm8pParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    void* C; int64_t ldc;                 // bf16, or f32 when out_f32 (gemm8p only)
    const float* bias;                 // may be null
    const void* residual; int64_t ldr;   // same element type as C
    int out_f32;
    int f16;                           // A, W (and C / residual unless out_f32) are IEEE half instead of bf16 (SC_GEMM_F16; persistent kernel only)
    int64_t M; int N; int K;
    int act;                           // SC_ACT_*
    int nk;                            // K / 64
    int kpair;                         // > 0: stride-2 kernel-3 conv as GEMM, walk K as (tap 0, tap 2) chunk pairs (see gemm.hip); A/B only
    int tn;                            // N / 256
    int rows;                          // K-rotation switch (A/B): 0 off, 1 by M panel, 2 / 3 by N tile
    int esteps;                        // 1 = the per-tile A/B kernel, else the persistent kernel
    int band;                          // gemm8p: N tiles per column band of the tile order (0: all of N)
    int sched;                         // in: >= 0 dynamic tile order allowed, -1 static (A/B); gemm8p.hip replaces it by the launch's counter slot
    unsigned sched_gen;                // set by gemm8p.hip: the launch's generation of that slot (see g_sched)
    unsigned long long* trace;         // PROBES: per-block cycle stamps
};

// gemm8p.hip.  0: launched; 1: shape outside the kernel's rules (nothing launched; the caller runs gemm256_kernel / gemm_bf16_kernel); < 0: error
int sc_gemm8p_try(const Gemm8pParams& p, hipStream_t s);
