// Internal interface between gemm.hip (the sc_gemm_bf16 dispatcher) and gemm_duet.hip (the phase-shifted two-group kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

struct DuetParams {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    void* C; int64_t ldc;                 // bf16, or f32 when out_f32 (gemm8p only)
    const float* bias;                 // may be null
    const void* residual; int64_t ldr;   // same element type as C
    int out_f32;
    int64_t M; int N; int K;
    int act;                           // SC_ACT_*
    int nk;                            // K / 64
    int kpair;                         // > 0: stride-2 kernel-3 conv as GEMM, walk K as (tap 0, tap 2) chunk pairs (see gemm.hip)
    int tn;                            // N / 256: blocks per row of blocks
    int rows;                          // rows of blocks (grid / tn); blocks beyond rows * tn exit at once
    int64_t units;                     // ceil(M / 128): 128-row half panels
    int esteps;                        // duet: epilogue steps per half tile: 4 or 8; gemm8p: 1 = the per-tile A/B kernel
    int band;                          // gemm8p: N tiles per column band of the tile order (0: all of N)
    unsigned long long* trace;         // PROBES: per-block cycle stamps
};

// 0: launched; 1: shape outside this kernel's rules (nothing launched; the caller runs gemm256_kernel); < 0: error
int sc_gemm_duet_try(const DuetParams& p, hipStream_t s);

// gemm8p.hip: the ping-pong ("8-phase") form.  Same return convention.
int sc_gemm8p_try(const DuetParams& p, hipStream_t s);
