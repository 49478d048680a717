#!/bin/bash
# GPU box: MFMA power ceiling (socket watts from rocm-smi) + the 4-wave k-loop prototype and its ablations + the product kernel on the same shapes, same box.
out=gpurun_out/${1:-r06b}; mkdir -p $out
P=tools/probes
$P/mfma_power_probe_bin 2.5 > $out/mfma_power.txt 2>&1
$P/gemm4w_probe_bin 1.0 15 3 > $out/gemm4w.txt 2>&1
$P/gemm4w_probe_h8_bin 1.0 3 3 >> $out/gemm4w.txt 2>&1
for a in 1 2 3 4; do $P/gemm4w_probe_abl${a}_bin 1.0 3 3 >> $out/gemm4w.txt 2>&1; done
SC_BENCH_SUSTAIN=1 python tools/gemm_bench.py qkv fc2 out sq8k > $out/gemm8p_same_box.txt 2>&1
cat $out/mfma_power.txt $out/gemm4w.txt; tail -5 $out/gemm8p_same_box.txt
