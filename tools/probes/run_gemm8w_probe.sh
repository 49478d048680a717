#!/bin/bash
# GPU box: the free-running 8-wave variants (two waves per SIMD, 128 x 64 wave tiles, one barrier per k-step) of the k-loop prototype
out=gpurun_out/${1:-r06c}; mkdir -p $out
P=tools/probes
$P/gemm4w_probe_bin 1.0 11 12 > $out/gemm8w.txt 2>&1
$P/gemm4w_probe_h4_bin 1.0 3 12 >> $out/gemm8w.txt 2>&1
for a in 2 4; do $P/gemm4w_probe_abl${a}_bin 1.0 3 12 >> $out/gemm8w.txt 2>&1; done
cat $out/gemm8w.txt
