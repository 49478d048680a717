#!/bin/bash
# copy the round-6 evidence set from gpurun_out/r06 (written by tools/r06_collect.sh on the GPU box) into profiles/ under the names profiles/INDEX_r06.md lists
O=gpurun_out/r06; P=profiles
cp $O/default_run.json $P/r06_bench_default_run.json; cp $O/train_run.json $P/r06_bench_train_run.json; cp $O/varlen_run.json $P/r06_bench_varlen_run.json
cp $O/large_run.json $P/r06_bench_large_run.json; cp $O/large_varlen_run.json $P/r06_bench_large_varlen_run.json; cp $O/casc_run.json $P/r06_bench_casc_run.json
cp $O/fwd_stats.txt $P/r06_bench_fwd_kernel_stats.txt; cp $O/fwd_serial_towers_stats.txt $P/r06_bench_fwd_serial_towers_kernel_stats.txt; cp $O/train_stats.txt $P/r06_bench_train_kernel_stats.txt
cp $O/casc_fwd_stats.txt $P/r06_bench_casc_fwd_kernel_stats.txt; cp $O/casc_train_stats.txt $P/r06_bench_casc_train_kernel_stats.txt; cp $O/large_fwd_stats.txt $P/r06_bench_large_fwd_kernel_stats.txt; cp $O/varlen_fwd_stats.txt $P/r06_bench_varlen_fwd_kernel_stats.txt
cp $O/pmc_traffic.txt $P/r06_bench_pmc_hbm_traffic.txt; cp $O/gemm_launches.json $P/r06_gemm_launches.json; cp $O/gemm_hbm_traffic.json $P/r06_gemm_hbm_traffic.json
cp $O/pmc_mfma_summary.txt $P/r06_bench_pmc_mfma_util.txt; cp $O/step_gemm_breakdown.txt $P/r06_step_gemm_breakdown.txt; cp $O/tile_trace.txt $P/r06_gemm8p_tile_trace.txt
