#!/bin/bash
# round-6 evidence in one gpurun call
R=$GRAFT_REPO_ROOT; cd $R
bash tools/collect_round_profiles.sh r06
bash tools/pmc_mfma.sh > gpurun_out/r06/pmc_mfma_summary.txt 2>&1
cd $R
SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_probes.so python tools/gemm_modes_check.py trace8pp qkv fc1 out_res fc2_res conv1 > gpurun_out/r06/tile_trace.txt 2>&1
python tools/blas_compare.py > gpurun_out/r06/blas_compare.txt 2>&1
tail -c 400 gpurun_out/r06/default_run.json
