#!/bin/bash
# PROBES build: cache-policy bits (sc0 | sc1 << 1 | nt << 2) of the fast epilogue's output stores, in-step A/B (ms per step) + sustained GEMM TF/s
O=gpurun_out/ab; mkdir -p $O; rm -f $O/stpol.txt
export SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_probes.so
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events"
for pass in 1 2; do
  for pol in 0 1 2 3 4 5 6 7; do
    SC_GEMM_ST_POLICY=$pol timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy', $pol, 'pass', $pass, d['ms_per_step'], d['loss'])" | tee -a $O/stpol.txt
  done
done
for pol in 0 1 2 3 4 5 6 7; do echo -n "policy $pol: " | tee -a $O/stpol.txt; SC_BENCH_SUSTAIN=1 SC_GEMM_ST_POLICY=$pol timeout 200 python tools/gemm_bench.py qkv fc1 conv1 2>&1 | tail -1 | tee -a $O/stpol.txt; done
