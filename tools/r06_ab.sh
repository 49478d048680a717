#!/bin/bash
# round 6: in-step A/B of the product library against tools/ab/lib<name>.so (three interleaved passes), after the GEMM / GELU / parity tests
#   tools/r06_ab.sh <tag> <name> [more names]   (AB_TESTS=0 skips the tests, AB_LARGE=1 adds P-large)
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
if [ "${AB_TESTS:-1}" = 1 ]; then
  python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py tests/test_headline_parity_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
fi
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2 3; do
  for n in base "$@"; do
    if [ $n = base ]; then L=; else L=tools/ab/lib$n.so; fi
    SPEECHCLIP_HIP_LIB=$L timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step.txt
  done
done
if [ -n "$AB_LARGE" ]; then
B="python bench.py --model large --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2; do
  for n in base "$@"; do
    if [ $n = base ]; then L=; else L=tools/ab/lib$n.so; fi
    SPEECHCLIP_HIP_LIB=$L timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('large $n pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step.txt
  done
done
fi
