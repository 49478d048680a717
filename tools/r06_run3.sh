#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
SPEECHCLIP_HIP_LIB=tools/ab/libumode.so python -m pytest tests/test_gemm8p_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -k "gelu or p_base" > $O/umode_tests.txt 2>&1; tail -3 $O/umode_tests.txt
for n in mo1 mo2; do SPEECHCLIP_HIP_LIB=tools/ab/lib$n.so python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -x > $O/${n}_tests.txt 2>&1; tail -1 $O/${n}_tests.txt; done
AB_TESTS=0 tools/r06_ab.sh r06h mo1 mo2 umode
