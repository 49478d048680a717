#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
for n in mo3 mo4 mo5; do SPEECHCLIP_HIP_LIB=tools/ab/lib$n.so python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -x > $O/${n}_tests.txt 2>&1; tail -n 1 $O/${n}_tests.txt; done
AB_TESTS=0 tools/r06_ab.sh r06i mo3 mo4 mo5
