#!/bin/bash
O=gpurun_out/r06w; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv0" 2>&1 | tail -n 3
python -m pytest tests/test_headline_parity_gpu.py tests/test_e2e_gpu.py tests/test_varlen_gpu.py -m gpu -q 2>&1 | tail -n 2
B="python bench.py --model large --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2 3; do
  for f in 1 0; do
    SC_CONV0_LN_FUSED=$f timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('large fused=$f pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step.txt
    SC_CONV0_LN_FUSED=$f timeout 300 $B --varlen 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('large varlen fused=$f pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step.txt
  done
done
