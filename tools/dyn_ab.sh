#!/bin/bash
O=gpurun_out/dyn; mkdir -p $O
python tools/gemm_modes_check.py check > $O/check.txt 2>&1; tail -3 $O/check.txt
python -m pytest tests/test_gemm8p_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/gemm8p_fuzz.py > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
python tools/gemm_modes_check.py time --modes 16,26 qkv fc1 out out_res fc2 fc2_res conv1 out_l_f32 vit_fc2_f32 2>&1 | grep mode | tee $O/time.txt
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2 3; do
  for m in -1 26; do
    SC_GEMM_KERNEL_MODE=$m timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $m pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step_ab.txt
  done
done
B="python bench.py --model large --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2; do
  for m in -1 26; do
    SC_GEMM_KERNEL_MODE=$m timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('large mode $m pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step_ab.txt
  done
done
