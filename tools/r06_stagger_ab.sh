#!/bin/bash
# round 6: block stagger of fp32-epilogue launches with few tile rounds (SC_GEMM_STAGGER_F32 = percent of a tile's time): isolated shapes, then the P-large step
O=gpurun_out/r06l; mkdir -p $O
for pass in 1 2; do for st in 0 25 50 75; do
  echo "== stagger $st pass $pass" | tee -a $O/time.txt
  SC_GEMM_STAGGER_F32=$st python tools/gemm_modes_check.py time --modes 16 out_l_f32 fc2_l_f32 vit_fc2_f32 vit_out_f32 2>&1 | grep mode | tee -a $O/time.txt
done; done
B="python bench.py --model large --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2 3; do for st in 0 25 50 75; do
  SC_GEMM_STAGGER_F32=$st timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('large stagger $st pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step.txt
done; done
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2; do for st in 0 50; do
  SC_GEMM_STAGGER_F32=$st timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base stagger $st pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/step.txt
done; done
