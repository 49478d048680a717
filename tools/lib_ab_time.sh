#!/bin/bash
# isolated-launch A/B of the product library against tools/ab/lib<name>.so ...: correctness first, then TF/s per step shape, then the step
#   tools/lib_ab_time.sh ew0 [more names]
O=gpurun_out/libab; mkdir -p $O
python tools/gemm_modes_check.py check > $O/check.txt 2>&1; tail -1 $O/check.txt
python tools/gemm8p_fuzz.py > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
SH="${AB_SHAPES:-out_res fc2_res out_l_f32 fc2_l_f32 vit_fc2_f32 vit_out_f32 qkv}"
for pass in 1 2; do
  for n in base "$@"; do
    if [ $n = base ]; then L=; else L=tools/ab/lib$n.so; fi
    echo "== $n pass $pass" | tee -a $O/time.txt
    SPEECHCLIP_HIP_LIB=$L python tools/gemm_modes_check.py time --modes 16 $SH 2>&1 | grep mode | tee -a $O/time.txt
  done
done
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
