#!/bin/bash
# where the image tower enters the launch sequence (SC_VIT_START): three interleaved passes of the default step
O=gpurun_out/vitstart; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events --no-other-configs"
for pass in 1 2 3; do
  for v in "" extractor layer0 layer4 layer8; do
    SC_VIT_START=$v timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('start=[$v] pass $pass', d['ms_per_step'], d.get('loss'))" | tee -a $O/ab.txt
  done
done
