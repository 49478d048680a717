#!/bin/bash
# in-step A/B of the libraries under tools/ab (two passes each, interleaved with the product library)
O=gpurun_out/ab; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events"
for pass in $(seq 1 ${AB_PASSES:-2}); do
  for n in base "$@"; do
    if [ $n = base ]; then L=; else L=tools/ab/lib$n.so; fi
    SPEECHCLIP_HIP_LIB=$L timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', $pass, d['ms_per_step'], d['loss'])" | tee -a $O/ab.txt
  done
done
