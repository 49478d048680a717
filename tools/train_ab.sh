#!/bin/bash
# A/B of two library builds on the training step: SPEECHCLIP_HIP_LIB selects the .so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trainab; mkdir -p $O; cd $R
for i in 1 2 3; do
  for lib in libspeechclip_hip.so libspeechclip_hip_base.so; do
    SPEECHCLIP_HIP_LIB=speechclip_amd/$lib timeout 300 python bench.py --train --steps 10 --warmup 3 --no-other-configs --cpu-pairs 0 --no-vendor-comparator 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib pass $i', d['ms_per_step'], d.get('clock',{}).get('sclk_mhz_under_load'), d.get('loss'))
" >> $O/ab.log
  done
done
cat $O/ab.log
