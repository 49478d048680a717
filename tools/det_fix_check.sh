#!/bin/bash
# the packed-fp32 LayerNorm fix: determinism of the whole forward (B = 64 / 256, overlapped towers) + same-box A/B of the step time against the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/detfix; mkdir -p $O; cd $R
( echo "== new library"; timeout 400 python tools/determinism_probe.py 40 -1 256 2>&1 | tail -2; timeout 300 python tools/determinism_probe.py 40 -1 64 2>&1 | tail -2
  echo "== previous build (libspeechclip_hip_base.so)"; SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_base.so timeout 400 python tools/determinism_probe.py 40 -1 256 2>&1 | tail -2
  echo "== image-tower probe under front-end load, new library"; timeout 300 python tools/vit_race_probe.py 256 24 front default 2>&1 | tail -2 ) > $O/det.log 2>&1
COMMON="--steps 10 --warmup 3 --no-other-configs --cpu-pairs 0 --no-vendor-comparator"
for i in 1 2 3; do
  for lib in libspeechclip_hip.so libspeechclip_hip_base.so; do
    for cfg in "base" "large --model large"; do
      set -- $cfg; n=$1; shift
      SPEECHCLIP_HIP_LIB=speechclip_amd/$lib timeout 300 python bench.py $COMMON "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n $lib pass $i', d['ms_per_step'], (d.get('clock') or {}).get('sclk_mhz_under_load'), d.get('loss'))
" >> $O/ab.log
    done
  done
done
cat $O/det.log $O/ab.log
