#!/bin/bash
# packed (ragged) batches: 4-wave (128-row) attention blocks instead of 8-wave (256-row) ones?  PROBES library (SC_ATTN_NW is read there), three interleaved passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attnnw; mkdir -p $O; cd $R
COMMON="--steps 10 --warmup 3 --no-other-configs --cpu-pairs 0 --no-vendor-comparator --no-clock-probe"
for i in 1 2 3; do
  for nw in 0 4; do
    for cfg in "varlen --varlen" "large_ragged --model large --varlen --audio-len 240000"; do
      set -- $cfg; n=$1; shift
      SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_probes.so SC_ATTN_NW=$nw timeout 300 python bench.py $COMMON "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$n SC_ATTN_NW=$nw pass $i', d['ms_per_step'], d.get('loss'))
" >> $O/ab.log
    done
  done
done
cat $O/ab.log
