#!/bin/bash
# A/B of the attention query-row split: the library with it (SC_ATTN_SPLIT=1 / 0) and the previous build (libspeechclip_hip_base.so), training step / P-large / ragged P-base / headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attnsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_varlen_gpu.py tests/test_f16_operands_gpu.py tests/test_dropout_gpu.py -q -x 2>&1 | tail -8 > $O/tests.log
run() {   # tag, env..., -- bench args
  local tag=$1; shift
  env "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['ms_per_step'], d.get('clock',{}).get('sclk_mhz_under_load'), d.get('loss'))
" >> $O/ab.log
}
COMMON="--steps 10 --warmup 3 --no-other-configs --cpu-pairs 0 --no-vendor-comparator"
for i in 1 2 3; do
  for cfg in "train --train" "large --model large" "varlen --varlen" "base"; do
    set -- $cfg; n=$1; shift
    run "$n split pass$i"   SC_ATTN_SPLIT=1 timeout 300 python bench.py $COMMON "$@"
    run "$n nosplit pass$i" SC_ATTN_SPLIT=0 timeout 300 python bench.py $COMMON "$@"
    [ $n = base ] && run "$n oldlib pass$i" SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_base.so timeout 300 python bench.py $COMMON "$@"
  done
done
cat $O/tests.log $O/ab.log
