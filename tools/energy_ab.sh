#!/bin/bash
# Energy / clock A/B on the headline step: the socket sits at its power cap, so a variant that moves less data can win through the
# clock it leaves room for even when isolated kernel timings are neutral.  Runs the PROBES build (env-tunable dispatch) of the same sources:
#   default | column-banded tile order for every wide GEMM (fc1 / QKV too) | small-M launches on the 128^2 kernel | LayerNorm folded into the GEMMs
# and prints ms/step, sclk, socket power, J/step per pass (interleaved passes).   usage (GPU box): bash tools/energy_ab.sh [passes]
R=$GRAFT_REPO_ROOT
export SPEECHCLIP_HIP_LIB=$R/speechclip_amd/libspeechclip_hip_probes.so
P=${1:-2}
run() {  # label, env assignments...
    local label=$1; shift
    env "$@" python $R/bench.py --steps 20 --warmup 5 --cpu-pairs 0 --no-vendor-comparator --no-other-configs --no-roofline-events 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['clock'] or {}
print('%-34s %8.3f ms/step %8.1f pairs/s  sclk %5s MHz  %7s W  %6s J/step' % ('$label', d['ms_per_step'], d['value'], c.get('sclk_mhz_under_load'), c.get('socket_power_w'), c.get('joules_per_step')))"
}
for i in $(seq $P); do
    run "default" SC_NOP=1
    run "band=4 (fc1, QKV banded too)" SC_GEMM_BAND=4
    run "band=3" SC_GEMM_BAND=3
    run "min_tiles=160 (ViT out/fc2 on 128^2)" SC_GEMM_MIN_TILES=160
    run "min_tiles=460 (+ ViT QKV)" SC_GEMM_MIN_TILES=460
    run "SC_OVERLAP_VIT=0 (serial towers)" SC_OVERLAP_VIT=0
done
