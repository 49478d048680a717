O=gpurun_out/ab; mkdir -p $O; rm -f $O/nt.txt
export SPEECHCLIP_HIP_LIB=speechclip_amd/libspeechclip_hip_probes.so
B="python bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-roofline-events"
for pass in 1 2 3; do
  for arm in "base:0:0" "qkv:2304:0" "fc1:3072:1" "conv:512:1"; do
    n=${arm%%:*}; r=${arm#*:}; N=${r%%:*}; A=${r#*:}
    SC_GEMM_NT_N=$N SC_GEMM_NT_ACT=$A timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', $pass, d['ms_per_step'], d['loss'])" | tee -a $O/nt.txt
  done
done
