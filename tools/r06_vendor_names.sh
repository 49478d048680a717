#!/bin/bash
# which kernels does the hipBLASLt comparator run for the step's plain GEMMs?  (full Tensile names carry macro tile, MFMA shape, wave tiling, stream-K ...)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O
SC_GEMM_VENDOR=1 SC_OVERLAP_VIT=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o v -- python $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe --no-other-configs --no-roofline-events > $O/v.json 2> $O/v.err
python - "$O" <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]; agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(O+"/tr/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r.get("Kernel_Name") or r.get("kernel_name"); d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
        if n and n.startswith("Cijk"): agg[n][0]+=1; agg[n][1]+=d
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(f"{c:5d} launches {t:9.3f} ms total {t/c:7.4f} ms avg  {n}")
PY
