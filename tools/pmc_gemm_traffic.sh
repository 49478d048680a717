#!/bin/bash
# FETCH_SIZE (x2: gfx950 correction, MI355X_MICROARCH.md) of single GEMM launches vs their algorithmic operand bytes.
# usage (GPU box): bash tools/pmc_gemm_traffic.sh [lib.so] -> gpurun_out/pmc_gemm_traffic/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm_traffic${2:+_$2}
mkdir -p $O
SHAPES=${SHAPES:-"conv1:4096000,512,1536,1024,1 conv1_n256:4096000,256,1536,1024,1 conv2:2048000,512,1536,1024,1 qkv:128000,2304,768,768,0 fc1:128000,3072,768,768,1 out:128000,768,768,768,0 fc2:128000,768,3072,3072,0"}
SPEECHCLIP_HIP_LIB=$1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/raw -- python $R/tools/gemm_traffic_probe.py $SHAPES > $O/run.log 2>&1
python - "$O" <<'PY' | tee $O/summary.txt
import glob, csv, sys
O = sys.argv[1]
shapes = [l.split() for l in open(O + "/run.log") if "algorithmic_read_bytes" in l]
rows = []
for f in glob.glob(O + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
rows.sort()
per = len(rows) // max(1, len(shapes))
for i, s in enumerate(shapes):
    mine = rows[i * per:(i + 1) * per][-3:]
    fetch = 2 * sum(v for _, v in mine) / max(1, len(mine))
    alg = float(s[6])
    print(f"{s[0]:12s} M={s[1]:>8s} N={s[2]:>5s} K={s[3]:>5s} lda={s[4]:>5s}  fetched {fetch/1e9:7.3f} GB  algorithmic {alg/1e9:7.3f} GB  ratio {fetch/alg:5.2f}")
PY
