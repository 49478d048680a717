#!/bin/bash
# Where do the waves of gemm256_kernel spend their cycles?  SQ / TCP / TA counters of one GEMM shape under the k-loop probes
# (SC_GEMM_ABL = 0 baseline, 7 DMA on hot lines, 1 no in-loop DMA).  Counters only (no trace domains), one rocprofv3 run per counter group.
# usage (on the GPU box): bash tools/pmc_gemm_stalls.sh "qkv" -> gpurun_out/pmc_stalls/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_stalls
mkdir -p $O
SHAPE=${1:-qkv}
rocprofv3 -L > $O/avail.txt 2>&1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
G3="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
G4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_LDS"
G5="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
G6="TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum"
G7="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_REQ_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1))
  for abl in 0 7 1; do
    SC_GEMM_ABL=$abl timeout 300 rocprofv3 --pmc $G --output-format csv -d $O/g${i}_abl$abl -- python $R/tools/gemm_bench.py $SHAPE > $O/g${i}_abl$abl.log 2>&1
  done
done
python - "$O" <<'PY'
import glob,csv,collections,sys,os
O=sys.argv[1]
res=collections.defaultdict(dict)
for d in sorted(glob.glob(O+"/g*_abl*")):
    if not os.path.isdir(d): continue
    abl=d.rsplit("abl",1)[1]
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        per=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm256" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c,v in per.items(): res[c][abl]=sum(v[2:])/max(1,len(v[2:]))     # skip the two warm-up launches
print(f"{'counter':44s} {'ABL0':>14s} {'ABL7':>14s} {'ABL1':>14s}")
for c in sorted(res): print(f"{c:44s} " + " ".join(f"{res[c].get(a,float('nan')):14.4g}" for a in ("0","7","1")))
PY
