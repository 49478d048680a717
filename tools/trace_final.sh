cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
for n in fwd fwd_serial_towers; do
  if [ $n = fwd_serial_towers ]; then export SC_OVERLAP_VIT=0; fi
  timeout 600 rocprofv3 --kernel-trace -d $O/$n.d -o $n -- python $R/bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-vendor-comparator --no-clock-probe > $O/$n.json 2> $O/$n.err
  db=$(find $O/$n.d -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/${n}_stats.txt
done
head -8 $O/fwd_serial_towers_stats.txt | cut -c1-120
