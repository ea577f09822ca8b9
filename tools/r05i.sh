# kernel-trace statistics of c5 with the list kept as a ring / rewritten (same box)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out; mkdir -p $O
for MODE in ring rewrite; do
  [ $MODE = rewrite ] && export HNB_CTX_OPTIONS=ring_lists=0 || unset HNB_CTX_OPTIONS
  rm -rf $O/prof_c5_$MODE
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5_$MODE -- python $R/bench.py --config c5 --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm --no-extra-configs --full-json /tmp/z.json < /dev/null > /dev/null 2> $O/prof_c5_$MODE.err
  f=$(find $O/prof_c5_$MODE -name "*kernel_stats.csv" | head -1)
  echo "=== c5 $MODE" >> $O/r05i_c5_kernel_stats.txt
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $O/r05i_c5_kernel_stats.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "hnb::" in r["Name"] or "copyBuffer" in r["Name"]:
        print("%-70s calls %5s avg %8.2f us min %8.2f max %8.2f"%(r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $O/prof_c5_$MODE
done
cat $O/r05i_c5_kernel_stats.txt
