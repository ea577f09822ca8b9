#!/bin/bash
# Development tool (GPU box): the default bench command under rocprofv3 — kernel statistics in one run, then the HBM
# counters in two more (FETCH_SIZE and WRITE_SIZE cannot share a pass, and PMC runs never combine with the stats/trace
# domains gpurun refuses). Writes gpurun_out/prof_<tag>* and, through tools/pmc_traffic.py, profiles/traffic.json +
# profiles/<tag>_pmc_*.csv. Usage: tools/prof_bench.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out
rm -rf $O/prof_$TAG $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python $R/bench.py --no-cpu-baseline < /dev/null > $O/prof_${TAG}_bench.json 2> $O/prof_${TAG}_bench.err
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/prof_${TAG}_kernel_stats.csv && python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.05 or "k_" in r["Name"]:
        print("%-90s calls %5s avg %9.2f us min %9.2f max %9.2f  %5.1f%%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${TAG}_$C -- python $R/bench.py --no-cpu-baseline < /dev/null > $O/pmc_${TAG}_$C.json 2> $O/pmc_${TAG}_$C.err
done
ff=$(find $O/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find $O/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python3 $R/tools/pmc_traffic.py "$ff" "$fw" $TAG && cp $R/profiles/traffic.json $R/profiles/${TAG}_pmc_*.csv $O/
# the raw trees are large: keep only the summaries for the merge back
rm -rf $O/prof_$TAG $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE
