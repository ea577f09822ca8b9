#!/bin/bash
# Development tool (GPU box): the default bench command under `rocprofv3 --kernel-trace --stats` (one process over every configuration; the
# bench's own counter passes are switched off inside it: a profiler does not nest) -> gpurun_out/<tag>_kernel_stats.csv + a readable summary.
# Usage: tools/prof_round.sh <tag>
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
rm -rf $O/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python $R/bench.py --steps 20 --warmup 5 --pmc off --no-cpu-baseline < /dev/null > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof_bench.err
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats.csv && python3 - "$f" > $O/${TAG}_kernel_stats.txt <<'PY'
import csv, sys
print("rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --pmc off --no-cpu-baseline   (one process, every configuration)")
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.05 or "hnb::" in r["Name"]:
        print("%-100s calls %6s avg %9.2f us min %9.2f max %9.2f  %5.1f%%" % (r["Name"].split("(")[0][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
cat $O/${TAG}_kernel_stats.txt | head -40
rm -rf $O/prof_$TAG
