#!/bin/bash
# Development tool (GPU box): rocprofv3 kernel stats of the default bench command + the bench line of the same process.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --no-cpu-baseline < /dev/null > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
tail -1 $R/gpurun_out/prof_bench.json
f=$(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/prof_bench_kernel_stats.csv && python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %5s avg %9.2f us min %9.2f max %9.2f  %5.1f%%"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
