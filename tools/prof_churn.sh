#!/bin/bash
# Development tool (GPU box): per-kernel times of the churn workload.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_churn -- python $R/tools/bench_configs.py churn < /dev/null > $R/gpurun_out/prof_churn.log 2>&1
f=$(find $R/gpurun_out/prof_churn -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s avg %9.1f us min %9.1f max %9.1f"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
