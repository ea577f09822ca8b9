#!/bin/bash
# Development tool (GPU box): per-kernel times of the C5 ribbon churn incl. the sort kernels.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sort -- python $R/tools/bench_configs.py c5 < /dev/null > $R/gpurun_out/prof_sort.log 2>&1
f=$(find $R/gpurun_out/prof_sort -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-160
tail -2 $R/gpurun_out/prof_sort.log
