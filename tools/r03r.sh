#!/bin/bash
# per-kernel durations of the churn configs on the current build
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03r}
for cfg in c2_mixed c2_events c2_dieoff; do
  rm -rf /tmp/prof_$cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o p --output-format csv -- python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg > /dev/null 2>&1
  f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1)
  echo "== $cfg"; cut -c1-400 $f | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    print('%-100s calls %6s avg %10.2f us min %9.2f max %9.2f'%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))" | head -12
done 2>&1 | tee gpurun_out/${T}_kernels.log
