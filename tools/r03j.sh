#!/bin/bash
# per-kernel durations of c3 / c5, library before the horizon commit against the current one
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03j}
for v in prev cur; do for cfg in c3 c5; do
  case $v in prev) export HNB_LIB=$R/bevy_hanabi_amd/libhanabi_amd_prev.so;; cur) unset HNB_LIB;; esac
  rm -rf /tmp/prof_$v$cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v$cfg -o p --output-format csv -- python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg > /dev/null 2>&1
  f=$(find /tmp/prof_$v$cfg -name '*kernel_stats.csv' | head -1)
  echo "== $v $cfg"; cut -c1-400 $f | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    print('%-90s calls %6s avg %10.2f us min %9.2f max %9.2f'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))" | head -12
done; done 2>&1 | tee gpurun_out/${T}_kernels.log
