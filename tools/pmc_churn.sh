#!/bin/bash
# Development tool (GPU box): HBM bytes of the update kernel under churn, small vs large instances.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for cfg in "65536 1024" "1048576 64"; do
  set -- $cfg
  for C in FETCH_SIZE WRITE_SIZE; do
    CHURN_CAP=$1 C4_INST=$2 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_churn_$1_$C -- python $R/tools/bench_configs.py churn < /dev/null > $R/gpurun_out/pmc_churn_$1_$C.log 2>&1
  done
done
