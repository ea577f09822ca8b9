#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03y}
timeout 900 python -m pytest tests/test_scene_merge.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do echo "$(timeout 300 python tools/scene_bench.py 1 600 2>&1 | tail -1)"; done | tee gpurun_out/${T}_scene.log
rm -rf /tmp/prof_scene
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_scene -o p --output-format csv -- python tools/scene_bench.py 1 600 > /dev/null 2>&1
f=$(find /tmp/prof_scene -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/${T}_scene_kernel_stats.csv
cut -c1-400 $f | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    print('%-100s calls %6s avg %10.2f us total %10.2f ms'%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))" | head -8
