#!/bin/bash
# merged launches for small programs: parity of the scene, then the scene bench with the merge on / off (1 and 4 copies) + kernel stats
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03u}
timeout 900 python -m pytest tests/test_scene_merge.py tests/test_reference_examples.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
for i in 1 2; do for m in 1 0; do for copies in 1 4; do
  echo "HNB_SCENE_MERGE=$m copies=$copies: $(HNB_SCENE_MERGE=$m timeout 300 python tools/scene_bench.py $copies 600 2>&1 | tail -1)"
done; done; done 2>&1 | tee gpurun_out/${T}_scene.log
rm -rf /tmp/prof_scene
HNB_SCENE_MERGE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_scene -o p --output-format csv -- python tools/scene_bench.py 1 600 > /dev/null 2>&1
f=$(find /tmp/prof_scene -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/${T}_scene_kernel_stats.csv
cut -c1-400 $f | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    print('%-100s calls %6s avg %10.2f us total %10.2f ms'%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))" | head -24
