#!/bin/bash
# Per-kernel durations of the 26-effect scene (tools/scene_bench.py), merged launches on the interpreters and on the set module.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
for mode in off default; do
  d=/tmp/prof_scene_$mode; rm -rf $d
  SCENE_MODE=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python -c "
import os, sys
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tools')
import scene_bench
scene_bench.run(1, 300, quiet=True, set_module=0 if os.environ['SCENE_MODE'] == 'off' else None)
" < /dev/null > $O/scene_$mode.out 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] || { echo "no stats for $mode"; tail -5 $O/scene_$mode.out; continue; }
  cp "$f" $O/scene_${mode}_kernel_stats.csv
  echo "== $mode"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print('%-70s calls %6s avg %8.2f us  min %7.2f  max %8.2f  %5.1f %%' % (r['Name'].split('(')[0][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['Percentage'])))
PY
done
