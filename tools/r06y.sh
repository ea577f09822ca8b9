#!/bin/bash
# round 6: where does a frame of the 26-effect scene go now that the host no longer waits for an upload? kernel statistics + the timeline of a few frames
R=$GRAFT_REPO_ROOT; cd $R || exit 1
export HNB_JIT_CACHE=$R/bevy_hanabi_amd/jit_cache
for i in 1 2; do timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1; done | tee gpurun_out/r06y_scene.log
export TMPDIR=/tmp; cd /tmp; d=/tmp/prof_scene; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python -c "
import sys
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tools')
import scene_bench
scene_bench.run(1, 300, quiet=True)
" < /dev/null > /tmp/scene.out 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee -a $R/gpurun_out/r06y_scene.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print('%-70s calls %6s avg %8.2f us  min %7.2f  max %8.2f  %5.1f %%' % (r['Name'].split('(')[0][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['Percentage'])))
PY
t=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY' | tee -a $R/gpurun_out/r06y_scene.log
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); sel=rows[n*2//3:n*2//3+24]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    print("%-50s start %8.2f us  dur %7.2f us  grid %s wg %s"%(r["Kernel_Name"].split('(')[0][:50],(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X", r.get("Grid_Size","?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size","?"))))
PY
