#!/bin/bash
# round 6: the timeline of c2_events (the real firework.rs: rocket -> sparkle trail + trails through GPU spawn events): which kernels are on the critical path?
R=$GRAFT_REPO_ROOT; cd $R || exit 1
export HNB_JIT_CACHE=$R/bevy_hanabi_amd/jit_cache
export TMPDIR=/tmp; cd /tmp; d=/tmp/prof_ev; rm -rf $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --config c2_events --no-parity --no-extra-configs --no-scene --no-cpu-baseline --pmc off --no-comm --windows 4 --full-json /tmp/x.json > /tmp/ev.json 2>/tmp/ev.err
tail -c 300 /tmp/ev.json; echo
t=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY' | tee $R/gpurun_out/r06ad_c2_events_timeline.log
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); sel=rows[n*3//4:n*3//4+40]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    print("%-64s q%-3s start %8.2f us  end %8.2f  dur %7.2f us  grid %s"%(r["Kernel_Name"].split('(')[0][:64], r.get("Queue_Id","?"),(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X", r.get("Grid_Size","?"))))
PY
