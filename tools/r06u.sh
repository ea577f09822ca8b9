#!/bin/bash
# round 6: the access pattern of C5's update on its own (tools/probes/age_stream_probe.hip), and the product's C5 kernels by rocprofv3's kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
tools/probes/age_stream_probe 2>&1 | tee gpurun_out/r06u_age_stream_probe.log
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --no-parity --no-extra-configs --no-scene --no-cpu-baseline --pmc off --no-comm --windows 10 --full-json /tmp/x.json > /tmp/c5.json 2>/tmp/c5.err
tail -c 600 /tmp/c5.json
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r06u_c5_kernel_stats.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls %5s avg %9.2f us min %9.2f max %9.2f  %5.1f%%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
# the timeline of a few steady-state frames: start / end of every kernel
t=$(find /tmp/prof_c5 -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r06u_c5_kernel_stats.txt
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); sel=rows[n//2:n//2+16]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    print("%-60s start %8.2f us  dur %7.2f us"%(r["Kernel_Name"][:60],(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
