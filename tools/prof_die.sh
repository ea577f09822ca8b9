#!/bin/bash
# Development tool (GPU box): per-dispatch kernel times of the firework die-off frames (update / list rows / compact).
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_die -- python $R/tools/bench_configs.py c2die < /dev/null > $R/gpurun_out/prof_die.log 2>&1
f=$(find $R/gpurun_out/prof_die -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
by=collections.defaultdict(list)
for r in rows: by[r["Kernel_Name"].split("(")[0][:50]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in by.items():
    print("%-52s n=%3d last26: %s"%(k,len(v)," ".join("%.0f"%x for x in v[-26:])))
PY
tail -30 $R/gpurun_out/prof_die.log
