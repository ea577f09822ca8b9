#!/bin/bash
# refresh of the end-of-round evidence on the final sources: GPU tests, smoke, the default bench line (own PMC passes, traffic.json), kernel stats
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03zz}
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest_gpu.log; cat gpurun_out/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${T}_smoke.log
( time timeout 900 python bench.py --keep-pmc gpurun_out/${T}_pmc --write-traffic > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err ) 2>&1 | grep real
cp profiles/traffic.json gpurun_out/${T}_traffic.json 2>/dev/null
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
( cd /tmp; rm -rf /tmp/prof_$T; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -- python $R/bench.py --pmc off --no-cpu-baseline > $R/gpurun_out/${T}_bench_under_rocprof.json 2> /dev/null
  f=$(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${T}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"]: print("%-100s calls %5s avg %9.2f us min %9.2f max %9.2f"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
) | tee gpurun_out/${T}_kernel_stats.txt
for m in 1 0; do echo "HNB_SCENE_MERGE=$m copies=1: $(HNB_SCENE_MERGE=$m timeout 300 python tools/scene_bench.py 1 600 2>&1 | tail -1)"; done | tee gpurun_out/${T}_scene.log
