#!/bin/bash
# round 3, fourth GPU call: everything so far through the GPU suite (c2_mixed / c2_events at 16.7M, multi-GPU host path), the bench line incl. c2_events
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03d}
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --keep-pmc gpurun_out/${T}_pmc --write-traffic > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err ) 2>&1 | grep real
cp profiles/traffic.json gpurun_out/${T}_traffic.json
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
tail -c 400 gpurun_out/${T}_bench.err
timeout 300 python bench.py --gpus 2 --launcher threads --force-device 0 --steps 20 > gpurun_out/${T}_bench_threads_n2_dryrun.json 2> gpurun_out/${T}_bench_threads.err; tail -c 700 gpurun_out/${T}_bench_threads_n2_dryrun.json; tail -c 300 gpurun_out/${T}_bench_threads.err
timeout 300 python bench.py --gpus 2 --backend gloo --force-device 0 --steps 10 --pmc off > gpurun_out/${T}_bench_n2_dryrun.json 2> gpurun_out/${T}_bench_n2_dryrun.err; tail -c 500 gpurun_out/${T}_bench_n2_dryrun.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python $R/bench.py --no-cpu-baseline --pmc off > $R/gpurun_out/${T}_bench_under_rocprof.json 2>/dev/null
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"]: print("%-100s calls %5s avg %9.2f us min %9.2f max %9.2f"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $R/gpurun_out/${T}_prof
