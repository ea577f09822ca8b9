#!/bin/bash
# round 6: k_update_slots_stream_age with three kinds of step: tests, then c5 against the first version of the kernel (pub0) and the in-template path (base), kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
timeout 1500 python -m pytest tests -m gpu -x -q -k "ribbon or c5 or ring or scene or reference_examples or skip or horizon or verification or bench_gate or timed or frozen or visible" 2>&1 | tail -3 | tee gpurun_out/r06v_pytest.log
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_pub0.so tools/variants/libhanabi_base.so; do
r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config c5 --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['parity']['ok'])")
echo "round $round ${lib:-default} c5: $r" | tee -a gpurun_out/r06v_ab_age_kernel.log
done; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --no-parity --no-extra-configs --no-scene --no-cpu-baseline --pmc off --no-comm --windows 10 --full-json /tmp/x.json > /tmp/c5.json 2>/tmp/c5.err
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r06v_c5_kernel_stats.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 1: print("%-90s calls %5s avg %9.2f us min %9.2f max %9.2f  %5.1f%%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
