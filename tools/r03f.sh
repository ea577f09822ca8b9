#!/bin/bash
# round 3, sixth GPU call: cohort state 4 (mixed chunks run the lean per-particle path), k_emit_events v3; A/B of the cohort kernels' register budget
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03g}
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
( time timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err ) 2>&1 | grep real
python3 tools/bench_summary.py gpurun_out/${T}_bench.json | tee gpurun_out/${T}_bench_summary.txt
tail -c 300 gpurun_out/${T}_bench.err
for i in 1 2; do for lib in "" "$R/bevy_hanabi_amd/libhanabi_amd_w4.so"; do
  for cfg in c2 c2_mixed; do
  HNB_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs --config $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%-10s %-9s ms/step %.4f min %.4f kernel %.4f lists %.4f init %.4f'%('$lib'[-10:] or 'w5','$cfg',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg'],d['stages']['lists_ms_avg'],d['stages']['init_ms_avg']))"
  done
done; done 2>&1 | tee gpurun_out/${T}_waves_ab.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python $R/bench.py --no-cpu-baseline --pmc off --no-extra-configs --config c2_events > /dev/null 2>&1
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${T}_events_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"]: print("%-100s calls %5s avg %9.2f us min %9.2f max %9.2f"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $R/gpurun_out/${T}_prof
