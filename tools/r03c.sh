#!/bin/bash
# round 3, third GPU call: new math + died-bit lists through the full GPU suite; the new bench line with live PMC; A/Bs
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03c}
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.log; cat gpurun_out/${T}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --keep-pmc gpurun_out/${T}_pmc --write-traffic > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err ) 2>&1 | grep real
cp profiles/traffic.json gpurun_out/${T}_traffic.json
python3 - gpurun_out/${T}_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def line(k,v):
    r=v["roofline"]; st=v["stages"]
    print("%-11s value %.3e ms/step %.4f (min %.4f) | init %.4f upd %.4f lists %.4f | moved %.1f B/upd frac %.3f alg-whole-step %.3f | %s"%(k,v["value"],v["ms_per_step"],v["windows"]["min_ms_per_step"],st["init_ms_avg"],st["update_ms_avg"],st["lists_ms_avg"],r.get("moved_bytes_per_update") or 0,r["frac"],r["algorithmic"]["whole_step_over_peak"],r["traffic_source"][:40]))
    if v.get("init"): print("            init burst %.4f ms frac %.3f"%(v["init"]["kernel_ms"],v["init"]["frac"]))
line("c2",d)
for k,v in d.get("configs",{}).items():
    if "error" in v: print(k,v)
    else: line(k,v)
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","threads","host_physical_cores","error")})
PY
tail -c 600 gpurun_out/${T}_bench.err
for i in 1 2; do for lib in "" "$R/bevy_hanabi_amd/libhanabi_amd_w6.so"; do
  HNB_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --pmc off --no-extra-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=%-10s c2 ms/step %.4f min %.4f kernel %.4f'%('$lib'[-10:] or 'w5',d['ms_per_step'],d['windows']['min_ms_per_step'],d['stages']['update_ms_avg']))"
  HNB_LIB=$lib FRAMES=40 timeout 300 python tools/mixed_probe.py 2>&1 | grep "^mixed" | tail -1
done; done 2>&1 | tee gpurun_out/${T}_waves_ab.log
for M in 1 2 0; do echo "== HNB_COUNT_LOAD=$M"; HNB_COUNT_LOAD=$M FRAMES=40 timeout 300 python tools/mixed_probe.py 2>&1 | grep "^mixed" | tail -1; done 2>&1 | tee gpurun_out/${T}_count_load.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -- python $R/bench.py --no-cpu-baseline --pmc off > $R/gpurun_out/${T}_bench_under_rocprof.json 2>/dev/null
f=$(find $R/gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"]: print("%-100s calls %5s avg %9.2f us min %9.2f max %9.2f"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $R/gpurun_out/${T}_prof
