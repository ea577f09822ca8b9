#!/bin/bash
# Development tool (GPU box): requested bytes of the update kernel during the firework's die-off (tools/bench_configs.py c2die), per launch.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/dieoff_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/dieoff_$C -- python $R/tools/bench_configs.py c2die > $O/dieoff_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
def series(c):
    f = glob.glob(f"{O}/dieoff_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_update_slots_stream" in r["Kernel_Name"] and r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]
fe, wr = series("FETCH_SIZE"), series("WRITE_SIZE")
print("launch  fetched_MB(x2)  written_MB  B/slot")
for i, (a, b) in enumerate(zip(fe, wr)):
    if i % 3 == 0 or i > 44:
        print(i, round(a * 2 * 1024 / 1e6, 1), round(b * 1024 / 1e6, 1), round((a * 2 + b) * 1024 / 16777216, 1))
PY
rm -rf $O/dieoff_FETCH_SIZE $O/dieoff_WRITE_SIZE
