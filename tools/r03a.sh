#!/bin/bash
# round 3, first GPU call: GPU tests on the round-2 kernels + ADVICE fixes, and where the general path (c2_mixed) stands today
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03a_pytest.log; cat gpurun_out/r03a_pytest.log
timeout 300 python tools/mixed_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a_mixed.log
ORDER=slot timeout 300 python tools/mixed_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a_mixed_slot.log
cd /tmp
WARM=200 FRAMES=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03a_prof -- python $R/tools/mixed_probe.py > /dev/null 2>&1
f=$(find $R/gpurun_out/r03a_prof -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r03a_mixed_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"]: print("%-100s calls %5s avg %9.2f us min %9.2f max %9.2f"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
for C in FETCH_SIZE WRITE_SIZE; do
  WARM=200 FRAMES=10 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/r03a_pmc_$C -- python $R/tools/mixed_probe.py > /dev/null 2>&1
  ff=$(find $R/gpurun_out/r03a_pmc_$C -name "*counter_collection.csv" | head -1)
  python3 - "$ff" $C <<'PY'
import csv,sys,statistics,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]==sys.argv[2] and "k_" in r["Kernel_Name"]: d[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
for k,v in d.items(): print(sys.argv[2], "%-90s n %4d median(last 30) %12.1f KiB"%(k,len(v),statistics.median(v[-30:])))
PY
done
rm -rf $R/gpurun_out/r03a_prof $R/gpurun_out/r03a_pmc_FETCH_SIZE $R/gpurun_out/r03a_pmc_WRITE_SIZE
