#!/bin/bash
# SQ counters of C5's three kernels (what is a 19 us update of 37 MB waiting for?)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/c5pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/c5pmc -- python $R/bench.py --config c5 --no-cpu-baseline --no-extra-configs --no-parity --pmc off --no-scene --no-comm --steps 10 --windows 3 > /tmp/c5pmc.log 2>&1
  python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/c5pmc/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counters", open("/tmp/c5pmc.log").read()[-500:])
else:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:70]
        if "hnb::" in k: per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in per.items():
        print(k, {n: sorted(v)[len(v) // 2] for n, v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
done
