#!/bin/bash
# Development tool (GPU box): HBM bytes fetched / written by the update kernel per slab placement
# (tools/bimodal_probe.py multi: five slabs in one context, placement search off).
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
export HNB_SLAB_CANDIDATES=1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_place_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_place_$C -- python $R/tools/bimodal_probe.py multi < /dev/null > $R/gpurun_out/pmc_place_$C.log 2>&1
  f=$(find $R/gpurun_out/pmc_place_$C -name "*counter_collection.csv" | head -1)
  echo "== $C ($f)"
  grep "effect" $R/gpurun_out/pmc_place_$C.log | head -5
  [ -n "$f" ] && python3 - "$f" $C <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_update_slots_stream" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[2]]
# dispatches in order; the active effect's kernel has the large value
vals=[float(r["Counter_Value"]) for r in rows]
big=[v for v in vals if v>1e5]
# group consecutive big values in runs of 31 (6 warm + 25 timed per effect)
print("dispatches", len(vals), "active", len(big))
per=31
for i in range(0, min(len(big), per*5), per):
    chunk=big[i:i+per]
    print("effect %d: mean %.4e  min %.4e max %.4e" % (i//per, sum(chunk)/len(chunk), min(chunk), max(chunk)))
PY
done
