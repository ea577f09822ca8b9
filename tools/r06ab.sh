#!/bin/bash
# round 6: what is in the age kernel's 10.7 us beyond the 5.9 of its access pattern? timing-only builds with pieces cut out (HNB_AGEK_CUT: 1 publish, 2 clock,
# 4 nobody may die, 8 per-chunk state stores, 16 the whole epilogue), rocprofv3 kernel statistics of the C5 bench
R=$GRAFT_REPO_ROOT; cd $R || exit 1
export HNB_JIT_CACHE=$R/bevy_hanabi_amd/jit_cache
export TMPDIR=/tmp; cd /tmp
L=$R/gpurun_out/r06ab_age_kernel_cuts.log; : > $L
for v in "" cut1 cut2 cut4 cut8 cut16 cut31 ""; do
  d=/tmp/prof_$v; rm -rf $d
  HNB_LIB=${v:+$R/tools/variants/libhanabi_$v.so} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --config c5 --no-parity --no-extra-configs --no-scene --no-cpu-baseline --pmc off --no-comm --windows 6 --full-json /tmp/x.json > /tmp/c5.json 2>/tmp/c5.err
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "${v:-product}" <<'PY' | tee -a $L
import csv,sys
o=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    for k in ("k_update_slots_stream_age","k_init<","k_compact"):
        if k in n: o.append("%s avg %.2f min %.2f max %.2f"%(k.strip("<"), float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
print("%-8s"%sys.argv[2], " | ".join(o))
PY
done
