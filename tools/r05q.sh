# HBM-side traffic of EVERY kernel of a c5 frame, list kept as a ring / rewritten (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes; same formula as bench.py:
# FETCH_SIZE KiB x 1024 x 2 (gfx950) + WRITE_SIZE KiB x 1024, median launch of the steady state)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out; mkdir -p $O; OUT=$O/r05q_c5_traffic_per_kernel.txt; : > $OUT
BENCH="python $R/bench.py --config c5 --steps 10 --windows 5 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm --no-extra-configs --full-json /tmp/z.json"
for MODE in ring rewrite; do
  [ $MODE = rewrite ] && export HNB_CTX_OPTIONS=ring_lists=0 || unset HNB_CTX_OPTIONS
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/q_${MODE}_$C
    timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/q_${MODE}_$C -- $BENCH < /dev/null > $O/q_${MODE}_$C.log 2>&1
  done
  echo "=== c5, list $MODE" >> $OUT
  python3 - $O/q_${MODE}_FETCH_SIZE $O/q_${MODE}_WRITE_SIZE <<'PY' >> $OUT 2>&1
import csv, glob, sys, collections, statistics as st
def med(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(list)
    if not fs: return per
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter: continue
        n = r["Kernel_Name"].split("(")[0]
        n = "k_update_slots_stream" if "k_update_slots_stream" in n else n.replace("void ", "").replace("hnb::", "")
        per[n].append(float(r["Counter_Value"]))
    return per
f, w = med(sys.argv[1], "FETCH_SIZE"), med(sys.argv[2], "WRITE_SIZE")
tot = 0.0
for n in sorted(set(f) | set(w)):
    if not any(k in n for k in ("k_init", "k_update", "k_compact", "k_count_rows")): continue
    fv, wv = f.get(n, [0.0]), w.get(n, [0.0])
    fs, ws = st.median(fv[-150:]), st.median(wv[-150:])     # steady state: the last launches
    mb = (fs * 1024 * 2 + ws * 1024) / 1e6
    if len(fv) > 50: tot += mb
    print(f"  {n[:40]:40s} launches {len(fv):4d}  FETCH_SIZE {fs:10.1f} KiB  WRITE_SIZE {ws:10.1f} KiB  -> {mb:8.3f} MB per launch")
print(f"  per frame (kernels that run every frame): {tot:.2f} MB")
PY
  rm -rf $O/q_${MODE}_FETCH_SIZE $O/q_${MODE}_WRITE_SIZE
done
cat $OUT
