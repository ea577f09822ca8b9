# c5 with the list kept as a ring / rewritten, same box: why does the SAME update kernel take longer behind a ring frame?
# pass A: per-dispatch kernel trace (durations over time, gaps between dependent launches); pass B: SQ / GRBM / TCC counters of the update kernel
# (cycles against wall time = the effective clock; L2 hit rate).
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out; mkdir -p $O
OUT=$O/r05l_c5_ring_vs_rewrite.txt; : > $OUT
BENCH="python $R/bench.py --config c5 --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm --no-extra-configs --full-json /tmp/z.json"
for MODE in ring rewrite; do
  [ $MODE = rewrite ] && export HNB_CTX_OPTIONS=ring_lists=0 || unset HNB_CTX_OPTIONS
  rm -rf $O/l_trace_$MODE $O/l_pmc_$MODE
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/l_trace_$MODE -- $BENCH < /dev/null > $O/l_trace_$MODE.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l_pmc_$MODE -- $BENCH < /dev/null > $O/l_pmc_$MODE.log 2>&1
  if [ -z "$(find $O/l_pmc_$MODE -name '*counter_collection.csv' | head -1)" ]; then
    rm -rf $O/l_pmc_$MODE
    timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/l_pmc_$MODE -- $BENCH < /dev/null > $O/l_pmc_$MODE.log 2>&1
  fi
  echo "=== c5 $MODE: $(grep '^{"metric"' $O/l_trace_$MODE.log | tail -1 | python3 -c 'import sys,json
try:
    l=json.loads(sys.stdin.read()); print("ms_per_step (under the tracer)", l["ms_per_step"], "windows", l.get("windows",{}).get("ms_per_step_min_median_max"))
except Exception as e: print("no line:", e)')" >> $OUT
  python3 - $O/l_trace_$MODE $O/l_pmc_$MODE <<'PY' >> $OUT 2>&1
import csv, glob, sys, collections, statistics as st
def short(n):
    for k in ("k_init", "k_update_slots_stream", "k_compact", "k_count_rows", "copyBuffer"):
        if k in n: return k.replace("k_update_slots_stream", "k_update")
    return None
def trace(d):
    fs = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not fs: return []
    rows = []
    for r in csv.DictReader(open(fs[0])):
        s = short(r["Kernel_Name"])
        if s: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), s, int(r.get("Dispatch_Id", 0))))
    rows.sort()
    return rows
rows = trace(sys.argv[1])
print(f"pass A (kernel trace only): {len(rows)} dispatches")
if rows:
    steady = rows[-1500:]
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for i, (s, e, n, _) in enumerate(steady):
        dur[n].append((e - s) / 1e3)
        if i: gap[(steady[i - 1][2], n)].append((s - steady[i - 1][1]) / 1e3)
    for n, v in dur.items():
        v2 = sorted(v); print(f"  {n:10s} n={len(v):4d} mean {st.mean(v):6.2f} us  p10 {v2[len(v2)//10]:6.2f}  p50 {v2[len(v2)//2]:6.2f}  p90 {v2[len(v2)*9//10]:6.2f}")
    for k, v in gap.items():
        if len(v) > 20:
            v2 = sorted(v); print(f"  gap {k[0]:>10s} -> {k[1]:10s} n={len(v):4d} mean {st.mean(v):6.2f} us  p50 {v2[len(v2)//2]:6.2f}  p90 {v2[len(v2)*9//10]:6.2f}")
    frames = [(s, e) for s, e, n, _ in rows if n == "k_update"]
    period = [(frames[i + 1][0] - frames[i][0]) / 1e3 for i in range(len(frames) - 1)]
    tail = sorted(period[-400:]); print(f"  frame period (update start to update start), last 400: p50 {tail[len(tail)//2]:.2f} us  p10 {tail[len(tail)//10]:.2f}  p90 {tail[len(tail)*9//10]:.2f}")
    ud = [(e - s) / 1e3 for s, e in frames]
    print("  update duration by 60-frame bins:", " ".join(f"{st.mean(ud[i:i+60]):.1f}" for i in range(0, len(ud) - 59, 60)))
rows = trace(sys.argv[2])
fs = glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True)
print(f"pass B (counters): {len(rows)} dispatches, counter file: {bool(fs)}")
if fs:
    durs = {d: (e - s) / 1e3 for s, e, n, d in rows}
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    hdr = None
    for r in csv.DictReader(open(fs[0])):
        hdr = hdr or list(r.keys())
        n = short(r["Kernel_Name"])
        if n in ("k_update", "k_init", "k_compact"):
            per[n][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            if "Start_Timestamp" in r and "End_Timestamp" in r and int(r["Dispatch_Id"]) not in durs:
                durs[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("  columns:", hdr)
    for n, disp in per.items():
        ids = sorted(disp)[-200:]
        med = {c: st.median(disp[i][c] for i in ids if c in disp[i]) for c in disp[ids[-1]]}
        dd = [durs[i] for i in ids if i in durs]
        d_us = st.median(dd) if dd else float("nan")
        g = med.get("GRBM_GUI_ACTIVE", 0.0)
        print(f"  {n}: median of last {len(ids)} dispatches: duration {d_us:.2f} us; " + ", ".join(f"{c}={v:.5g}" for c, v in sorted(med.items())))
        if g and dd:
            print(f"     GRBM_GUI_ACTIVE / duration = {g / (d_us * 1e3):.3f} GHz (if summed over 8 XCDs: {g / 8 / (d_us * 1e3):.3f} GHz)")
        if med.get("TCC_HIT_sum") is not None and med.get("TCC_MISS_sum") is not None:
            h, m = med["TCC_HIT_sum"], med["TCC_MISS_sum"]; print(f"     L2 hit rate {h / max(h + m, 1):.3f} ({h:.4g} hits, {m:.4g} misses)")
        if med.get("SQ_WAVE_CYCLES"):
            wc = med["SQ_WAVE_CYCLES"]; print(f"     wave cycles: waiting {med.get('SQ_WAIT_ANY', 0) / wc:.3f}, issue-stalled {med.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}, active {med.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
else:
    print(open(sys.argv[2] + ".log").read()[-600:])
PY
  rm -rf $O/l_trace_$MODE $O/l_pmc_$MODE
done
cat $OUT
