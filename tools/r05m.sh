# host side of a c5 frame: which HIP calls hnb_simulate makes and what each costs (rocprofv3 --hip-trace --stats, small capacity so that nothing waits for the GPU)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out; mkdir -p $O
rm -rf $O/m_hip
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $O/m_hip -- python $R/bench.py --config c5 --capacity 65536 --steps 20 --no-parity --pmc off --no-cpu-baseline --no-scene --no-comm --no-extra-configs --full-json /tmp/z.json < /dev/null > $O/m_hip.log 2>&1
f=$(find $O/m_hip -name "*hip_api_stats.csv" | head -1)
echo "files: $(find $O/m_hip -name '*.csv' | xargs -n1 basename | tr '\n' ' ')" > $O/r05m_c5_hip_calls.txt
[ -n "$f" ] && python3 - "$f" <<'PY' >> $O/r05m_c5_hip_calls.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("TotalNs", 0)) or 0))
frames = 120 + 25 * 20 + 5
for r in rows[:16]:
    calls = int(r["Calls"]); avg = float(r.get("AverageNs", 0)) / 1e3
    print("%-34s calls %7d (%.2f per frame)  avg %7.2f us  min %7.2f  max %9.2f" % (r["Name"][:34], calls, calls / frames, avg, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep '^{"metric"' $O/m_hip.log | tail -1 | python3 -c 'import sys,json; l=json.loads(sys.stdin.read()); print("ms_per_step under --hip-trace:", l["ms_per_step"])' >> $O/r05m_c5_hip_calls.txt 2>&1
rm -rf $O/m_hip
cat $O/r05m_c5_hip_calls.txt
