R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out
d=/tmp/prof_scene_nomerge; rm -rf $d
HNB_CTX_OPTIONS=scene_merge=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python -c "
import sys
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tools')
import scene_bench
scene_bench.run(1, 300, quiet=True, set_module=0)
" < /dev/null > $O/scene_nomerge.out 2>&1
f=$(find $d -name '*kernel_stats.csv' | head -1)
cp "$f" $O/scene_nomerge_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print('%-110s calls %6s avg %8.2f us  min %7.2f  max %8.2f' % (r['Name'].split('(')[0][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
