#!/bin/bash
# Development tool (GPU box): the end-of-round check in one gpurun call: the GPU suite, smoke, the default bench line (its own two PMC passes kept as
# CSVs), and the rocprofv3 kernel statistics of the headline configuration. Usage: gpu_round_check.sh <tag>   ->  gpurun_out/<tag>_*
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --keep-pmc gpurun_out/${TAG}_pmc --write-traffic > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/${TAG}_bench.json
cp profiles/bench_full.json gpurun_out/${TAG}_bench_full.json 2>/dev/null; cp profiles/traffic.json gpurun_out/${TAG}_traffic.json 2>/dev/null
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --no-extra-configs --no-scene --no-cpu-baseline --pmc off --no-comm --full-json /tmp/x.json < /dev/null > $R/gpurun_out/${TAG}_stats_bench.json 2> $R/gpurun_out/${TAG}_stats_bench.err
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv && python3 - "$f" <<'PY' | tee $R/gpurun_out/${TAG}_kernel_stats.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.05 or "k_" in r["Name"]:
        print("%-90s calls %5s avg %9.2f us min %9.2f max %9.2f  %5.1f%%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
rm -rf $R/gpurun_out/prof_$TAG
cd $R
# the N > 1 code path of the bench on one device (gloo, both ranks on GPU 0): launcher, slab sharding, the counter all-reduce, the line
# (round 6: the alive totals go through the product's own collective - hnb_comm_create_rank + hnb_comm_allreduce_alive; two ranks on ONE device need the stand-in library, the real librccl refuses a device twice)
timeout 300 python bench.py --gpus 2 --backend gloo --force-device 0 --comm-lib tests/fake_rccl/libfake_rccl.so --steps 10 --windows 5 --pmc off > gpurun_out/${TAG}_bench_n2_dryrun.json 2> gpurun_out/${TAG}_bench_n2_dryrun.err; tail -c 400 gpurun_out/${TAG}_bench_n2_dryrun.json
[ -z "$FULL" ] && exit 0   # (FULL=1: + fuzz, SQ counters, instruction-cache counters)
# a short differential fuzz of the build against the oracle (random assets specialised / typed / in scenes of 8 interpreted)
L=gpurun_out/${TAG}_fuzz.log; : > $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --seeds 7000:7100 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --typed --seeds 7400:7480 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --capacity 9000 --frames 40 --seeds 7600:7640 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 0 --scene 8 --seeds 9000:9160 2>&1 | tail -2 >> $L
# (round 6) ... and every eligible random program with EVERY init pass slot-major (HNB_OPT_SLOT_INIT = 2: k_spawn_mark + k_init_slots), specialised and interpreted
echo "HNB_CTX_OPTIONS=slot_init=2:" >> $L
HNB_CTX_OPTIONS=slot_init=2 timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --seeds 12000:12100 2>&1 | tail -2 >> $L
HNB_CTX_OPTIONS=slot_init=2 timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --capacity 9000 --frames 40 --seeds 12600:12640 2>&1 | tail -2 >> $L
HNB_CTX_OPTIONS=slot_init=2 timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 0 --seeds 12800:12860 2>&1 | tail -2 >> $L
cat $L
# SQ counters of the init / update kernels (is a kernel bound by VALU issue, memory, or instruction issue?) and what the instruction cache says about the churn init
bash tools/valu_pmc.sh 2>&1 | tail -28 > gpurun_out/${TAG}_valu_counters.txt; tail -6 gpurun_out/${TAG}_valu_counters.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQC_" | head -40 > $R/gpurun_out/${TAG}_counter_names.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES --output-format csv -d $R/gpurun_out/icache_$TAG -- python $R/bench.py --config c2_mixed --no-cpu-baseline --no-extra-configs --no-parity --pmc off --no-scene --no-comm --steps 10 --windows 3 --full-json /tmp/y.json > $R/gpurun_out/${TAG}_icache.log 2>&1
python3 - <<'PY' | tee $R/gpurun_out/${TAG}_icache.txt
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
fs = glob.glob(O + "/icache_*/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no icache counters collected")
else:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "hnb::" in k: per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in per.items():
        print(k, {n: sorted(v)[len(v) // 2] for n, v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
rm -rf $R/gpurun_out/icache_$TAG
