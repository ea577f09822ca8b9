#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
tail -c 6000 gpurun_out/r02b_bench.json; tail -3 gpurun_out/r02b_bench.err
bash tools/prof_bench.sh r02b 2>&1 | tail -40
timeout 300 ./tools/layout_probe 16777216 4 20 > gpurun_out/r02b_layout_probe.log 2>&1; tail -12 gpurun_out/r02b_layout_probe.log
export TMPDIR=/tmp; cd /tmp; rocprofv3 -L > $R/gpurun_out/r02b_counters.txt 2>&1; wc -l $R/gpurun_out/r02b_counters.txt
