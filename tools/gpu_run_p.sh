#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; tail -c 600 gpurun_out/r02p_bench.json
bash tools/prof_bench.sh r02p 2>&1 | tail -26
