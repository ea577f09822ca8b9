#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r02l_bench.json 2> gpurun_out/r02l_bench.err; tail -c 2500 gpurun_out/r02l_bench.json
bash tools/prof_bench.sh r02l 2>&1 | tail -28
timeout 300 python -m pytest tests/test_ron.py tests/test_host_capi.py tests/test_examples.py -m gpu -q 2>&1 | tail -3
