#!/bin/bash
# Development tool (GPU box): first measurement pass of a round: GPU tests, default bench, N=2 dry run on one GPU, layout probe.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r02a_pytest.log
timeout 600 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
timeout 300 python bench.py --gpus 2 --backend gloo --force-device 0 --steps 10 > gpurun_out/r02a_bench_n2.json 2> gpurun_out/r02a_bench_n2.err
timeout 300 python bench.py --gpus 2 --backend gloo --force-device 0 --steps 10 --config c4 --instances 64 > gpurun_out/r02a_bench_n2_c4.json 2> gpurun_out/r02a_bench_n2_c4.err
timeout 300 ./tools/layout_probe 16777216 8 20 > gpurun_out/r02a_layout_probe.log 2>&1
tail -5 gpurun_out/r02a_pytest.log; cat gpurun_out/r02a_bench.json | cut -c1-1500; tail -3 gpurun_out/r02a_bench.err; cat gpurun_out/r02a_bench_n2.json | cut -c1-600; tail -3 gpurun_out/r02a_bench_n2.err; cat gpurun_out/r02a_layout_probe.log
