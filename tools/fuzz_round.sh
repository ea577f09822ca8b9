#!/bin/bash
# Development tool (GPU box): differential fuzz sweep of the round's build - random assets (float and typed generators, specialised and
# interpreted kernels, small and chunk-sized capacities), random worlds under both list orders, random systems of linked effects.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; L=gpurun_out/${1:-r02}_fuzz_sweep.log; : > $L
timeout 500 python tests/fuzz_sweep.py --backend gpu --jit 1 --seeds 7000:7250 2>&1 | tail -3 >> $L
timeout 500 python tests/fuzz_sweep.py --backend gpu --jit 0 --seeds 7250:7400 2>&1 | tail -3 >> $L
timeout 500 python tests/fuzz_sweep.py --backend gpu --jit 1 --typed --seeds 7400:7600 2>&1 | tail -3 >> $L
timeout 500 python tests/fuzz_sweep.py --backend gpu --jit 1 --capacity 9000 --frames 40 --seeds 7600:7700 2>&1 | tail -3 >> $L
timeout 500 python tests/fuzz_sweep.py --backend gpu --jit 1 --abstract --seeds 7700:7850 2>&1 | tail -3 >> $L
timeout 500 python tests/fuzz_sweep.py --backend gpu --jit 1 --abstract --typed --seeds 7850:8000 2>&1 | tail -3 >> $L
timeout 400 python tests/fuzz_sweep.py --backend gpu --jit 0 --scene 8 --seeds 9000:9400 2>&1 | tail -3 >> $L
SOAK_LO=${SOAK_LO:-300} SOAK_HI=${SOAK_HI:-340} timeout 900 python tools/soak_fuzz.py 2>&1 | grep -E "MISMATCH|soak:" >> $L
cat $L
