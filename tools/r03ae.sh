#!/bin/bash
# differential fuzz of the merged launches: random assets, several per context
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; L=gpurun_out/${1:-r03ae}_scene_fuzz.log; : > $L
timeout 400 python tests/fuzz_sweep.py --backend gpu --jit 0 --scene 8 --seeds 9000:9400 2>&1 | tail -4 >> $L
timeout 400 python tests/fuzz_sweep.py --backend gpu --jit 0 --scene 5 --capacity 5000 --frames 24 --seeds 9400:9500 2>&1 | tail -4 >> $L
timeout 400 python tests/fuzz_sweep.py --backend gpu --jit 1 --scene 6 --seeds 9500:9560 2>&1 | tail -4 >> $L
cat $L
