# differential fuzzing of the final tree on the GPU (the same four sweeps tools/gpu_round_check.sh runs with FULL=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05p_fuzz.log; : > $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --seeds 7000:7100 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --typed --seeds 7400:7480 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --capacity 9000 --frames 40 --seeds 7600:7640 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 0 --scene 8 --seeds 9000:9160 2>&1 | tail -2 >> $L
timeout 300 python tests/fuzz_sweep.py --backend gpu --jit 1 --seeds 11000:11100 2>&1 | tail -2 >> $L
cat $L
