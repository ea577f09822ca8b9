#!/bin/bash
# round 6, first GPU pass: the new tests, then re-burst timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_device_view.py tests/test_verification.py -m gpu -x -q -k "slot_major or slot_init or zoo_through or large_spawns or auto_mode or test_hook or broken_proof or many_chunks" > gpurun_out/r06a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06a_pytest.log
tail -5 gpurun_out/r06a_pytest.log
timeout 600 python tools/r06_reburst.py > gpurun_out/r06a_reburst.log 2>&1
cat gpurun_out/r06a_reburst.log | tail -12
timeout 1200 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "emptied_ring or c2_firework_full or c2_mixed_full or ring" > gpurun_out/r06a_pytest2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06a_pytest2.log
tail -5 gpurun_out/r06a_pytest2.log
