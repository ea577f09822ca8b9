cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_device_view.py tests/test_verification.py tests/test_scene_merge.py tests/test_reference_examples.py -m gpu -q -k "ribbon or c5 or ring or view or verification or gate or scene or example or healthy or broken" --timeout 600 -rf -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r05h_pytest.log; tail -12 gpurun_out/r05h_pytest.log
OPT=ring_lists=0 CONFIGS="c5" timeout 300 bash tools/ab_option.sh gpurun_out/r05h_ab_ring.log
