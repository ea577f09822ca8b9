# A/B on one box: the ahead-of-time kernels launched through hipModuleLaunchKernel (default of this build) against <<<>>> (module_launch=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05r_ab_module_launch.log; : > $L
echo "=== parity tests (module launches)" >> $L
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_scene_merge.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2 >> $L
OPT=module_launch=0 CONFIGS="c5 c2_mixed" timeout 600 bash tools/ab_option.sh gpurun_out/r05r_ab.tmp > /dev/null; cat gpurun_out/r05r_ab.tmp >> $L; rm -f gpurun_out/r05r_ab.tmp
for rep in 1 2 3; do
  for O in "" "module_launch=0"; do
    echo "=== scene ${O:-default} (rep $rep)" >> $L
    HNB_CTX_OPTIONS=$O timeout 200 python tools/scene_bench.py 1 1500 2>&1 | grep "frames:" >> $L
  done
done
for O in "" "module_launch=0"; do
  echo "=== host split, c5 ${O:-default}" >> $L
  HNB_CTX_OPTIONS=$O timeout 200 python tools/host_bound_probe.py c5 2000 2>&1 | grep "^c5" >> $L
done
cat $L
