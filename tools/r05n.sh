# A/B on one box: who waits for the frame's parameter upload - the host (hipStreamSynchronize of the upload stream: 10 us per frame, r05m) or the simulation stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OPT=upload_wait=1 CONFIGS="c5 c2 c2_mixed c3" timeout 700 bash tools/ab_option.sh gpurun_out/r05n_ab_upload_wait.log
for rep in 1 2; do
  for O in "" "upload_wait=1"; do
    echo "=== scene ${O:-default} (rep $rep)" >> gpurun_out/r05n_ab_upload_wait.log
    HNB_CTX_OPTIONS=$O timeout 200 python tools/scene_bench.py 1 1500 2>&1 | tail -3 >> gpurun_out/r05n_ab_upload_wait.log
  done
done
echo "=== host split, c5" >> gpurun_out/r05n_ab_upload_wait.log
HNB_CTX_OPTIONS=upload_wait=1 timeout 200 python tools/host_bound_probe.py c5 2000 2>&1 | grep "^c5" >> gpurun_out/r05n_ab_upload_wait.log
cat gpurun_out/r05n_ab_upload_wait.log
