# experiment: the frame's kernels read their parameter block from the pinned HOST buffer (no copy, no host wait); upload_mode 1 = coherent mapping, 2 = non-coherent
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05o_ab_zero_copy.log; : > $L
for M in 2 1; do
  echo "=== parity tests under upload_mode=$M" >> $L
  HNB_CTX_OPTIONS=upload_mode=$M timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_scene_merge.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -4 >> $L
done
OPT=upload_mode=2 CONFIGS="c5 c2 c2_mixed c3 c4" timeout 900 bash tools/ab_option.sh gpurun_out/r05o_ab.tmp > /dev/null; cat gpurun_out/r05o_ab.tmp >> $L
OPT=upload_mode=1 CONFIGS="c5 c2" timeout 400 bash tools/ab_option.sh gpurun_out/r05o_ab.tmp > /dev/null; cat gpurun_out/r05o_ab.tmp >> $L; rm -f gpurun_out/r05o_ab.tmp
for rep in 1 2; do
  for O in "" "upload_mode=2" "upload_mode=1"; do
    echo "=== scene ${O:-default} (rep $rep)" >> $L
    HNB_CTX_OPTIONS=$O timeout 200 python tools/scene_bench.py 1 1500 2>&1 | grep "frames:" >> $L
  done
done
echo "=== host split, c5, upload_mode=2" >> $L
HNB_CTX_OPTIONS=upload_mode=2 timeout 200 python tools/host_bound_probe.py c5 2000 2>&1 | grep "^c5" >> $L
cat $L
