cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OPT=upload_inline=1 CONFIGS="c5 c2_mixed" timeout 600 bash tools/ab_option.sh gpurun_out/r05j_ab_upload.log
for o in "" "upload_inline=1"; do HNB_CTX_OPTIONS=$o timeout 200 python tools/scene_bench.py 1 600 2>&1 | grep -v amdgpu.ids | tail -2 | head -1 >> gpurun_out/r05j_ab_upload.log; done
tail -3 gpurun_out/r05j_ab_upload.log
