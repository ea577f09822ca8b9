#!/bin/bash
# round 6: HNB_OPT_DIRECT_UPLOAD (the host writes the frame's parameter block into fine-grained device memory): its test, then same-box A/B on every configuration kind
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
timeout 900 python -m pytest tests -m gpu -x -q -k "frame_parameters or c5 or ribbon or scene or batched or lifecycle" 2>&1 | tail -3 | tee gpurun_out/r06x_pytest.log
L=gpurun_out/r06x_ab_direct_upload.log; : > $L
for round in 1 2 3; do for opt in "" direct_upload=0; do
  for cfg in c5 c2 c2_mixed c4 c3; do
    r=$(HNB_CTX_OPTIONS=$opt timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
    echo "round $round ${opt:-default} $cfg: $r" | tee -a $L
  done
  r=$(HNB_CTX_OPTIONS=$opt timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1)
  echo "round $round ${opt:-default} scene: $r" | tee -a $L
done; done
