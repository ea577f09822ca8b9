#!/bin/bash
# round 6: A/B of the frame's parameter upload - k_stage_copy on the simulation stream (HNB_OPT_STAGE_KERNEL 1) against hipMemcpyAsync + host wait (0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r06c_ab_stage_kernel.log
: > $out
for round in 1 2; do
  for sk in 1 0; do
    for cfg in c5 c2_mixed c2; do
      r=$(HNB_CTX_OPTIONS=stage_kernel=$sk timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['windows']['ms_per_step_min_median_max'])")
      echo "round $round stage_kernel=$sk $cfg: $r" | tee -a $out
    done
    r=$(HNB_CTX_OPTIONS=stage_kernel=$sk timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1)
    echo "round $round stage_kernel=$sk scene: $r" | tee -a $out
  done
done
