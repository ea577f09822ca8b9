#!/bin/bash
# round 6: the head of every init / list kernel requested in ONE round (DevFrameInst's first 32 bytes, the slab, the counters, the casualty counter; single-instance
# programs skip the instance search): the GPU suite, then same-box A/B against the previous commit (its own library AND its own jit cache)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 env HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06ag_pytest.log
L=gpurun_out/r06ag_ab_heads.log; : > $L
run() {  # $1 = "" (this tree) or prev
  if [ -z "$1" ]; then export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache; unset HNB_LIB; else export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/tools/variants/jit_prev HNB_LIB=$GRAFT_REPO_ROOT/tools/variants/libhanabi_prev.so; fi
}
for round in 1 2 3; do for v in "" prev; do
  run "$v"
  for cfg in c5 c3 c2_mixed c2; do
    r=$(timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
    echo "round $round ${v:-this} $cfg: $r" | tee -a $L
  done
  r=$(timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1)
  echo "round $round ${v:-this} scene: $r" | tee -a $L
done; done
