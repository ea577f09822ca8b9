#!/bin/bash
# round 6: wave reductions over DPP instead of __shfl_xor butterflies: the GPU suite, then same-box A/B (ahead-of-time kernels only: the variant shares the jit cache)
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r06ac_pytest.log
L=gpurun_out/r06ac_ab_dpp.log; : > $L
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_shfl.so; do
  for cfg in c5 c2 c2_mixed c3; do
    r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
    echo "round $round ${lib:-default} $cfg: $r" | tee -a $L
  done
done; done
for i in 1 2 3; do timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1; done | tee -a $L
