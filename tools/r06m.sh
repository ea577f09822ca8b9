#!/bin/bash
# A/B: the no-death bound published by one wave without barriers / waits (default library) against the whole workgroup 0 with two barriers around the host-mapped store
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
out=gpurun_out/r06m_ab_publish.log; : > $out
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_oldpublish.so; do for cfg in c5 c2 c2_mixed c4; do
r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
echo "round $round ${lib:-new} $cfg: $r" | tee -a $out
done; done; done
