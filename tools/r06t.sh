#!/bin/bash
# round 6: k_update_slots_stream_age (the age-only update as its own kernel) and "the first workgroup publishes": ribbon / ring / scene tests, then same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
timeout 1500 python -m pytest tests -m gpu -x -q -k "ribbon or c5 or ring or scene or reference_examples or skip or horizon or verification or bench_gate or timed or frozen or visible" 2>&1 | tail -3 | tee gpurun_out/r06t_pytest.log
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_pub0.so tools/variants/libhanabi_base.so; do
r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config c5 --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['parity']['ok'])")
echo "round $round ${lib:-default} c5: $r" | tee -a gpurun_out/r06t_ab_age_kernel.log
done; done
for round in 1 2; do for lib in "" tools/variants/libhanabi_pub0.so; do for cfg in c2 c2_mixed c4; do
r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config $cfg --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])")
echo "round $round ${lib:-default} $cfg: $r" | tee -a gpurun_out/r06t_ab_age_kernel.log
done; done; done
