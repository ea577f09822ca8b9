#!/bin/bash
# round 6: k_emit_events with 2048 events per split instead of 16,384: the event tests, then c2_events A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
timeout 1500 python -m pytest tests -m gpu -x -q -k "event or firework or c2_events or reference_examples" 2>&1 | tail -3 | tee gpurun_out/r06ae_pytest.log
L=gpurun_out/r06ae_ab_event_splits.log; : > $L
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_split16k.so; do
    r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config c2_events --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['windows']['ms_per_step_min_median_max'])")
    echo "round $round ${lib:-default} c2_events: $r" | tee -a $L
done; done
bash tools/r06ad.sh 2>&1 | tail -22
