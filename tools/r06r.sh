#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_leannt.so; do
r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config c5 --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])")
echo "round $round ${lib:-default} c5: $r" | tee -a gpurun_out/r06r_ab_lean_nt.log
done; done
