#!/bin/bash
# round 6: four workgroups per chunk for the streaming members of the merged launches (SlotArgs::quarters): the whole GPU suite, then the scene A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r06aa_pytest.log
L=gpurun_out/r06aa_ab_quarters.log; : > $L
for round in 1 2 3; do for lib in "" tools/variants/libhanabi_noquarters.so; do
  r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames:" | tail -1)
  echo "round $round ${lib:-default} scene: $r" | tee -a $L
done; done
bash tools/prof_scene.sh 2>&1 | tail -24 | tee -a $L
