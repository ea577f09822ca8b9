#!/bin/bash
# A/B: generic update split (default library) against one workgroup per chunk (HNB_GENERIC_SPLIT_MAX=0 build)
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
for round in 1 2; do for lib in "" tools/variants/libhanabi_nosplit.so; do
HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python - <<'PY' | tee -a gpurun_out/r06j_ab_generic_split.log
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed
out = []
for cap in (32768, 262144):
    ctx = bh.Context(0)
    a = effects.firework_rocket(cap, 5, 1000)
    a.spawner = bh.SpawnerSettings.rate(cap / 2.0)
    prog = ctx.create_program(bh.lower(a)); fx = prog.create_effect()
    sp, rng = bh.EffectSpawner(a.spawner), bh.Pcg32()
    for f in range(300):
        ctx.frame_begin(1 / 60, f / 60); fx.set_frame(sp.tick(1 / 60, rng), frame_seed(f)); ctx.simulate()
    ctx.synchronize(); t0 = time.perf_counter()
    for f in range(300, 900):
        ctx.frame_begin(1 / 60, f / 60); fx.set_frame(sp.tick(1 / 60, rng), frame_seed(f)); ctx.simulate()
    ctx.synchronize()
    out.append(f"capacity {cap}: {(time.perf_counter() - t0) / 600 * 1e6:.1f} us per frame")
    ctx.close()
print(("one workgroup per chunk: " if os.environ.get("HNB_LIB") else "one workgroup per 256 slots: ") + "; ".join(out) + " (lone rocket effect of firework.rs, generic update)")
PY
r=$(HNB_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 600 python bench.py --config c2_events --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
echo "   c2_events ${lib:-default}: $r" | tee -a gpurun_out/r06j_ab_generic_split.log
done; done
