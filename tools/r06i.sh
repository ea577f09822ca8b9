#!/bin/bash
# round 6: generic-update programs of a few chunks launch one workgroup per 256 slots: the GPU suite's event / zoo / fuzz legs, then c2_events and a lone rocket effect
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -x -q -k "events or zoo or fuzz or slot_order or generic or reference_examples or scene" 2>&1 | tail -3
for round in 1 2; do
  r=$(timeout 600 python bench.py --config c2_events --no-parity --pmc off --no-cpu-baseline --no-extra-configs --no-scene --no-comm --windows 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['windows']['ms_per_step_min_median_max'])")
  echo "round $round c2_events: $r" | tee -a gpurun_out/r06i_generic_split.log
done
python - <<'PY' | tee -a gpurun_out/r06i_generic_split.log
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed
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
    print(f"lone rocket effect (generic update, no children listening), capacity {cap}: {(time.perf_counter() - t0) / 600 * 1e6:.1f} us per frame, alive {fx.alive_count()}")
    ctx.close()
PY
