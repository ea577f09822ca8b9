#!/bin/bash
# round 6: is a small frame waiting for the device or for the host? (hnb_program_kernel_info: frames in which hnb_simulate found the ring slot still in use)
cd "$GRAFT_REPO_ROOT" || exit 1
export HNB_JIT_CACHE=$GRAFT_REPO_ROOT/bevy_hanabi_amd/jit_cache
for i in 1 2; do timeout 600 python tools/scene_bench.py 1 600 2>/dev/null | grep "frames\b\|waited" ; done | tee gpurun_out/r06z_waits.log
python - <<'PY' | tee -a gpurun_out/r06z_waits.log
import time, sys
sys.path.insert(0, "tools")
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
import bench
for cap in (1 << 22, 1 << 16):
    ctx = bh.Context(0)
    asset = effects.ribbon(cap)
    prog = ctx.create_program(bh.lower(asset)); fx = prog.create_effect()
    per = cap // 90
    for f in range(400):
        ctx.frame_begin(1 / 60, f / 60); fx.set_frame(per, 1234 + f); ctx.simulate()
    ctx.synchronize()
    n = 3000; t0 = time.perf_counter()
    for f in range(400, 400 + n):
        ctx.frame_begin(1 / 60, f / 60); fx.set_frame(per, 1234 + f); ctx.simulate()
    ctx.synchronize()
    print("ribbon capacity %d: %.2f us per frame;" % (cap, (time.perf_counter() - t0) / n * 1e6), [l for l in prog.kernel_info().split("\n") if l.startswith("hnb_simulate waited")][0])
    ctx.close()
PY
