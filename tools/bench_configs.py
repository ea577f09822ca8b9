"""Development tool: throughput of the other BASELINE.json configs (C3 force_field, C4 instancing,
C5 ribbon churn) and of the firework die-off (compaction-heavy frames). Not the headline bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import frame_seed, DT

which = sys.argv[1:] or ["c2die", "c3", "c4", "c5"]
ctx = bh.Context(0)

def run(label, prog, fxs, frames, spawn_fn, bytes_per_update, warm=3):
    f = 0
    def step(timed):
        nonlocal f
        ctx.frame_begin(DT, f * DT)
        if len(fxs) > 16:   # batches: one call for all instances
            prog.set_frames([spawn_fn(f, i) for i in range(len(fxs))], [(f * 4099 + i) * 2654435761 & 0xffffffff for i in range(len(fxs))])
        else:
            for i, fx in enumerate(fxs):
                fx.set_frame(spawn_fn(f, i), frame_seed(f * 4099 + i))
        ctx.simulate(); f += 1
    for _ in range(warm): step(False)
    ctx.synchronize()
    a0 = sum(fx.alive_count() for fx in fxs) if len(fxs) <= 8 else None
    ctx.enable_kernel_timing(1)
    t0 = time.perf_counter()
    for _ in range(frames): step(True)
    ctx.synchronize()
    el = time.perf_counter() - t0
    tm = ctx.kernel_timing(); ctx.enable_kernel_timing(0)
    a1 = sum(fx.alive_count() for fx in fxs[:8])
    print(f"{label}: {el/frames*1e3:.3f} ms/frame wall (events on), update {tm['update_ms_avg']:.4f} ms, compact {tm['compact_ms_avg']:.4f} ms, init {tm['init_ms_avg']:.4f} ms; alive(first 8 inst) {a1}", flush=True)
    return tm

if "c2die" in which:
    cap = 1 << 24
    prog = ctx.create_program(bh.lower(effects.firework_trails(cap))); fx = prog.create_effect()
    ctx.frame_begin(DT, 0); fx.set_frame(cap, frame_seed(0)); ctx.simulate()
    f = 1
    for _ in range(46):
        ctx.frame_begin(DT, f * DT); fx.set_frame(0, frame_seed(f)); ctx.simulate(); f += 1
    ctx.synchronize()
    ctx.enable_kernel_timing(1)
    rows = []
    for _ in range(26):
        a0 = fx.alive_count()
        ctx.enable_kernel_timing(1)
        ctx.frame_begin(DT, f * DT); fx.set_frame(0, frame_seed(f)); ctx.simulate(); f += 1
        tm = ctx.kernel_timing(); a1 = fx.alive_count()
        rows.append((a0, a0 - a1, tm["update_ms_avg"], tm["compact_ms_avg"]))
    ctx.enable_kernel_timing(0)
    print("firework die-off: alive_in, died, update_ms, compact_ms")
    for r in rows: print("   %9d %9d  %.4f  %.4f   -> %.1f GB/s @68B" % (r[0], r[1], r[2], r[3], r[0] * 68 / ((r[2] + r[3]) * 1e-3) / 1e9 if r[0] else 0))
    prog.destroy()

if "c3" in which:
    cap = 1 << 23
    prog = ctx.create_program(bh.lower(effects.force_field(cap))); fx = prog.create_effect()
    tm = run("C3 force_field 8M", prog, [fx], 60, lambda f, i: cap if f == 0 else 0, 68, warm=3)
    a = fx.alive_count()
    print(f"   -> {a * 68 / (tm['update_ms_avg'] * 1e-3) / 1e9:.1f} GB/s @68B on {a} alive")
    prog.destroy()

if "c5" in which:
    cap = 1 << 22
    asset = effects.ribbon(cap)
    prog = ctx.create_program(bh.lower(asset)); fx = prog.create_effect()
    sp = bh.EffectSpawner(asset.spawner); rng = bh.Pcg32()
    counts = [sp.tick(DT, rng) for _ in range(400)]
    tm = run("C5 ribbon 4M churn", prog, [fx], 200, lambda f, i: counts[f], 20, warm=120)
    a = fx.alive_count()
    print(f"   -> steady alive {a}; update+compact {(tm['update_ms_avg'] + tm['compact_ms_avg']) * 1e3:.1f} us, {a * 20 / ((tm['update_ms_avg'] + tm['compact_ms_avg']) * 1e-3) / 1e9:.1f} GB/s @20B")
    prog.destroy()

if "c4" in which:
    cap, n_inst = 65536, int(os.environ.get("C4_INST", "4096"))
    asset = effects.instancing(cap)
    prog = ctx.create_program(bh.lower(asset))
    t0 = time.perf_counter()
    fxs = [prog.create_effect() for _ in range(n_inst)]
    print(f"C4: created {n_inst} instances in {time.perf_counter() - t0:.1f} s")
    tm = run(f"C4 instancing {n_inst}x65536 (burst, all alive)", prog, fxs, 20, lambda f, i: cap if f == 0 else 0, 68, warm=3)
    tot = cap * n_inst
    print(f"   -> {tot * 68 / (tm['update_ms_avg'] * 1e-3) / 1e9:.1f} GB/s @68B, {tot / ((tm['update_ms_avg'] + tm['compact_ms_avg']) * 1e-3):.3e} updates/s (kernels)")
    prog.destroy()
if "churn" in which:
    # steady spawn/kill churn over many instances: every instance spawns and loses ~3 % of its particles per frame
    A = bh.Attribute
    cap, n_inst = int(os.environ.get("CHURN_CAP", "65536")), int(os.environ.get("C4_INST", "1024"))
    w = bh.ExprWriter()
    mods = [bh.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), bh.ShapeDimension.Volume),
            bh.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(0.3).uniform(w.lit(0.6)).expr())]
    asset = bh.EffectAsset(cap, bh.SpawnerSettings.rate(cap * 2.0), w.finish())
    for m in mods: asset.init(m)
    asset.update(bh.AccelModifier(w.lit((0.0, -3.0, 0.0)).expr())) if False else None
    per_frame = int(cap * 2.0 / 60)
    for order in ("spawn", "slot"):
        ctx.set_list_order(order)
        prog = ctx.create_program(bh.lower(asset))
        fxs = [prog.create_effect() for _ in range(n_inst)]
        tm = run(f"churn {n_inst}x{cap}, {per_frame} spawns/instance/frame, list order = {order}", prog, fxs, 30, lambda f, i: per_frame, 56, warm=45)
        alive = sum(fx.alive_count() for fx in fxs[:8]) / 8
        tot = alive * n_inst
        k = tm['update_ms_avg'] + tm['compact_ms_avg'] + tm['init_ms_avg']
        print(f"   -> steady alive/instance {alive:.0f} ({alive / cap:.2f} full); init+update+compact(+reorder) {k * 1e3:.0f} us per frame = {tot / (k * 1e-3):.3e} updates/s; update alone {tot * 56 / (tm['update_ms_avg'] * 1e-3) / 1e9:.0f} GB/s @56B")
        prog.destroy()
    ctx.set_list_order("spawn")

if "events" in which:
    # 1M-particle parent, every dying particle spawns 16 children (OnDie) into a 16M-particle child effect
    A = bh.Attribute
    w = bh.ExprWriter()
    pinit = [bh.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(5.0).expr(), bh.ShapeDimension.Volume),
             bh.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(3.0).expr()),
             bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(0.2).uniform(w.lit(0.6)).expr())]
    pupd = [bh.AccelModifier(w.lit((0.0, -9.0, 0.0)).expr()), bh.EmitSpawnEventModifier(bh.EventEmitCondition.OnDie, w.lit(bh.Value.u32(16)).expr(), 0)]
    pcap, ccap = 1 << 20, 1 << 24
    parent = bh.EffectAsset(pcap, bh.SpawnerSettings.once(float(pcap)), w.finish())
    for m in pinit: parent.init(m)
    for m in pupd: parent.update(m)
    w = bh.ExprWriter()
    cinit = [bh.InheritAttributeModifier(A.POSITION), bh.SetAttributeModifier(A.VELOCITY, ((w.rand(bh.VectorType.VEC3F) * w.lit(2.0) - w.lit(1.0)).normalized() * w.lit(4.0)).expr()),
             bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(0.5).expr())]
    child = bh.EffectAsset(ccap, bh.SpawnerSettings(), w.finish())
    for m in cinit: child.init(m)
    child.update(bh.LinearDragModifier(w.lit(1.0).expr())) if False else None
    pp, cp = ctx.create_program(bh.lower(parent)), ctx.create_program(bh.lower(child))
    pf, cf = pp.create_effect(), cp.create_effect()
    cf.set_parent(pf, 0, 1 << 24)
    print(pp.kernel_info().split("\n")[0], "|", cp.kernel_info().split("\n")[0])
    for f in range(45):
        ctx.enable_kernel_timing(1)
        ctx.frame_begin(DT, f * DT); pf.set_frame(pcap if f == 0 else 0, frame_seed(f)); cf.set_frame(0, frame_seed(1000 + f)); ctx.simulate()
        tm = ctx.kernel_timing()
        if f % 4 == 1 or f < 3:
            print(f"  frame {f:2d}: parent alive {pf.alive_count():8d} child alive {cf.alive_count():9d} | per-program avg: update {tm['update_ms_avg']*1e3:7.1f} us, emit+compact {tm['compact_ms_avg']*1e3:7.1f} us, init {tm['init_ms_avg']*1e3:7.1f} us")
    ctx.enable_kernel_timing(0)
    pp.destroy(); cp.destroy()

if "generic" in which:
    # a non-streamable update stack (expression-driven SetAttribute in update): k_update_generic
    cap = 1 << 24
    h = bh
    A = bh.Attribute
    w = h.ExprWriter()
    init = [h.SetAttributeModifier(A.POSITION, w.lit((0.0, 0.0, 0.0)).expr()),
            h.SetAttributeModifier(A.VELOCITY, ((w.rand(h.VectorType.VEC3F) * w.lit(2.0) - w.lit(1.0)).normalized() * w.lit(40.0).uniform(w.lit(60.0))).expr()),
            h.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), h.SetAttributeModifier(A.LIFETIME, w.lit(0.8).uniform(w.lit(1.2)).expr())]
    ratio = w.attr(A.AGE) / w.attr(A.LIFETIME)
    upd = [h.LinearDragModifier(w.lit(4.0).expr()), h.AccelModifier(w.lit((0.0, -16.0, 0.0)).expr()),
           h.SetAttributeModifier(A.SIZE, (w.lit(1.0) - ratio * ratio).expr())]
    asset = h.EffectAsset(cap, h.SpawnerSettings.once(float(cap)), w.finish())
    for m in init: asset.init(m)
    for m in upd: asset.update(m)
    for jit in ("1", "0"):
        os.environ["HNB_JIT"] = jit
        prog = ctx.create_program(bh.lower(asset)); fx = prog.create_effect()
        print(prog.kernel_info().split("\n")[0])
        tm = run(f"generic update 16M (HNB_JIT={jit})", prog, [fx], 20, lambda f, i: cap if f == 0 else 0, 76, warm=3)
        print(f"   -> {cap * 76 / (tm['update_ms_avg'] * 1e-3) / 1e9:.1f} GB/s @76B (68 + size write 4 + ... )")
        prog.destroy()
ctx.close()
