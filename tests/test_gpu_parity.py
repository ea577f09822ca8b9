"""GPU parity (the product path: lowering -> C ABI -> HIP kernels) against the oracle.

Bit-exact on every attribute, counter and list: the arithmetic of both sides is pinned by
the hanabi-math definition, so the 1e-5 relative tolerance BASELINE.json allows on
position/velocity is met with zero difference.
"""
import struct

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from helpers import A, Frame, GpuRunner, OracleRunner, assert_same_state, frame_seed, run_script, translation
from test_lowering_cpu import ZOO, burst_then_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = bh.Context(0)
    c.set_option("age_cohort", 1)   # HNB_AGE_COHORT_LEAN: the headless configuration the bench times (the default, AUTO, keeps per-particle ages for assets whose
    #                                 render modifiers read AGE - firework.rs, instancing.rs; tests/test_device_view.py covers that choice)
    yield c
    c.close()


def test_native_library_is_the_one_running(ctx):
    import os
    maps = open("/proc/self/maps").read()
    assert "libhanabi_amd.so" in maps and os.path.dirname(bh.__file__) in maps


def test_c1_single_particle(ctx):
    asset = effects.single_particle(16)
    st = run_script(GpuRunner(asset, ctx=ctx), [Frame(1 / 60, 16, 0)], OracleRunner(asset))
    assert st["counters"]["alive_count"] == 16
    assert (st["attrs"]["position"].view(np.float32) == np.array([0.1, 0.2, 0.3], dtype=np.float32)).all()


def test_c2_firework_life_cycle(ctx):
    cap = 20000  # 5 chunks: exercises the cross-chunk prefix of k_compact
    asset = effects.firework_trails(cap)
    frames = burst_then_run(cap, 80) + [Frame(1 / 60, 7777, frame_seed(100))] + [Frame(1 / 60, 0, frame_seed(101 + f)) for f in range(30)]
    st = run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=6)
    assert 0 < st["counters"]["alive_count"] <= 7777


def test_c3_force_field(ctx):
    cap = 30000
    asset = effects.force_field(cap)
    frames = burst_then_run(cap, 100)
    frames[40].props = {"repulsor_position": (0.1, 0.2, 0.0), "repulsor_accel": -25.0}
    run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=10)


def test_c4_instancing_batch_of_instances(ctx):
    """Several instances of one program: one init + one update + one compact launch for the whole batch."""
    cap, n_inst = 9000, 5
    asset = effects.instancing(cap, rate=cap / 0.25)
    blob = bh.lower(asset)
    prog = ctx.create_program(blob)
    fxs = [prog.create_effect() for _ in range(n_inst)]
    oracles = [OracleRunner(asset) for _ in range(n_inst)]
    spawners = [bh.EffectSpawner(asset.spawner) for _ in range(n_inst)]
    rng = bh.Pcg32()
    for f in range(40):
        ctx.frame_begin(1 / 60, f / 60)
        for i, (fx, orc, sp) in enumerate(zip(fxs, oracles, spawners)):
            n = sp.tick(1 / 60, rng) if (f + i) % 3 else 0
            xf = translation(10.0 * i, -5.0, 0.5 * i)
            seed = frame_seed(f * 16 + i)
            fx.set_frame(n, seed, xf)
            orc.step(Frame(1 / 60, n, seed, xf, time=f / 60))
        ctx.simulate()
    for fx, orc in zip(fxs, oracles):
        ref = orc.state()
        np.testing.assert_array_equal(ref["alive"], fx.alive_list())
        np.testing.assert_array_equal(ref["dead"], fx.dead_list())
        assert ref["counters"]["alive_count"] == fx.alive_count()
        for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME):
            np.testing.assert_array_equal(ref["attrs"][a.name], fx.read_attr(a.id).view(np.uint32))
    prog.destroy()


def test_c5_ribbon_churn(ctx):
    cap = 12000
    asset = effects.ribbon(cap)
    sp = bh.EffectSpawner(asset.spawner)
    rng = bh.Pcg32()
    frames = []
    for f in range(200):
        t = f / 60.0
        frames.append(Frame(1 / 60, sp.tick(1 / 60, rng), frame_seed(f), translation(np.sin(t), np.cos(t), 0.0), time=t))
    st = run_script(GpuRunner(asset, ctx=ctx), frames, OracleRunner(asset), every=25)
    assert st["counters"]["dead_count"] > 0


@pytest.fixture(params=["jit", "interp"])
def kernels(request, monkeypatch):
    """Both ways the generic kernels can run a program: specialised at creation (hiprtc) or interpreted."""
    monkeypatch.setenv("HNB_JIT", "1" if request.param == "jit" else "0")
    return request.param


@pytest.mark.parametrize("name", sorted(ZOO))
def test_zoo(ctx, name, kernels):
    asset = ZOO[name]()
    cap = asset.capacity
    xf = np.array([0.0, -1.0, 0.0, 4.0, 1.0, 0.0, 0.0, -2.0, 0.0, 0.0, 1.0, 0.5], dtype=np.float32)
    frames = [Frame(1 / 60, cap // 2, frame_seed(0), xf)]
    for f in range(1, 60):
        frames.append(Frame(1 / 60 if f % 7 else 1 / 30, (cap // 9) if f % 11 == 0 else 0, frame_seed(f), xf, time=f / 60.0))
    g = GpuRunner(asset, ctx=ctx)
    info = g.prog.kernel_info()
    if kernels == "jit":
        assert "interp" not in info.split("\n")[0], info  # nothing left to the interpreter
    else:
        assert "jit" not in info, info
    run_script(g, frames, OracleRunner(asset), every=10)
    g.fx.destroy()
    g.prog.destroy()


# ---- slot-major init of large spawns (round 6; hnb_kernels.hip.h "slot-major init") -----------------------------------------------------
_OP_LDPC, _OP_LDPARENT = 5, 7                      # include/hanabi_amd.h: HnbOp
_PROG_HAS_RIBBONS, _PROG_READS_PARENT = 0x2, 0x4   # ... HNB_PROG_*


def _slot_init_frames(prog):
    line = [l for l in prog.kernel_info().split("\n") if l.startswith("slot-major init")]
    return int(line[0].split(":")[1].split()[0]) if line else 0


@pytest.mark.parametrize("name", sorted(ZOO))
def test_zoo_through_the_slot_major_init(name, kernels):
    """HNB_OPT_SLOT_INIT = 2: every eligible program of the zoo runs EVERY init pass slot-major (k_spawn_mark + k_init_slots, specialised and
    interpreted), through partial re-fills into recycled slots; programs that read PARTICLE_COUNTER and the ribbon programs keep the row-major
    pass. Same state as the oracle, bit for bit."""
    c = bh.Context(0)
    c.set_option("age_cohort", 1)
    c.set_option("slot_init", 2)
    asset = ZOO[name]()
    cap = asset.capacity
    xf = np.array([0.0, -1.0, 0.0, 4.0, 1.0, 0.0, 0.0, -2.0, 0.0, 0.0, 1.0, 0.5], dtype=np.float32)
    frames = [Frame(1 / 60, cap // 2, frame_seed(0), xf)]
    for f in range(1, 60):
        frames.append(Frame(1 / 60 if f % 7 else 1 / 30, (cap // 9) if f % 11 == 0 else (cap if f == 40 else 0), frame_seed(f), xf, time=f / 60.0))
    g = GpuRunner(asset, ctx=c)
    run_script(g, frames, OracleRunner(asset), every=10)
    # eligible (hanabi_amd.hip slot_init_eligible): no ribbons, no parent, the init stream reads neither PARTICLE_COUNTER nor a parent particle
    hdr = struct.unpack_from("<24I", g.blob)
    flags, init_len, init_off = hdr[4], hdr[9], hdr[17]
    ops = [struct.unpack_from("<I", g.blob, init_off + 8 * i)[0] & 0xFF for i in range(init_len)]
    eligible = not (flags & (_PROG_HAS_RIBBONS | _PROG_READS_PARENT)) and _OP_LDPC not in ops and _OP_LDPARENT not in ops
    assert _slot_init_frames(g.prog) == (7 if eligible else 0), (eligible, g.prog.kernel_info())   # frames 0, 11, 22, 33, 40, 44, 55 spawn
    c.close()


def test_large_spawns_slot_major_equal_the_row_major_init_and_the_oracle():
    """A burst, a die-off in large steps (the dead stack ends up in killing order), a partial re-fill (k_spawn_mark), a re-burst of `capacity`
    (every free slot, host-proven), a request just below the capacity that the device finds to be a complete re-fill, small spawns in between
    (row-major): the default context (slot-major from an eighth of the slots on), one with HNB_OPT_SLOT_INIT off and the oracle hold the same
    state after every frame; SpawnerSettings::burst(count, period) semantics (src/spawn.rs:472; pop order vfx_init.wgsl:141-143)."""
    cap = 300_007            # 74 chunks, the last one ragged
    asset = effects.firework_trails(cap)
    on, off = bh.Context(0), bh.Context(0)
    off.set_option("slot_init", 0)
    g_on, g_off, orc = GpuRunner(asset, ctx=on), GpuRunner(asset, ctx=off), OracleRunner(asset, omp=True)
    script = [(1 / 60, cap), (0.3, 0), (0.3, 0), (0.3, 0),            # burst; ages 0.32, 0.62, 0.92: the short-lived third dies
              (1 / 60, cap // 4),                                     # partial re-fill into the freed slots: marks
              (1 / 60, 1000), (0.3, 0),                               # a small spawn (row-major); everybody of the first burst dies
              (1 / 60, cap),                                          # re-burst: every free slot
              (0.5, 0), (0.5, 0),                                     # ... most of them die
              (1 / 60, cap - 5),                                      # less than the capacity, more than there are free slots: complete on the device
              (1 / 60, 0), (0.45, cap // 7), (0.45, cap // 3), (0.45, cap)]
    t = 0.0
    for f, (dt, spawn) in enumerate(script):
        fr = Frame(dt, spawn, frame_seed(f), time=t)
        t += dt
        for x in (g_on, g_off, orc):
            x.step(fr)
        ref = orc.state()
        assert_same_state(ref, g_on.state(), f"slot-major, frame {f}")
        assert_same_state(ref, g_off.state(), f"row-major, frame {f}")
        assert g_on.fx.check()["ok"] == 1
    assert _slot_init_frames(g_on.prog) == 7 and _slot_init_frames(g_off.prog) == 0, g_on.prog.kernel_info()
    assert g_on.fx.compare(g_off.fx)["equal"] == 1
    on.close(); off.close()


def _upload_frames(prog):
    line = [l for l in prog.kernel_info().split("\n") if l.startswith("frame parameters (context)")][0]
    words = line.split()
    return int(words[words.index("in") + 1]), int(words[words.index("copied") + 2])   # (written by the host, copied)


def test_frame_parameters_written_by_the_host_equal_the_copied_ones():
    """HNB_OPT_DIRECT_UPLOAD (round 6): the frame's parameter block written by the host into fine-grained device memory (the default where the device has a
    large BAR) against hipMemcpyAsync + a host wait, and against the oracle: a program with per-frame uniforms that change EVERY frame (translated
    emitter, a property, alternating ticks), several instances, the staging ring reused many times over (4 slots, 120 frames), and a switch of the
    option in the middle of a run (the slots are re-created between two frames). A stale or torn block would show as a wrong transform / tick / seed."""
    cap = 9000
    asset = effects.instancing(cap)
    d, c = bh.Context(0), bh.Context(0)
    c.set_option("direct_upload", 0)
    gd, gc, orc = GpuRunner(asset, ctx=d), GpuRunner(asset, ctx=c), OracleRunner(asset)
    for f in range(120):
        if f == 60:
            d.set_option("direct_upload", 0); c.set_option("direct_upload", 1)   # the other way round from here on
        fr = Frame(1 / 60 if f % 3 else 1 / 45, cap if f in (0, 70) else (f * 13) % 97, frame_seed(f), translation(0.01 * f, -0.02 * f, 0.5), time=f / 60.0)
        for x in (gd, gc, orc):
            x.step(fr)
        if f % 10 == 9 or f in (0, 60, 61):
            ref = orc.state()
            assert_same_state(ref, gd.state(), f"host-written first, frame {f}")
            assert_same_state(ref, gc.state(), f"copied first, frame {f}")
    wd, cd = _upload_frames(gd.prog)
    wc, cc = _upload_frames(gc.prog)
    if "no large BAR" in gd.prog.kernel_info():
        assert (wd, wc) == (0, 0) and cd == 120 and cc == 120
    else:
        assert (wd, cd) == (60, 60) and (wc, cc) == (60, 60), (gd.prog.kernel_info(), gc.prog.kernel_info())
    d.close(); c.close()


def test_slot_major_init_over_several_instances(ctx):
    """One launch for all instances of a program: a complete burst, a partial one, an instance that spawns nothing and a frozen one side by side
    (the frame qualifies as a whole: half of the program's slots spawn), death horizons and age cohorts in play; translated emitters."""
    cap = 70_000            # 18 chunks per instance
    asset = effects.instancing(cap)
    prog = ctx.create_program(bh.lower(asset))
    fxs = [prog.create_effect() for _ in range(4)]
    orcs = [OracleRunner(asset) for _ in range(4)]
    plan = {0: [cap, cap // 2, 0, cap], 3: [0, cap, cap // 3, cap], 5: [cap // 5, 0, cap, cap], 7: [cap, cap, cap, cap]}
    t = 0.0
    for f in range(9):
        dt = 1 / 60 if f in plan else 5.0     # (lifetime 12 s: three long frames kill everything spawned before)
        ctx.frame_begin(dt, t)
        for i, (fx, orc) in enumerate(zip(fxs, orcs)):
            frozen = i == 3 and f in (3, 4)
            fx.set_simulated(not frozen)
            if frozen:
                continue
            n = plan.get(f, [0] * 4)[i]
            xf = translation(10.0 * i, -5.0, 0.5 * i)
            fx.set_frame(n, frame_seed(f * 16 + i), xf)
            orc.step(Frame(dt, n, frame_seed(f * 16 + i), xf, time=t))
        ctx.simulate()
        t += dt
        for i, (fx, orc) in enumerate(zip(fxs, orcs)):
            ref = orc.state()
            m = fx.metadata()
            got = {"counters": {k: m[k] for k in ref["counters"]}, "alive": fx.alive_list(), "dead": fx.dead_list(),
                   "attrs": {a.name: fx.read_attr(a.id).view(np.uint32) for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME)}}
            ref["attrs"] = {k: ref["attrs"][k] for k in got["attrs"]}
            assert_same_state(ref, got, f"frame {f}, instance {i}")
    assert _slot_init_frames(prog) == 4, prog.kernel_info()
    prog.destroy()


def test_kernel_selection_for_the_baseline_configs(ctx):
    """C1-C5: init specialised at creation, update on a pre-built streaming kernel (or specialised)."""
    want = {"single_particle": "update=aot-stream:ProgNone", "firework_trails": "update=aot-stream:ProgDragAccel", "force_field": "update=aot-stream:ProgForceField",
            "instancing": "update=aot-stream:ProgAgeEuler", "ribbon": "update=aot-stream:ProgAge"}
    for name, upd in want.items():
        prog = ctx.create_program(bh.lower(getattr(effects, name)(4096)))
        info = prog.kernel_info()
        assert info.startswith("init=jit"), info
        assert upd in info, info
        prog.destroy()


def test_reference_contract_vectors_through_the_abi(ctx):
    """shader_contract_tests.rs:1254-1486 (vfx_indirect) replayed end to end: capacities (200, 5),
    alive (130, 1) -> max_update (130, 1), max_spawn (70, 4), instance_count after an update that
    kills nothing (130, 1)."""
    for cap, alive in ((200, 130), (5, 1)):
        asset = effects.single_particle(cap)  # only POSITION/SIZE3: nothing dies (cf. shader_contract_tests.rs:888)
        r = GpuRunner(asset, ctx=ctx)
        r.step(Frame(1 / 60, alive, 3))
        r.step(Frame(1 / 60, 0, 4))
        m = r.fx.metadata()
        assert (m["max_update"], m["max_spawn"], m["instance_count"], m["alive_count"]) == (alive, cap - alive, alive, alive)
        assert m["dispatch_x"] == (alive + 63) // 64
        np.testing.assert_array_equal(r.fx.alive_list(), np.arange(alive, dtype=np.uint32))


def test_large_burst_properties(ctx):
    """BASELINE config size (16M): size-independent checks — every slot allocated exactly once,
    alive + dead == capacity, ages advance by dt, and a sampled window matches the oracle."""
    cap = 1 << 24
    asset = effects.firework_trails(cap)
    r = GpuRunner(asset, ctx=ctx)
    seeds = [frame_seed(f) for f in range(4)]
    for f in range(4):
        r.step(Frame(1 / 60, cap if f == 0 else 0, seeds[f]))
    m = r.fx.metadata()
    assert m["alive_count"] == cap and m["dead_count"] == 0 and m["fault"] == 0
    alive = r.fx.alive_list()
    assert np.array_equal(alive, np.arange(cap, dtype=np.uint32))
    age = r.fx.read_attr(A.AGE.id)[:, 0]
    expect = np.float32(0)
    for _ in range(4):
        expect = np.float32(expect + np.float32(1 / 60))
    assert (age == expect).all()
    # window check: slots [base, base+4096) of the big effect == a 4096-capacity oracle with slot_base
    base = 12345 * 1024
    orc = OracleRunner(effects.firework_trails(4096), slot_base=base)
    for f in range(4):
        orc.step(Frame(1 / 60, 4096 if f == 0 else 0, seeds[f]))
    ref = orc.state()
    for a in (A.POSITION, A.VELOCITY, A.LIFETIME, A.COLOR):
        np.testing.assert_array_equal(ref["attrs"][a.name], r.fx.read_attr(a.id).view(np.uint32)[base:base + 4096])
    r.fx.destroy()
    r.prog.destroy()


def test_committed_golden_fixtures(ctx):
    """The product against tests/golden/states_small.npz (made by tests/golden/make_golden.py from
    the oracle); the oracle is not called here."""
    import os
    from golden.make_golden import SMALL, scripts
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "states_small.npz"))
    sc = scripts()
    for name in SMALL:
        asset, frames = sc[name]
        r = GpuRunner(asset, ctx=ctx)
        for fr in frames:
            r.step(fr)
        st = r.state()
        np.testing.assert_array_equal(gold[f"{name}/alive"], st["alive"], err_msg=name)
        np.testing.assert_array_equal(gold[f"{name}/dead"], st["dead"], err_msg=name)
        np.testing.assert_array_equal(gold[f"{name}/counters"], np.array([st["counters"][k] for k in sorted(st["counters"])], dtype=np.uint32), err_msg=name)
        for an, v in st["attrs"].items():
            np.testing.assert_array_equal(gold[f"{name}/attr/{an}"], v, err_msg=f"{name}/{an}")
        r.fx.destroy()
        r.prog.destroy()


def test_many_chunks_die_off_and_respawn(ctx):
    """318 chunks (> one workgroup's worth of chunk counts): k_compact's strided prefix, the
    other-column move and the last-killed-first slot reuse, against the (OpenMP) oracle."""
    cap = 1_300_000
    asset = effects.firework_trails(cap)
    g, o = GpuRunner(asset, ctx=ctx), OracleRunner(asset, omp=True)
    frames = [Frame(1 / 60, cap, frame_seed(0))] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(1, 58)]
    frames += [Frame(1 / 60, 400_000, frame_seed(58), time=58 / 60)] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(59, 66)]
    for i, fr in enumerate(frames):
        g.step(fr)
        o.step(fr)
        if i in (47, 50, 55, 57, 58, 61, 65):
            assert_same_state(o.state(), g.state(), f"frame {i}")
    assert 0 < g.fx.alive_count() < cap
    g.fx.destroy()
    g.prog.destroy()


def test_full_size_die_off_invariants(ctx):
    """BASELINE config size (16,777,216) through the die-off, checked with size-independent
    properties computed in numpy (no oracle): stable compaction (the alive list stays strictly
    increasing), survivors == particles with age < lifetime, the k-th casualty in list order lands
    on dead row n-1-k (vfx_update.wgsl:150-151), alive + dead is a partition of the slots."""
    cap = 1 << 24
    asset = effects.firework_trails(cap)
    r = GpuRunner(asset, ctx=ctx)
    r.step(Frame(1 / 60, cap, frame_seed(0)))
    for f in range(1, 46):
        r.step(Frame(1 / 60, 0, frame_seed(f), time=f / 60))
    life = r.fx.read_attr(A.LIFETIME.id)[:, 0]
    assert r.fx.alive_count() == cap
    checked = 0
    check_frames = {48, 54, 60, 66, 72, 75}
    for f in range(46, 76):
        if f in check_frames:
            prev_alive = r.fx.alive_list()
        r.step(Frame(1 / 60, 0, frame_seed(f), time=f / 60))
        if f not in check_frames:
            continue
        n = len(prev_alive)
        age = r.fx.read_attr(A.AGE.id)[:, 0]
        alive = r.fx.alive_list()
        dead = r.fx.dead_list()
        keep = age[prev_alive] < life[prev_alive]
        np.testing.assert_array_equal(alive, prev_alive[keep])
        cas = prev_alive[~keep]
        # dead rows [alive_count, n) hold this frame's casualties, last casualty on top of the stack
        np.testing.assert_array_equal(dead[: n - len(alive)], cas[::-1])
        assert len(alive) + len(dead) == cap
        if len(alive) > 1:
            assert (np.diff(alive.astype(np.int64)) > 0).all()
        m = r.fx.metadata()
        assert m["alive_count"] == len(alive) and m["max_update"] == n and m["dead_count"] == n - len(alive)
        checked += 1
    assert checked == len(check_frames)
    assert r.fx.alive_count() == 0 or r.fx.alive_count() < cap // 100
    # all slots are free again: the dead list is a permutation of [0, cap)
    for f in range(76, 80):
        r.step(Frame(1 / 60, 0, frame_seed(f), time=f / 60))
    assert r.fx.alive_count() == 0
    dead = r.fx.dead_list()
    assert len(dead) == cap and np.array_equal(np.sort(dead), np.arange(cap, dtype=np.uint32))
    r.fx.destroy()
    r.prog.destroy()


def test_simulation_condition_freezes_an_instance(ctx):
    """SimulationCondition::WhenVisible (spawn.rs:983-991, mod.rs:4347-4356): an invisible instance is neither
    ticked nor simulated; the other instances of the same batch go on. Oracle: simply not stepped."""
    cap = 9000
    asset = effects.instancing(cap, rate=cap / 0.25)
    prog = ctx.create_program(bh.lower(asset))
    fxs = [prog.create_effect() for _ in range(3)]
    orcs = [OracleRunner(asset) for _ in range(3)]
    sps = [bh.EffectSpawner(asset.spawner) for _ in range(3)]
    rng = bh.Pcg32()
    for f in range(50):
        ctx.frame_begin(1 / 60, f / 60)
        for i, (fx, orc, sp) in enumerate(zip(fxs, orcs, sps)):
            visible = not (i == 1 and 10 <= f < 30) and not (i == 2 and f % 4 == 0)
            fx.set_simulated(visible)
            if not visible:
                fx.set_frame(12345, 1)  # a request made while frozen is dropped
                continue
            n, seed = sp.tick(1 / 60, rng), frame_seed(f * 8 + i)
            fx.set_frame(n, seed)
            orc.step(Frame(1 / 60, n, seed, time=f / 60))
        ctx.simulate()
        if f in (9, 10, 29, 30, 49):
            for fx, orc in zip(fxs, orcs):
                ref = orc.state()
                np.testing.assert_array_equal(ref["alive"], fx.alive_list())
                np.testing.assert_array_equal(ref["dead"], fx.dead_list())
                for a in (A.POSITION, A.VELOCITY, A.AGE):
                    np.testing.assert_array_equal(ref["attrs"][a.name], fx.read_attr(a.id).view(np.uint32))
    prog.destroy()


def test_ribbon_sort_is_stable_on_equal_keys(ctx):
    """Equal (RIBBON_ID, AGE) keys keep their list order (vfx_sort.wgsl only moves past strictly greater keys)."""
    cap = 10000
    asset = effects.ribbon(cap)
    g = GpuRunner(asset, ctx=ctx)
    g.step(Frame(1 / 60, cap, 5))                       # list = identity, every key equal
    np.testing.assert_array_equal(g.fx.alive_list(), np.arange(cap, dtype=np.uint32))
    rid = (np.arange(cap, dtype=np.uint32) * 7919 % 3).reshape(cap, 1)   # three ribbons, every age still equal
    g.fx.write_attr(A.RIBBON_ID.id, rid)
    g.step(Frame(1 / 60, 0, 6))
    want = np.concatenate([np.nonzero(rid[:, 0] == r)[0] for r in range(3)]).astype(np.uint32)   # stable partition
    np.testing.assert_array_equal(g.fx.alive_list(), want)
    g.fx.destroy()
    g.prog.destroy()


# ---- HNB_LIST_ORDER_SLOT: the second canonical schedule (lists kept in increasing slot order) -----------------

@pytest.fixture(scope="module")
def slot_ctx():
    c = bh.Context(0)
    c.set_list_order("slot")
    yield c
    c.close()


def _slot_oracle(asset, **kw):
    o = OracleRunner(asset, **kw)
    o.fx.set_list_order(True)
    return o


def test_slot_order_firework_life_cycle(slot_ctx):
    cap = 20000
    asset = effects.firework_trails(cap)
    frames = burst_then_run(cap, 80) + [Frame(1 / 60, 7777, frame_seed(100))] + [Frame(1 / 60, 0, frame_seed(101 + f)) for f in range(30)]
    g = GpuRunner(asset, ctx=slot_ctx)
    st = run_script(g, frames, _slot_oracle(asset), every=6)
    alive = st["alive"].astype(np.int64)
    assert len(alive) > 1 and (np.diff(alive) > 0).all()   # the list really is in slot order
    g.prog.destroy()


def test_slot_order_with_large_spawns_slot_major(slot_ctx):
    """HNB_LIST_ORDER_SLOT and the slot-major init together (round 6): the dead list holds the free slots ascending, the lists are rebuilt from the alive
    bytes after every frame - the marks of a partial re-fill (alive byte 2) must all have become particles before that rebuild counts bytes."""
    cap = 120_011      # 30 chunks
    asset = effects.firework_trails(cap)
    g, o = GpuRunner(asset, ctx=slot_ctx), _slot_oracle(asset, omp=True)
    t = 0.0
    for f, (dt, spawn) in enumerate([(1 / 60, cap), (0.3, 0), (0.3, 0), (0.3, 0), (1 / 60, cap // 4), (0.3, 1000), (1 / 60, cap), (0.5, 0), (0.5, cap // 3), (1 / 60, cap - 7)]):
        fr = Frame(dt, spawn, frame_seed(f), time=t)
        t += dt
        g.step(fr)
        o.step(fr)
        assert_same_state(o.state(), g.state(), f"slot order, frame {f}")
        alive = g.fx.alive_list().astype(np.int64)
        assert len(alive) < 2 or (np.diff(alive) > 0).all()
    assert _slot_init_frames(g.prog) == 5, g.prog.kernel_info()
    g.prog.destroy()


def test_slot_order_churn_batch(slot_ctx):
    """Steady spawn/kill churn over a batch of instances: slots are recycled last-killed-first, the list stays sorted."""
    cap, n_inst = 9000, 4
    asset = effects.instancing(cap, rate=cap / 0.25)
    w = bh.ExprWriter()
    prog = slot_ctx.create_program(bh.lower(asset))
    fxs = [prog.create_effect() for _ in range(n_inst)]
    orcs = [_slot_oracle(asset) for _ in range(n_inst)]
    # shorten lifetimes through AGE writes? not needed: rate = cap / 0.25 s fills the capacity, then spawns are capped
    sps = [bh.EffectSpawner(asset.spawner) for _ in range(n_inst)]
    rng = bh.Pcg32()
    for f in range(60):
        slot_ctx.frame_begin(1 / 60, f / 60)
        for i, (fx, orc, sp) in enumerate(zip(fxs, orcs, sps)):
            n, seed = (sp.tick(1 / 60, rng) if (f + i) % 3 else 0), frame_seed(f * 16 + i)
            fx.set_frame(n, seed)
            orc.step(Frame(1 / 60, n, seed, time=f / 60))
        slot_ctx.simulate()
        if f % 10 == 9:
            for fx, orc in zip(fxs, orcs):
                ref = orc.state()
                np.testing.assert_array_equal(ref["alive"], fx.alive_list())
                np.testing.assert_array_equal(ref["dead"], fx.dead_list())
                for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME):
                    np.testing.assert_array_equal(ref["attrs"][a.name], fx.read_attr(a.id).view(np.uint32))
    prog.destroy()


@pytest.mark.parametrize("name", ["update_generic", "update_all_macros", "shapes_volume_local"])
def test_slot_order_zoo(slot_ctx, name):
    asset = ZOO[name]()
    cap = asset.capacity
    frames = [Frame(1 / 60, cap // 2, frame_seed(0))]
    for f in range(1, 60):
        frames.append(Frame(1 / 60, (cap // 9) if f % 5 == 0 else 0, frame_seed(f), time=f / 60.0))
    g = GpuRunner(asset, ctx=slot_ctx)
    run_script(g, frames, _slot_oracle(asset), every=10)
    g.prog.destroy()


@pytest.mark.parametrize("cap", [1, 3, 63, 64, 255, 256, 257, 1023, 1025, 4095, 4096, 4097, 8191, 12289])
@pytest.mark.parametrize("order", ["spawn", "slot"])
def test_ragged_capacities_and_random_spawn_requests(ctx, slot_ctx, cap, order):
    """Capacities around the wave / step / chunk boundaries, spawn requests from 0 to beyond the free capacity,
    short lifetimes so that slots are recycled many times: both list orders against the oracle."""
    c = ctx if order == "spawn" else slot_ctx
    w = bh.ExprWriter()
    mods = [bh.SetPositionSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(1.0).expr(), bh.ShapeDimension.Volume),
            bh.SetVelocitySphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.0).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(0.02).uniform(w.lit(0.2)).expr())]
    upd = [bh.LinearDragModifier(w.lit(2.0).expr()), bh.AccelModifier(w.lit((0.0, -5.0, 0.0)).expr())]
    asset = bh.EffectAsset(cap, bh.SpawnerSettings.once(1.0), w.finish())
    for m in mods:
        asset.init(m)
    for m in upd:
        asset.update(m)
    g = GpuRunner(asset, ctx=c)
    o = OracleRunner(asset)
    o.fx.set_list_order(order == "slot")
    rng = np.random.default_rng(cap)
    frames = []
    for f in range(48):
        r = rng.random()
        spawn = 0 if r < 0.25 else (cap * 3 if r > 0.9 else int(rng.integers(0, max(2, cap // 2 + 2))))
        frames.append(Frame(1 / 60, spawn, frame_seed(f + cap), time=f / 60))
    run_script(g, frames, o, every=8)
    g.fx.destroy()
    g.prog.destroy()


def test_batched_frame_inputs_match_per_instance_calls(ctx):
    """hnb_program_set_frames == one hnb_effect_set_frame per instance."""
    cap, n_inst = 5000, 6
    asset = effects.firework_trails(cap)
    pa, pb = ctx.create_program(bh.lower(asset)), ctx.create_program(bh.lower(asset))
    fa, fb = [pa.create_effect() for _ in range(n_inst)], [pb.create_effect() for _ in range(n_inst)]
    rng = np.random.default_rng(7)
    for f in range(70):
        spawns = [int(rng.integers(0, 900)) if f % 5 == 0 else 0 for _ in range(n_inst)]
        seeds = [frame_seed(f * 16 + i) for i in range(n_inst)]
        xf = np.stack([translation(i, -i, 0.5 * f) for i in range(n_inst)])
        ctx.frame_begin(1 / 60, f / 60)
        for i, fx in enumerate(fa):
            fx.set_frame(spawns[i], seeds[i], xf[i])
        pb.set_frames(spawns, seeds, xf)
        ctx.simulate()
    for x, y in zip(fa, fb):
        np.testing.assert_array_equal(x.alive_list(), y.alive_list())
        np.testing.assert_array_equal(x.dead_list(), y.dead_list())
        for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME, A.COLOR):
            np.testing.assert_array_equal(x.read_attr(a.id).view(np.uint32), y.read_attr(a.id).view(np.uint32))
    pa.destroy()
    pb.destroy()


def test_lifecycle_create_destroy_reparent(ctx):
    """Instances come and go (slab slots are recycled, the last instance moves into a destroyed one's place),
    children are re-parented and parents destroyed first: the survivors keep simulating correctly."""
    cap = 3000
    asset = effects.firework_trails(cap)
    prog = ctx.create_program(bh.lower(asset))
    fxs = [prog.create_effect() for _ in range(5)]
    orcs = [OracleRunner(asset) for _ in range(5)]

    def step(f):
        ctx.frame_begin(1 / 60, f / 60)
        for i, (fx, orc) in enumerate(zip(fxs, orcs)):
            n, seed = (cap if f % 37 == 0 else 0), frame_seed(f * 8 + i)
            fx.set_frame(n, seed)
            orc.step(Frame(1 / 60, n, seed, time=f / 60))
        ctx.simulate()

    def check():
        for fx, orc in zip(fxs, orcs):
            ref = orc.state()
            np.testing.assert_array_equal(ref["alive"], fx.alive_list())
            np.testing.assert_array_equal(ref["dead"], fx.dead_list())
            np.testing.assert_array_equal(ref["attrs"]["position"], fx.read_attr(A.POSITION.id).view(np.uint32))

    for f in range(20):
        step(f)
    check()
    fxs.pop(1).destroy(); orcs.pop(1)      # the last instance takes index 1
    for f in range(20, 50):
        step(f)
    check()
    fxs.append(prog.create_effect()); orcs.append(OracleRunner(asset))   # reuses the freed slab
    fxs.pop(0).destroy(); orcs.pop(0)
    for f in range(50, 90):
        step(f)
    check()
    prog.destroy()

    # event links: re-parent a child, destroy a parent before its child
    rocket_a = ctx.create_program(bh.lower(effects.firework_rocket())).create_effect()
    rocket_b = ctx.create_program(bh.lower(effects.firework_rocket())).create_effect()
    child_prog = ctx.create_program(bh.lower(effects.firework_trails_child(4000)))
    child = child_prog.create_effect()
    child.set_parent(rocket_a, 1, 512)
    for f in range(5):
        ctx.frame_begin(1 / 60, f / 60)
        rocket_a.set_frame(2, frame_seed(f)); rocket_b.set_frame(2, frame_seed(100 + f)); child.set_frame(0, frame_seed(200 + f))
        ctx.simulate()
    child.set_parent(rocket_b, 1, 512)     # re-parent
    rocket_a.destroy()                      # old parent goes away
    for f in range(5, 90):
        ctx.frame_begin(1 / 60, f / 60)
        rocket_b.set_frame(1 if f % 10 == 0 else 0, frame_seed(100 + f)); child.set_frame(0, frame_seed(200 + f))
        ctx.simulate()
    assert child.metadata()["particle_counter"] > 0   # rocket_b's explosions reached the child
    rocket_b.destroy()                      # parent destroyed before the child
    ctx.frame_begin(1 / 60, 2.0)
    child.set_frame(0, 1)
    with pytest.raises(bh.HanabiError):     # the child reads its parent particle and has none any more
        ctx.simulate()
    child_prog.destroy()


def test_instances_with_different_property_values_share_one_launch(ctx):
    """The reference cannot merge instances whose property values differ (batch.rs:153-173, property_key);
    here every instance carries its own parameter block and the whole program still runs in one launch."""
    cap, n_inst = 6000, 4
    asset = effects.force_field(cap)
    prog = ctx.create_program(bh.lower(asset))
    fxs = [prog.create_effect() for _ in range(n_inst)]
    orcs = [OracleRunner(asset) for _ in range(n_inst)]
    for i, (fx, orc) in enumerate(zip(fxs, orcs)):
        props = {"repulsor_position": (0.1 * i, 0.5 - 0.1 * i, 0.05 * i), "repulsor_accel": -15.0 - 3.0 * i, "attraction_accel": 20.0 + i}
        for k, v in props.items():
            fx.set_property(k, v)
            orc.fx.set_property(k, v)
    for f in range(60):
        ctx.frame_begin(1 / 60, f / 60)
        for i, (fx, orc) in enumerate(zip(fxs, orcs)):
            if f == 30 and i == 2:   # a property changes mid-run on one instance only
                fx.set_property("sticky_factor", 3.5)
                orc.fx.set_property("sticky_factor", 3.5)
            n, seed = (cap if f == 0 else 0), frame_seed(f * 8 + i)
            fx.set_frame(n, seed)
            orc.step(Frame(1 / 60, n, seed, time=f / 60))
        ctx.simulate()
    for fx, orc in zip(fxs, orcs):
        ref = orc.state()
        np.testing.assert_array_equal(ref["alive"], fx.alive_list())
        for a in (A.POSITION, A.VELOCITY, A.AGE):
            np.testing.assert_array_equal(ref["attrs"][a.name], fx.read_attr(a.id).view(np.uint32))
    prog.destroy()


def test_denormals_and_special_values_match(ctx, kernels):
    """f32 denormals are preserved (no flush to zero), and inf / NaN propagate identically (normalize of a zero vector,
    0 * inf, division by zero) in the oracle and on the GPU, for the streaming kernel's macro ops and for the VM."""
    w = bh.ExprWriter()
    F = bh.ValueType(bh.ScalarType.Float)
    init = [bh.SetAttributeModifier(A.POSITION, w.lit((0.0, 0.0, 0.0)).expr()),
            bh.SetAttributeModifier(A.VELOCITY, (w.rand(bh.VectorType.VEC3F) * w.lit(1e-30)).expr()),     # tiny: Euler products underflow to denormals
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(10.0).expr()),
            bh.SetAttributeModifier(A.F32_0, (w.rand(F) * w.lit(1e-20)).expr()),
            bh.SetAttributeModifier(A.F32X3_0, (w.attr(A.POSITION) - w.attr(A.POSITION)).normalized().expr()),  # normalize(0) -> NaN
            bh.SetAttributeModifier(A.F32_1, (w.lit(1.0) / (w.rand(F) - w.rand(F)) * w.lit(0.0)).expr())]
    upd = [bh.LinearDragModifier(w.lit(30.0).expr()),      # velocity *= 0.5 per frame: walks through the denormal range to zero
           bh.SetAttributeModifier(A.F32_0, (w.attr(A.F32_0) * w.lit(1e-6)).expr()),
           bh.SetAttributeModifier(A.F32_1, (w.attr(A.F32_1) + w.attr(A.F32_0)).expr())]
    asset = bh.EffectAsset(3000, bh.SpawnerSettings.once(3000.0), w.finish())
    for m in init:
        asset.init(m)
    for m in upd:
        asset.update(m)
    frames = [Frame(1 / 60, 3000, 11)] + [Frame(1 / 60, 0, 12 + f, time=f / 60) for f in range(1, 80)]
    g = GpuRunner(asset, ctx=ctx)
    st = run_script(g, frames, OracleRunner(asset), every=10)
    f0 = st["attrs"]["f32_0"].view(np.float32)
    assert (f0 == 0.0).all()   # ... and went through denormals on the way (checked bit-exact at every 10th frame)
    assert np.isnan(st["attrs"]["f32x3_0"].view(np.float32)).all()
    g.fx.destroy()
    g.prog.destroy()


def test_lifetime_culling_respects_host_writes_and_partial_respawns(ctx):
    """Lifetime culling (k_update_slots_stream): after frames in which no chunk read the LIFETIME plane, (1) a host write
    that shortens some lifetimes must kill exactly those particles in the next frame, (2) a partial respawn into culled
    chunks must be aged and killed by its own lifetimes. Both against the oracle's rule age + dt < lifetime."""
    cap = 3 * 4096 + 500
    w = bh.ExprWriter()
    init = [bh.SetAttributeModifier(A.POSITION, w.rand(bh.VectorType.VEC3F).expr()), bh.SetAttributeModifier(A.VELOCITY, w.lit((0.0, 1.0, 0.0)).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()), bh.SetAttributeModifier(A.LIFETIME, w.lit(5.0).uniform(w.lit(6.0)).expr())]
    accel = bh.AccelModifier(w.lit((0.0, -1.0, 0.0)).expr())
    asset = bh.EffectAsset(cap, bh.SpawnerSettings.once(float(cap)), w.finish())
    for m in init:
        asset.init(m)
    asset.update(accel)
    g = GpuRunner(asset, ctx=ctx)
    assert "stream" in g.prog.kernel_info()
    dt = 1 / 60
    g.step(Frame(dt, cap, frame_seed(0)))
    for f in range(1, 8):   # culled frames: every lifetime is at least 5 s
        g.step(Frame(dt, 0, frame_seed(f)))
    assert g.fx.alive_count() == cap
    life = g.fx.read_attr(A.LIFETIME.id).reshape(-1).copy()
    age = g.fx.read_attr(A.AGE.id).reshape(-1)
    doomed = np.zeros(cap, dtype=bool)
    doomed[::7] = True
    doomed[4096:4096 + 64] = True
    life[doomed] = age[doomed]            # age + dt < lifetime fails in the next frame
    g.fx.write_attr(A.LIFETIME.id, life)
    g.step(Frame(dt, 0, frame_seed(8)))
    assert g.fx.alive_count() == cap - int(doomed.sum())
    dead_now = set(int(s) for s in g.fx.dead_list())
    assert dead_now == set(np.nonzero(doomed)[0].tolist())
    for f in range(9, 14):  # culled again on the new bounds
        g.step(Frame(dt, 0, frame_seed(f)))
    assert g.fx.alive_count() == cap - int(doomed.sum())
    # respawn half of the free slots: their lifetimes are fresh draws, the survivors' are not touched
    n_new = int(doomed.sum()) // 2
    g.step(Frame(dt, n_new, frame_seed(14)))
    assert g.fx.alive_count() == cap - int(doomed.sum()) + n_new
    life2 = g.fx.read_attr(A.LIFETIME.id).reshape(-1)
    keep = ~doomed
    np.testing.assert_array_equal(life2[keep].view(np.uint32), life[keep].view(np.uint32))
    # shorten ONE respawned particle's lifetime by a host write and check that only it dies
    alive = g.fx.alive_list()
    victim = int(alive[-1])
    life3 = life2.copy()
    life3[victim] = 0.0
    g.fx.write_attr(A.LIFETIME.id, life3)
    g.step(Frame(dt, 0, frame_seed(15)))
    assert g.fx.alive_count() == cap - int(doomed.sum()) + n_new - 1
    assert victim not in set(int(s) for s in g.fx.alive_list())
    g.prog.destroy()


@pytest.mark.parametrize("order", ["spawn", "slot"])
@pytest.mark.parametrize("gap", [1, 2])
def test_instance_index_reuse_after_destroy(ctx, slot_ctx, order, gap):
    """An instance created at a table index that a destroyed instance used before must not inherit that instance's last
    casualty count (either frame parity): it once left the new instance's counters unrotated and, with slot-ordered lists,
    produced an alive count of billions."""
    c = ctx if order == "spawn" else slot_ctx
    cap = 5000
    w = bh.ExprWriter()
    mods = [bh.SetAttributeModifier(A.POSITION, w.rand(bh.VectorType.VEC3F).expr()), bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            bh.SetAttributeModifier(A.LIFETIME, w.lit(0.02).uniform(w.lit(0.08)).expr())]
    asset = bh.EffectAsset(cap, bh.SpawnerSettings.once(float(cap)), w.finish())
    for m in mods:
        asset.init(m)
    prog = c.create_program(bh.lower(asset))
    a, b = prog.create_effect(), prog.create_effect()
    f = 0

    def frame(pairs):
        nonlocal f
        c.frame_begin(1 / 60, f / 60)
        for fx, orc, spawn in pairs:
            seed = frame_seed(1000 + f * 3 + (0 if fx is a else 1))
            fx.set_frame(spawn, seed)
            if orc is not None:
                orc.step(Frame(1 / 60, spawn, seed, time=f / 60))
        c.simulate()
        f += 1

    frame([(a, None, 100), (b, None, cap)])
    for _ in range(3):          # particles of b (index 1) die in these frames: its casualty rows are non-zero
        frame([(a, None, 0), (b, None, 0)])
    assert b.alive_count() < cap
    b.destroy()                 # index 1 is free again
    for _ in range(gap):        # 1 or 2 frames: the new instance's first frame lands on either parity
        frame([(a, None, 0)])
    n = prog.create_effect()    # reuses index 1
    orc = OracleRunner(asset)
    orc.fx.set_list_order(order == "slot")
    frame([(a, None, 0), (n, orc, 0)])          # nothing to update: only the counters rotate
    frame([(a, None, 0), (n, orc, 700)])
    for _ in range(6):
        frame([(a, None, 0), (n, orc, 0)])
    ref = orc.state()
    m = n.metadata()
    assert {k: m[k] for k in ref["counters"]} == ref["counters"]
    np.testing.assert_array_equal(ref["alive"], n.alive_list())
    np.testing.assert_array_equal(ref["dead"], n.dead_list())
    prog.destroy()


def test_effect_properties_reach_the_gpu(ctx):
    """EffectProperties (the reference's per-instance property component) applied through the binding: the declared values
    are set by name, a value for an undeclared property is ignored, results equal the oracle's with the same values."""
    asset = effects.force_field(6000)
    g, o = GpuRunner(asset, ctx=ctx), OracleRunner(asset)
    ep = bh.EffectProperties().with_properties([("repulsor_accel", -25.0), ("repulsor_position", (0.1, 0.4, 0.0)), ("not_declared", 1.0)])
    g.fx.apply_properties(ep)
    props = {"repulsor_accel": [-25.0], "repulsor_position": [0.1, 0.4, 0.0]}
    frames = [Frame(1 / 60, 6000 if f == 0 else 0, frame_seed(f), time=f / 60, props=props if f == 0 else None) for f in range(40)]
    for fr in frames:
        fr_gpu = Frame(fr.dt, fr.spawn, fr.seed, time=fr.time)   # the GPU side got its values from apply_properties
        g.step(fr_gpu)
        o.step(fr)
    assert_same_state(o.state(), g.state(), "force field with EffectProperties")
    g.prog.destroy()
