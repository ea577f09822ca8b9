"""Differential fuzzing of the lowering and the program interpreters: seeded random assets (fuzz_assets.py),
oracle vs the product, bit-exact. CPU: the host build of the product interpreters (tests/cpu_vm); GPU: the real
path through the C ABI, with specialised kernels and with the interpreters."""
import pytest

import bevy_hanabi_amd as bh
from fuzz_assets import random_asset, random_frames
from helpers import CpuVmRunner, GpuRunner, OracleRunner, run_script

CPU_SEEDS = list(range(60))
GPU_SEEDS = list(range(100, 124))


def _run(seed, make_runner, abstract=False):
    asset = random_asset(seed, abstract=abstract)
    try:
        blob = bh.lower(asset)
    except bh.ShaderGenerateError as e:
        # register pressure of a deep random tree: a lowering error, never a wrong result; the oracle agrees it is an asset
        pytest.skip(f"seed {seed}: {e}")
    except bh.ExprError as e:
        # a type error of the expression language: the oracle has to refuse the same asset
        import oracle
        o = OracleRunner(asset)
        with pytest.raises(oracle.OracleError):
            for fr in random_frames(seed, asset.capacity):
                o.step(fr)
        pytest.skip(f"seed {seed}: rejected by both: {e}")
    bh.validate_program(blob)
    run_script(make_runner(asset), random_frames(seed, asset.capacity), OracleRunner(asset), every=6)


@pytest.mark.parametrize("seed", CPU_SEEDS)
def test_fuzz_cpu(seed):
    _run(seed, lambda a: CpuVmRunner(a))


@pytest.fixture(scope="module")
def ctx():
    c = bh.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", GPU_SEEDS)
@pytest.mark.parametrize("jit", ["1", "0"])
def test_fuzz_gpu(ctx, seed, jit, monkeypatch):
    monkeypatch.setenv("HNB_JIT", jit)
    holder = {}

    def mk(a):
        holder["g"] = GpuRunner(a, ctx=ctx)
        return holder["g"]

    try:
        _run(seed, mk)
    finally:
        if "g" in holder:
            holder["g"].prog.destroy()


# ---- programs above the 32-register fast file ---------------------------------------------------------
def wide_asset(n_rands, capacity=300):
    """One statement keeping `n_rands` hoisted vec4 draws alive at once (4 registers each)."""
    w = bh.ExprWriter()
    A = bh.Attribute
    acc = w.rand(bh.VectorType.VEC4F)
    for k in range(1, n_rands):
        acc = acc + w.rand(bh.VectorType.VEC4F) * w.lit(1.0 + 0.25 * k)
    upd = w.rand(bh.VectorType.VEC4F)
    for k in range(1, n_rands):
        upd = upd.max(w.rand(bh.VectorType.VEC4F) - w.attr(A.F32X4_0) * w.lit(0.125 * k))
    init = [bh.SetAttributeModifier(A.POSITION, w.rand(bh.VectorType.VEC3F).expr()),
            bh.SetAttributeModifier(A.VELOCITY, w.lit((0.0, 1.0, 0.0)).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            bh.SetAttributeModifier(A.LIFETIME, w.lit(0.2).uniform(w.lit(0.7)).expr()),
            bh.SetAttributeModifier(A.F32X4_0, acc.expr())]
    update = [bh.SetAttributeModifier(A.F32X4_1, upd.expr())]
    asset = bh.EffectAsset(capacity, bh.SpawnerSettings.once(float(capacity)), w.finish())
    for m in init:
        asset.init(m)
    for m in update:
        asset.update(m)
    return asset


def _header_regs(blob):
    import struct
    words = struct.unpack_from("<14I", blob)   # HnbProgramHeader: init_regs, update_regs are words 12, 13
    return words[12], words[13]


@pytest.mark.parametrize("n_rands", [7, 12, 26])
def test_wide_register_file_cpu(n_rands):
    asset = wide_asset(n_rands)
    blob = bh.lower(asset)
    bh.validate_program(blob)
    init_regs, update_regs = _header_regs(blob)
    assert (max(init_regs, update_regs) > 32) == (n_rands > 5)
    run_script(CpuVmRunner(asset), random_frames(n_rands, asset.capacity, n=18), OracleRunner(asset), every=3)


def test_register_limit_is_a_lowering_error():
    with pytest.raises(bh.ShaderGenerateError, match="more than 128 per-particle registers"):
        bh.lower(wide_asset(40))


@pytest.mark.gpu
@pytest.mark.parametrize("n_rands", [12, 26])
@pytest.mark.parametrize("jit", ["1", "0"])
def test_wide_register_file_gpu(ctx, n_rands, jit, monkeypatch):
    monkeypatch.setenv("HNB_JIT", jit)
    asset = wide_asset(n_rands)
    g = GpuRunner(asset, ctx=ctx)
    try:
        assert "wide-file" in g.prog.kernel_info()
        run_script(g, random_frames(n_rands, asset.capacity, n=18), OracleRunner(asset), every=3)
    finally:
        g.prog.destroy()


# ---- programs above 128 uniform parameter words (U bank bits of the instruction encoding) ----------------
def uniform_heavy_asset(n_lits, streamable_update, capacity=300):
    w = bh.ExprWriter()
    A = bh.Attribute
    rng = __import__("numpy").random.default_rng(n_lits)
    acc = w.rand(bh.VectorType.VEC4F)
    for k in range(n_lits):   # distinct vec4 literals: 4 parameter words each
        acc = acc * w.lit(tuple(float(x) for x in rng.uniform(0.5, 1.5, 4).round(4))) + w.lit(tuple(float(x) for x in rng.uniform(-1, 1, 4).round(4)))
    init = [bh.SetPositionSphereModifier(w.lit((0.0, 0.5, 0.0)).expr(), w.lit(1.5).expr(), bh.ShapeDimension.Volume),
            bh.SetAttributeModifier(A.VELOCITY, (w.rand(bh.VectorType.VEC3F) - w.lit(0.5)).expr()),
            bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()),
            bh.SetAttributeModifier(A.LIFETIME, w.lit(0.2).uniform(w.lit(0.9)).expr()),
            bh.SetAttributeModifier(A.F32X4_0, acc.expr())]
    # the update stream's operands are allocated after the init stream's: their U indices are above 128
    update = [bh.AccelModifier(w.lit((0.25, -9.0, 0.5)).expr()), bh.LinearDragModifier(w.lit(0.75).expr()),
              bh.KillSphereModifier(w.lit((0.0, 0.0, 0.0)).expr(), w.lit(2.5).expr(), False)]
    if not streamable_update:
        update.append(bh.SetAttributeModifier(A.F32X4_1, (w.attr(A.F32X4_0) * w.lit((0.9, 0.8, 0.7, 0.6)) + w.rand(bh.VectorType.VEC4F)).expr()))
    asset = bh.EffectAsset(capacity, bh.SpawnerSettings.once(float(capacity)), w.finish())
    for m in init:
        asset.init(m)
    for m in update:
        asset.update(m)
    return asset


@pytest.mark.parametrize("streamable", [True, False])
def test_many_uniform_words_cpu(streamable):
    import struct
    asset = uniform_heavy_asset(20, streamable)
    blob = bh.lower(asset)
    bh.validate_program(blob)
    assert struct.unpack_from("<12I", blob)[11] > 128   # HnbProgramHeader::n_uregs
    run_script(CpuVmRunner(asset), random_frames(5, asset.capacity, n=18), OracleRunner(asset), every=3)


def test_uniform_word_limit_is_a_lowering_error():
    with pytest.raises(bh.ShaderGenerateError, match="more than 256 uniform parameter words"):
        bh.lower(uniform_heavy_asset(40, True))


@pytest.mark.gpu
@pytest.mark.parametrize("streamable", [True, False])
@pytest.mark.parametrize("jit", ["1", "0"])
def test_many_uniform_words_gpu(ctx, streamable, jit, monkeypatch):
    monkeypatch.setenv("HNB_JIT", jit)
    asset = uniform_heavy_asset(20, streamable)
    g = GpuRunner(asset, ctx=ctx)
    try:
        run_script(g, random_frames(5, asset.capacity, n=18), OracleRunner(asset), every=3)
    finally:
        g.prog.destroy()


# ---- typed generator (float / int / uint / bool, NaN / inf / division by zero reachable) -------------------------
TYPED_CPU_SEEDS = list(range(2000, 2060))
TYPED_GPU_SEEDS = list(range(2100, 2116))


def _run_typed(seed, make_runner, abstract=False):
    from fuzz_assets import random_typed_asset
    asset = random_typed_asset(seed, abstract=abstract)
    try:
        bh.validate_program(bh.lower(asset))
    except bh.ExprError as e:
        if not abstract:
            raise
        import oracle   # an int literal the WGSL would not accept there: the oracle has to refuse the asset too
        o = OracleRunner(asset)
        with pytest.raises(oracle.OracleError):
            for fr in random_frames(seed, asset.capacity, n=24):
                o.step(fr)
        pytest.skip(f"seed {seed}: rejected by both: {e}")
    run_script(make_runner(asset), random_frames(seed, asset.capacity, n=24), OracleRunner(asset), every=4)


@pytest.mark.parametrize("seed", TYPED_CPU_SEEDS)
def test_fuzz_typed_cpu(seed):
    _run_typed(seed, lambda a: CpuVmRunner(a))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", TYPED_GPU_SEEDS)
@pytest.mark.parametrize("jit", ["1", "0"])
def test_fuzz_typed_gpu(ctx, seed, jit, monkeypatch):
    monkeypatch.setenv("HNB_JIT", jit)
    holder = {}

    def mk(a):
        holder["g"] = GpuRunner(a, ctx=ctx)
        return holder["g"]

    try:
        _run_typed(seed, mk)
    finally:
        if "g" in holder:
            holder["g"].prog.destroy()


# ---- i32 literals where floats / u32 are expected (WGSL abstract numerics, tests/test_abstract_numerics.py) --------------------
ABSTRACT_SEEDS = list(range(3000, 3050))


@pytest.mark.parametrize("seed", ABSTRACT_SEEDS)
def test_fuzz_abstract_literals_cpu(seed):
    if seed % 2:
        _run_typed(seed, lambda a: CpuVmRunner(a), abstract=True)
    else:
        _run(seed, lambda a: CpuVmRunner(a), abstract=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", ABSTRACT_SEEDS[:16])
def test_fuzz_abstract_literals_gpu(ctx, seed):
    holder = {}

    def mk(a):
        holder["g"] = GpuRunner(a, ctx=ctx)
        return holder["g"]

    try:
        (_run_typed if seed % 2 else _run)(seed, mk, abstract=True)
    finally:
        if "g" in holder:
            holder["g"].prog.destroy()


# ---- random worlds: several programs and instances in one context, random schedule -----------------------------
def _world(seed, order):
    """2-3 random programs x 1-5 instances (created and destroyed along the way), random spawn requests (none, some,
    more than fits), frozen instances, property changes, per-instance transforms; one simulate() per frame for everything.
    Every instance has its own oracle, stepped only when the instance is simulated."""
    import numpy as np
    from fuzz_assets import random_typed_asset
    from helpers import Frame, assert_same_state, frame_seed, stored_attrs

    import os
    trace = os.environ.get("HNB_FUZZ_TRACE") == "1"   # print the schedule and compare after every frame
    max_inst = int(os.environ.get("HNB_FUZZ_MAX_INST", "5"))   # soak runs raise it to grow the instance tables
    p_create, p_destroy = (0.08, 0.14) if max_inst <= 5 else (0.30, 0.50)
    rng = np.random.default_rng(seed)
    brng = np.random.default_rng(seed + 999)
    ctx = bh.Context(0)
    if order == "slot":
        ctx.set_list_order("slot")
    programs = []
    try:
        for p in range(int(rng.integers(2, 4))):
            cap = int([1, 63, 300, 4097, 9000][int(rng.integers(5))])
            s = int(rng.integers(1 << 20))
            asset = random_typed_asset(s, cap) if rng.random() < 0.6 else random_asset(s, cap)
            programs.append({"asset": asset, "prog": ctx.create_program(bh.lower(asset)), "inst": [], "props": list(asset.module().property_names)})

        def add_instance(pr):
            orc = OracleRunner(pr["asset"])
            orc.fx.set_list_order(order == "slot")
            pr["inst"].append({"fx": pr["prog"].create_effect(), "orc": orc, "xf": None})

        def state_of(pr, it):
            m = it["fx"].metadata()
            keys = ["capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index", "particle_counter", "instance_count", "dead_count"]
            return {"counters": {k: m[k] for k in keys}, "alive": it["fx"].alive_list(), "dead": it["fx"].dead_list(),
                    "attrs": {a.name: it["fx"].read_attr(a.id).view(np.uint32) for a in stored_attrs(pr["asset"])}}

        for pr in programs:
            for _ in range(int(rng.integers(1, 4))):
                add_instance(pr)
        for f in range(40):
            dt = 1 / 60 if rng.random() < 0.8 else 1 / 24
            ctx.frame_begin(dt, f / 60)
            for pr in programs:
                r = rng.random()
                if r < p_create and len(pr["inst"]) < max_inst:
                    add_instance(pr)
                    if trace:
                        print(f"frame {f}: program {programs.index(pr)} + instance (now {len(pr['inst'])})", flush=True)
                elif r < p_destroy and len(pr["inst"]) > 1:
                    victim = int(rng.integers(len(pr["inst"])))
                    it = pr["inst"].pop(victim)
                    it["fx"].destroy()
                    if trace:
                        print(f"frame {f}: program {programs.index(pr)} - instance {victim} (now {len(pr['inst'])})", flush=True)
                cap = pr["asset"].capacity
                batched = brng.random() < 0.4   # (its own generator: the schedules of the recorded seeds stay what they were)
                batch = {}
                for it in pr["inst"]:
                    visible = rng.random() > 0.15
                    it["fx"].set_simulated(visible)
                    if not visible:
                        if trace:
                            print(f"frame {f}: program {programs.index(pr)} instance {pr['inst'].index(it)} frozen", flush=True)
                        continue
                    props = {}
                    if pr["props"] and rng.random() < 0.3:
                        props[pr["props"][int(rng.integers(len(pr["props"])))]] = [float(np.float32(rng.uniform(-2, 2)))]
                    if rng.random() < 0.2:
                        it["xf"] = np.array([1, 0, 0, rng.uniform(-3, 3), 0, 1, 0, rng.uniform(-3, 3), 0, 0, 1, rng.uniform(-3, 3)], dtype=np.float32)
                    r = rng.random()
                    spawn = 0 if r < 0.35 else (cap * 2 + 1 if r > 0.92 else int(rng.integers(0, cap // 2 + 2)))
                    sd = frame_seed(seed * 977 + f * 31 + int(rng.integers(1 << 16)))
                    for k, v in props.items():
                        it["fx"].set_property(k, v)
                    if batched:
                        batch[it["fx"].index()] = (spawn, sd, it["xf"])
                    else:
                        it["fx"].set_frame(spawn, sd, it["xf"])
                    it["orc"].step(Frame(dt, spawn, sd, it["xf"], time=f / 60, props=props))
                if batched:   # one hnb_program_set_frames call, rows in table order (frozen instances: ignored values)
                    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)
                    rows = [batch.get(i, (12345, 1, None)) for i in range(len(pr["inst"]))]
                    pr["prog"].set_frames([r[0] for r in rows], [r[1] for r in rows], [ident if r[2] is None else r[2] for r in rows])
            ctx.simulate()
            if trace or f % 8 == 7 or f == 39:
                for pi, pr in enumerate(programs):
                    for ii, it in enumerate(pr["inst"]):
                        assert_same_state(it["orc"].state(), state_of(pr, it), f"seed {seed} frame {f} program {pi} instance {ii}")
    finally:
        for pr in programs:
            pr["prog"].destroy()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["spawn", "slot"])
@pytest.mark.parametrize("seed", list(range(8)) + [103])   # 103: an instance created at an index a destroyed instance used (stale casualty counter)
def test_fuzz_world_gpu(seed, order):
    _world(seed, order)


# ---- random systems of linked effects (GPU spawn events, parent reads) -------------------------------------------
def _random_system(seed):
    """A root effect with 1-3 event channels, children on them (some with a grand-child), random event counts (typed
    expressions), capacities around the chunk size, event buffers from 'always overflows' to 'never'."""
    import numpy as np
    from fuzz_assets import TypedGen
    from helpers import EffectSpec

    rng = np.random.default_rng(seed)
    A = bh.Attribute
    U = bh.ValueType(bh.ScalarType.Uint)

    def make(cap, n_channels, child_of_float_attrs):
        g = TypedGen(int(rng.integers(1 << 30)))
        g.treadable, g.props, g.has_age = {}, [], True
        w = g.w
        init = []
        if child_of_float_attrs is None:
            init.append(bh.SetAttributeModifier(A.POSITION, g.texpr("f", 3, 2, "init").expr()))
        else:   # a child: position from the parent, one attribute computed from the parent's
            init.append(bh.InheritAttributeModifier(A.POSITION))
            src = child_of_float_attrs[int(rng.integers(len(child_of_float_attrs)))]
            init.append(bh.SetAttributeModifier(A.F32_3, (w.parent_attr(src) * w.lit(0.5) + w.rand(bh.ValueType(bh.ScalarType.Float))).expr()))
        init.append(bh.SetAttributeModifier(A.VELOCITY, g.texpr("f", 3, 2, "init").expr()))
        init.append(bh.SetAttributeModifier(A.AGE, w.lit(0.0).expr()))
        init.append(bh.SetAttributeModifier(A.LIFETIME, w.lit(0.05).uniform(w.lit(float(rng.uniform(0.1, 0.5)))).expr()))
        init.append(bh.SetAttributeModifier(A.F32_0, g.texpr("f", 1, 2, "init").expr()))
        update = []
        if rng.random() < 0.5:
            update.append(bh.AccelModifier(w.lit((0.0, -3.0, 0.0)).expr()))
        for ch in range(n_channels):
            cond = bh.EventEmitCondition.OnDie if rng.random() < 0.5 else bh.EventEmitCondition.Always
            k = rng.random()
            if k < 0.4:
                count = w.lit(bh.Value.u32(int(rng.integers(0, 4))))
            else:
                count = (w.rand(bh.ValueType(bh.ScalarType.Float)) * w.lit(float(rng.uniform(0.5, 3.9)))).cast(U)
            update.append(bh.EmitSpawnEventModifier(cond, count.expr(), ch))
        asset = bh.EffectAsset(cap, bh.SpawnerSettings.once(float(cap)), w.finish())
        for m in init:
            asset.init(m)
        for m in update:
            asset.update(m)
        return asset

    caps = [37, 300, 4096, 4200, 9000]
    n_ch = int(rng.integers(1, 4))
    specs = [EffectSpec(make(int(rng.choice(caps)), n_ch, None))]
    for ch in range(n_ch):
        grand = rng.random() < 0.4
        specs.append(EffectSpec(make(int(rng.choice(caps)) * 2, 1 if grand else 0, [A.F32_0]), parent=0, channel=ch,
                                event_capacity=int(rng.choice([16, 256, 5000, 1 << 16]))))
        if grand:
            specs.append(EffectSpec(make(int(rng.choice(caps)), 0, [A.F32_0, A.F32_3]), parent=len(specs) - 1, channel=0,
                                    event_capacity=int(rng.choice([64, 4096]))))
    return specs


def _run_system(seed, ctx, slot_order=False):
    import numpy as np
    from helpers import Frame, GpuSystem, OracleSystem, assert_same_system_state, frame_seed

    specs = _random_system(seed)
    g = GpuSystem(specs, ctx)
    o = OracleSystem(specs, slot_order=slot_order)
    rng = np.random.default_rng(seed + 5)
    try:
        root_cap = specs[0].asset.capacity
        for f in range(36):
            dt = 1 / 60 if rng.random() < 0.8 else 1 / 30
            spawn = root_cap if f == 0 else (int(rng.integers(0, root_cap // 2 + 2)) if rng.random() < 0.4 else 0)
            frames = [Frame(dt, spawn if i == 0 else 0, frame_seed(seed * 101 + f * 7 + i), time=f / 60) for i in range(len(specs))]
            g.step(frames)
            o.step(frames)
            if f % 6 == 5:
                assert_same_system_state(o.state(), g.state(), f"seed {seed} frame {f}")
    finally:
        g.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("jit", ["1", "0"])
@pytest.mark.parametrize("seed", list(range(300, 310)))
def test_fuzz_systems_gpu(ctx, seed, jit, monkeypatch):
    monkeypatch.setenv("HNB_JIT", jit)
    _run_system(seed, ctx)
