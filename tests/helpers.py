"""Shared test plumbing: three executors of the same frame script with one interface.

  OracleRunner  oracle/ (C restatement of the reference's WGSL semantics)     -- the checker
  CpuVmRunner   tests/cpu_vm (product interpreters compiled for the host)     -- CPU-only lowering check
  GpuRunner     the product: lowering -> C ABI -> HIP kernels                 -- what ships
"""
import ctypes as C
import os

import numpy as np

import bevy_hanabi_amd as bh
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = bh.Attribute

IDENTITY = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)


def translation(x, y, z):
    t = IDENTITY.copy()
    t[3], t[7], t[11] = x, y, z
    return t


def frame_seed(f, base=0xC0FFEE):
    return oracle.pcg_hash(base + f)


class Frame:
    def __init__(self, dt=1.0 / 60.0, spawn=0, seed=0, transform=None, time=0.0, props=None):
        self.dt, self.spawn, self.seed, self.transform, self.time, self.props = dt, spawn, seed, transform, time, props or {}


def stored_attrs(asset):
    return [a for a in asset.particle_layout() if a.id >= 2]


class OracleRunner:
    name = "oracle"

    def __init__(self, asset, slot_base=0, omp=False, libm=False):
        self.asset = asset
        self.fx = oracle.OracleEffect(bh.serialize_asset(asset), slot_base, omp=omp, libm=libm)

    def step(self, fr: Frame):
        for k, v in fr.props.items():
            self.fx.set_property(k, v)
        self.fx.step(fr.dt, fr.spawn, fr.seed, time=fr.time, transform=fr.transform)

    def state(self):
        return {"counters": self.fx.counters(), "alive": self.fx.alive_list(), "dead": self.fx.dead_list(),
                "attrs": {a.name: self.fx.read_attr(a.id).view(np.uint32) for a in stored_attrs(self.asset)}}


_cvm = None


def _cvm_lib():
    global _cvm
    if _cvm is None:
        lib = C.CDLL(os.path.join(ROOT, "tests", "cpu_vm", "libcpu_vm.so"))
        lib.cvm_create.restype = C.c_void_p
        lib.cvm_create.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        lib.cvm_destroy.argtypes = [C.c_void_p]
        lib.cvm_streamable.argtypes = [C.c_void_p]
        lib.cvm_set_property.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32]
        lib.cvm_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        lib.cvm_counters.argtypes = [C.c_void_p, C.c_void_p]
        lib.cvm_read_attr.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.cvm_read_alive_list.argtypes = [C.c_void_p, C.c_void_p]
        lib.cvm_read_dead_list.argtypes = [C.c_void_p, C.c_void_p]
        _cvm = lib
    return _cvm


class CpuVmRunner:
    name = "cpu_vm"

    def __init__(self, asset, slot_base=0, force_generic=False):
        self.asset = asset
        self.blob = bh.lower(asset)
        bh.validate_program(self.blob)
        self.lib = _cvm_lib()
        self.h = self.lib.cvm_create(self.blob, len(self.blob), slot_base)
        assert self.h, "cpu_vm rejected the program blob"
        self.force_generic = force_generic
        self.capacity = asset.capacity

    @property
    def streamable(self):
        return bool(self.lib.cvm_streamable(self.h))

    def step(self, fr: Frame):
        for k, v in fr.props.items():
            w = np.atleast_1d(np.asarray(v))
            w = w.astype(np.float32).view(np.uint32) if w.dtype.kind == "f" else w.astype(np.uint32)
            w = np.ascontiguousarray(w)
            assert self.lib.cvm_set_property(self.h, k.encode(), w.ctypes.data, len(w)) == 0
        sim = np.array([fr.time, fr.dt, fr.time, fr.dt, fr.time, fr.dt], dtype=np.float32)
        xf = None if fr.transform is None else np.ascontiguousarray(np.asarray(fr.transform, dtype=np.float32))
        self.lib.cvm_step(self.h, sim.ctypes.data, fr.spawn, fr.seed & 0xFFFFFFFF, None if xf is None else xf.ctypes.data, int(self.force_generic))

    def state(self):
        c = np.zeros(8, dtype=np.uint32)
        self.lib.cvm_counters(self.h, c.ctypes.data)
        keys = ["capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index", "particle_counter", "instance_count", "dead_count"]
        counters = dict(zip(keys, (int(x) for x in c)))
        alive = np.zeros(counters["alive_count"], dtype=np.uint32)
        dead = np.zeros(self.capacity - counters["alive_count"], dtype=np.uint32)
        if len(alive):
            self.lib.cvm_read_alive_list(self.h, alive.ctypes.data)
        if len(dead):
            self.lib.cvm_read_dead_list(self.h, dead.ctypes.data)
        attrs = {}
        for a in stored_attrs(self.asset):
            buf = np.zeros((self.capacity, a.value_type.count), dtype=np.uint32)
            assert self.lib.cvm_read_attr(self.h, a.id, buf.ctypes.data) == a.value_type.count
            attrs[a.name] = buf
        return {"counters": counters, "alive": alive, "dead": dead, "attrs": attrs}

    def __del__(self):
        try:
            self.lib.cvm_destroy(self.h)
        except Exception:
            pass


class GpuRunner:
    name = "gpu"

    def __init__(self, asset, slot_base=0, ctx=None):
        self.asset = asset
        self.blob = bh.lower(asset)
        self.ctx = ctx or bh.Context(0)
        self.prog = self.ctx.create_program(self.blob)
        self.fx = self.prog.create_effect(slot_base)

    def step(self, fr: Frame):
        for k, v in fr.props.items():
            self.fx.set_property(k, v)
        self.ctx.frame_begin(fr.dt, fr.time)
        self.fx.set_frame(fr.spawn, fr.seed, fr.transform)
        self.ctx.simulate()

    def state(self):
        m = self.fx.metadata()
        keys = ["capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index", "particle_counter", "instance_count", "dead_count"]
        return {"counters": {k: m[k] for k in keys}, "alive": self.fx.alive_list(), "dead": self.fx.dead_list(),
                "attrs": {a.name: self.fx.read_attr(a.id).view(np.uint32) for a in stored_attrs(self.asset)}}


def _is_float_attr(name):
    a = bh.Attribute.from_name(name)
    return a is not None and a.value_type.elem == bh.ScalarType.Float


def _both_nan(a, b):
    """Float components that are NaN on both sides. WGSL leaves the bit pattern of a NaN unspecified, and it is not
    reproducible even between two host compilers (x86 propagates the payload of whichever operand the code generator put
    first, and produces a negative default NaN where gfx950 produces a positive one), so NaN == NaN here; everything
    else, including the sign of zero and every denormal, is compared by bits."""
    isnan = lambda x: (x & 0x7F800000 == 0x7F800000) & (x & 0x007FFFFF != 0)
    return isnan(a) & isnan(b)


def assert_same_state(ref, got, what=""):
    """Bit-exact comparison of counters, alive/dead lists and every attribute plane."""
    assert ref["counters"] == got["counters"], f"{what}: counters differ\n ref {ref['counters']}\n got {got['counters']}"
    np.testing.assert_array_equal(ref["alive"], got["alive"], err_msg=f"{what}: alive list differs")
    np.testing.assert_array_equal(ref["dead"], got["dead"], err_msg=f"{what}: dead list differs")
    assert ref["attrs"].keys() == got["attrs"].keys()
    for k in ref["attrs"]:
        a, b = ref["attrs"][k], got["attrs"][k]
        if not np.array_equal(a, b):
            differs = a != b
            if _is_float_attr(k):
                differs &= ~_both_nan(a, b)
                if not differs.any():
                    continue
            bad = np.argwhere(differs)
            i = bad[0][0]
            raise AssertionError(f"{what}: attribute '{k}' differs at {len(bad)} components; first slot {i}: "
                                 f"ref {a[i].view(np.float32)} ({a[i]}) got {b[i].view(np.float32)} ({b[i]})")


def run_script(runner, frames, check_against=None, every=1):
    for i, fr in enumerate(frames):
        runner.step(fr)
        if check_against is not None:
            check_against.step(fr)
            if (i + 1) % every == 0 or i == len(frames) - 1:
                assert_same_state(check_against.state(), runner.state(), f"frame {i}")
    return runner.state()


# ---- systems of linked effects (GPU spawn events) ---------------------------------------------------------
class EffectSpec:
    """One effect of a system: asset, optional parent (index into the system), event channel and capacity."""

    def __init__(self, asset, parent=None, channel=0, event_capacity=256):
        self.asset, self.parent, self.channel, self.event_capacity = asset, parent, channel, event_capacity


class OracleSystem:
    name = "oracle"

    def __init__(self, specs, omp=False, slot_order=False):
        self.specs = specs
        self.fx = [oracle.OracleEffect(bh.serialize_asset(s.asset), omp=omp) for s in specs]
        for fx in self.fx:
            fx.set_list_order(slot_order)
        for s, fx in zip(specs, self.fx):
            if s.parent is not None:
                fx.set_parent(self.fx[s.parent], s.channel, s.event_capacity)

    def step(self, frames):
        """frames: one Frame per effect. Every init pass (parents first: list order), then every update pass."""
        for fx, fr in zip(self.fx, frames):
            for k, v in fr.props.items():
                fx.set_property(k, v)
            fx.init_pass(fr.dt, fr.spawn, fr.seed, time=fr.time, transform=fr.transform)
        for fx, fr in zip(self.fx, frames):
            fx.update_pass(fr.dt, fr.seed, time=fr.time, transform=fr.transform)

    def state(self):
        out = []
        for s, fx in zip(self.specs, self.fx):
            out.append({"counters": fx.counters(), "alive": fx.alive_list(), "dead": fx.dead_list(),
                        "attrs": {a.name: fx.read_attr(a.id).view(np.uint32) for a in stored_attrs(s.asset)}})
        return out


class GpuSystem:
    name = "gpu"

    def __init__(self, specs, ctx):
        self.specs, self.ctx = specs, ctx
        self.progs = [ctx.create_program(bh.lower(s.asset)) for s in specs]
        self.fx = [p.create_effect() for p in self.progs]
        for s, fx in zip(specs, self.fx):
            if s.parent is not None:
                fx.set_parent(self.fx[s.parent], s.channel, s.event_capacity)

    def step(self, frames):
        self.ctx.frame_begin(frames[0].dt, frames[0].time)
        for fx, fr in zip(self.fx, frames):
            for k, v in fr.props.items():
                fx.set_property(k, v)
            fx.set_frame(fr.spawn, fr.seed, fr.transform)
        self.ctx.simulate()

    def state(self):
        keys = ["capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index", "particle_counter", "instance_count", "dead_count"]
        out = []
        for s, fx in zip(self.specs, self.fx):
            m = fx.metadata()
            out.append({"counters": {k: m[k] for k in keys}, "alive": fx.alive_list(), "dead": fx.dead_list(),
                        "attrs": {a.name: fx.read_attr(a.id).view(np.uint32) for a in stored_attrs(s.asset)}})
        return out

    def destroy(self):
        for p in self.progs:
            p.destroy()


def assert_same_system_state(ref, got, what=""):
    assert len(ref) == len(got)
    for i, (r, g) in enumerate(zip(ref, got)):
        assert_same_state(r, g, f"{what} effect #{i}")


# ---- GPU evaluation of the hanabi-math builtins (tests/test_gpu_scale.py, tools/warm_jit_cache.py) -------------
def math_probe_asset(capacity):
    """One effect whose UPDATE evaluates every transcendental builtin of the expression API on per-particle inputs:
         F32X4_0 = (sin, cos, tan, atan)(F32_0)      F32X4_1 = (asin, acos)(F32_1), (exp, exp2)(F32_2)
         F32X4_2 = (log, log2, sqrt, inverseSqrt)(F32_3)      F32X2_1 = (atan2(F32X2_0.x, F32X2_0.y), 0)
    The test writes the input planes through the ABI, runs one frame and reads the outputs back."""
    w = bh.ExprWriter()
    a, b, c, d = (w.attr(x) for x in (A.F32_0, A.F32_1, A.F32_2, A.F32_3))
    p = w.attr(A.F32X2_0)
    out0 = a.sin().vec3(a.cos(), a.tan()).vec4_xyz_w(a.atan())
    out1 = b.asin().vec3(b.acos(), c.exp()).vec4_xyz_w(c.exp2())
    out2 = d.log().vec3(d.log2(), d.sqrt()).vec4_xyz_w(d.inverse_sqrt())
    out3 = p.x().atan2(p.y()).vec2(w.lit(0.0))
    mods_init = [bh.SetAttributeModifier(A.POSITION, w.lit((0.0, 0.0, 0.0)).expr())]
    mods_update = [bh.SetAttributeModifier(A.F32X4_0, out0.expr()), bh.SetAttributeModifier(A.F32X4_1, out1.expr()),
                   bh.SetAttributeModifier(A.F32X4_2, out2.expr()), bh.SetAttributeModifier(A.F32X2_1, out3.expr())]
    asset = bh.EffectAsset(capacity, bh.SpawnerSettings.once(float(capacity)), w.finish())
    for m in mods_init:
        asset.init(m)
    for m in mods_update:
        asset.update(m)
    return asset
