"""ctypes wrapper around oracle/libhanabi_oracle{,_omp}.so (TEST INFRASTRUCTURE).

PARITY STATUS: control-plane integers are pinned by the reference's golden vectors; particle
float state is unpinned (no reference test asserts one) — see hanabi_oracle.c header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

ATTR_COMPONENTS = [1, 1, 3, 3, 1, 1, 1, 4, 1, 1, 2, 3, 1, 1, 3, 3, 3, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 1, 1, 1, 1, 1]
ATTR_IS_FLOAT = [0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0] + [1] * 16 + [0] * 5


def build(force=False):
    """Compile the oracle (building the checker is not using it)."""
    targets = ["libhanabi_oracle.so", "libhanabi_oracle_omp.so", "libhanabi_oracle_libm.so"]
    srcs = [os.path.join(_DIR, f) for f in ("hanabi_oracle.c", "oracle_math.h")]
    stale = lambda t: not os.path.exists(t) or any(os.path.getmtime(s) > os.path.getmtime(t) for s in srcs)
    if force or any(stale(os.path.join(_DIR, t)) for t in targets):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))


def _lib(omp=False, libm=False):
    """omp: OpenMP over particles (same arithmetic). libm: the independent flavour whose transcendental builtins go through
    the host libm (oracle_math.h, ORACLE_LIBM) — tolerance checks only, it is not bit-compatible with the product."""
    name = "libhanabi_oracle_libm.so" if libm else "libhanabi_oracle_omp.so" if omp else "libhanabi_oracle.so"
    if name not in _LIBS:
        path = os.path.join(_DIR, name)
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.hor_last_error.restype = C.c_char_p
        lib.hor_asset_parse.restype = C.c_void_p
        lib.hor_asset_parse.argtypes = [C.c_char_p, C.c_size_t]
        lib.hor_asset_free.argtypes = [C.c_void_p]
        lib.hor_effect_create.restype = C.c_void_p
        lib.hor_effect_create.argtypes = [C.c_void_p, C.c_uint32]
        lib.hor_effect_free.argtypes = [C.c_void_p]
        lib.hor_effect_set_property.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32]
        lib.hor_effect_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.hor_effect_init_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.hor_effect_update_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.hor_effect_set_parent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.hor_effect_event_count.restype = C.c_uint32
        lib.hor_effect_event_count.argtypes = [C.c_void_p, C.c_uint32]
        lib.hor_effect_read_events.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.hor_effect_set_list_order.argtypes = [C.c_void_p, C.c_int]
        lib.hor_effect_alive_count.restype = C.c_uint32
        lib.hor_effect_alive_count.argtypes = [C.c_void_p]
        lib.hor_effect_counters.argtypes = [C.c_void_p, C.c_void_p]
        lib.hor_effect_error.restype = C.c_char_p
        lib.hor_effect_error.argtypes = [C.c_void_p]
        lib.hor_effect_attr_components.argtypes = [C.c_void_p, C.c_uint32]
        lib.hor_effect_read_attr.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.hor_effect_write_attr.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.hor_effect_read_alive_list.argtypes = [C.c_void_p, C.c_void_p]
        lib.hor_effect_read_dead_list.argtypes = [C.c_void_p, C.c_void_p]
        lib.hor_pcg_hash.restype = C.c_uint32
        lib.hor_pcg_hash.argtypes = [C.c_uint32]
        lib.hor_to_float01.restype = C.c_float
        lib.hor_to_float01.argtypes = [C.c_uint32]
        lib.hor_round_literal.restype = C.c_float
        lib.hor_round_literal.argtypes = [C.c_float]
        lib.hor_math1.restype = C.c_float
        lib.hor_math1.argtypes = [C.c_int, C.c_float]
        lib.hor_math1d.restype = C.c_double
        lib.hor_math1d.argtypes = [C.c_int, C.c_double]
        lib.hor_math2.restype = C.c_float
        lib.hor_math2.argtypes = [C.c_int, C.c_float, C.c_float]
        lib.hor_spawner_tick.restype = C.c_uint32
        _LIBS[name] = lib
    return _LIBS[name]


class OracleError(RuntimeError):
    pass


class OracleEffect:
    """One effect instance simulated by the serial-order CPU restatement."""

    def __init__(self, asset_blob: bytes, slot_base=0, omp=False, libm=False):
        self._lib = _lib(omp, libm)
        self._asset = self._lib.hor_asset_parse(asset_blob, len(asset_blob))
        if not self._asset:
            raise OracleError(self._lib.hor_last_error().decode())
        self._fx = self._lib.hor_effect_create(self._asset, slot_base)
        self.capacity = self.counters()["capacity"]

    def close(self):
        if getattr(self, "_fx", None):
            self._lib.hor_effect_free(self._fx)
            self._lib.hor_asset_free(self._asset)
            self._fx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_property(self, name, values):
        v = np.atleast_1d(np.asarray(values))
        words = v.astype(np.float32).view(np.uint32) if v.dtype.kind == "f" else v.astype(np.uint32)
        words = np.ascontiguousarray(words)
        rc = self._lib.hor_effect_set_property(self._fx, name.encode(), words.ctypes.data, len(words))
        if rc != 0:
            raise OracleError(f"set_property({name}) failed: {rc}")

    def step(self, dt, spawn_count, seed, time=0.0, transform=None, sim=None):
        s = np.array(sim if sim is not None else [time, dt, time, dt, time, dt], dtype=np.float32)
        xf = None if transform is None else np.ascontiguousarray(np.asarray(transform, dtype=np.float32).reshape(12))
        rc = self._lib.hor_effect_step(self._fx, s.ctypes.data, int(spawn_count), int(seed) & 0xFFFFFFFF,
                                       None if xf is None else xf.ctypes.data)
        if rc != 0:
            raise OracleError(self._lib.hor_effect_error(self._fx).decode())

    def _args(self, dt, time, transform, sim):
        s = np.array(sim if sim is not None else [time, dt, time, dt, time, dt], dtype=np.float32)
        xf = None if transform is None else np.ascontiguousarray(np.asarray(transform, dtype=np.float32).reshape(12))
        return s, xf

    def init_pass(self, dt, spawn_count, seed, time=0.0, transform=None, sim=None):
        """vfx_init only. Systems with parent/child links run every init pass (parents first), then every update pass."""
        s, xf = self._args(dt, time, transform, sim)
        if self._lib.hor_effect_init_pass(self._fx, s.ctypes.data, int(spawn_count), int(seed) & 0xFFFFFFFF, None if xf is None else xf.ctypes.data):
            raise OracleError(self._lib.hor_effect_error(self._fx).decode())

    def update_pass(self, dt, seed, time=0.0, transform=None, sim=None):
        s, xf = self._args(dt, time, transform, sim)
        if self._lib.hor_effect_update_pass(self._fx, s.ctypes.data, int(seed) & 0xFFFFFFFF, None if xf is None else xf.ctypes.data):
            raise OracleError(self._lib.hor_effect_error(self._fx).decode())

    def set_list_order(self, by_slot):
        """Second canonical schedule: survivors of an update are left in increasing slot order."""
        self._lib.hor_effect_set_list_order(self._fx, int(bool(by_slot)))

    def set_parent(self, parent, channel, event_capacity=256):
        """EffectParent: this effect's init consumes the spawn events `parent` appends to `channel`."""
        if self._lib.hor_effect_set_parent(self._fx, parent._fx, int(channel), int(event_capacity)):
            raise OracleError("set_parent failed")
        self._parent = parent  # keep alive

    def events(self, channel):
        """(event_count, stored particle indices) of one of this effect's child channels."""
        n = int(self._lib.hor_effect_event_count(self._fx, int(channel)))
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self._lib.hor_effect_read_events(self._fx, int(channel), out.ctypes.data)
        return n, out

    def alive_count(self):
        return int(self._lib.hor_effect_alive_count(self._fx))

    def counters(self):
        out = np.zeros(8, dtype=np.uint32)
        self._lib.hor_effect_counters(self._fx, out.ctypes.data)
        keys = ["capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index", "particle_counter", "instance_count", "dead_count"]
        return dict(zip(keys, (int(x) for x in out)))

    def read_attr(self, attr_id):
        n = self._lib.hor_effect_attr_components(self._fx, int(attr_id))
        if n == 0:
            raise OracleError(f"attribute {attr_id} not in layout")
        out = np.zeros((self.capacity, n), dtype=np.uint32)
        self._lib.hor_effect_read_attr(self._fx, int(attr_id), out.ctypes.data)
        return out.view(np.float32) if ATTR_IS_FLOAT[int(attr_id)] else out

    def write_attr(self, attr_id, array):
        a = np.ascontiguousarray(array)
        n = self._lib.hor_effect_attr_components(self._fx, int(attr_id))
        assert n and a.nbytes == self.capacity * n * 4
        self._lib.hor_effect_write_attr(self._fx, int(attr_id), a.ctypes.data)

    def alive_list(self):
        out = np.zeros(self.alive_count(), dtype=np.uint32)
        if len(out):
            self._lib.hor_effect_read_alive_list(self._fx, out.ctypes.data)
        return out

    def dead_list(self):
        out = np.zeros(self.capacity - self.alive_count(), dtype=np.uint32)
        if len(out):
            self._lib.hor_effect_read_dead_list(self._fx, out.ctypes.data)
        return out


class _Spawner(C.Structure):
    _fields_ = [("count", C.c_float), ("spawn_duration", C.c_float), ("period", C.c_float), ("cycle_count", C.c_uint32), ("active", C.c_int),
                ("cycle_time", C.c_float), ("sampled_spawn_duration", C.c_float), ("sampled_period", C.c_float), ("sampled_count", C.c_float),
                ("spawn_remainder", C.c_float), ("completed_cycle_count", C.c_uint32), ("spawn_count", C.c_uint32)]


class OracleSpawner:
    """EffectSpawner::tick restated in C (spawn.rs:838-921), CpuValue::Single only."""

    def __init__(self, count, spawn_duration, period, cycle_count, starts_active=True, emit_on_start=True):
        self._lib = _lib()
        self.s = _Spawner()
        self._lib.hor_spawner_init(C.byref(self.s), C.c_float(count), C.c_float(spawn_duration), C.c_float(period), C.c_uint32(cycle_count),
                                   C.c_int(starts_active), C.c_int(emit_on_start))

    def tick(self, dt):
        return int(self._lib.hor_spawner_tick(C.byref(self.s), C.c_float(dt)))

    def reset(self):
        self._lib.hor_spawner_reset(C.byref(self.s))

    @property
    def active(self):
        return bool(self.s.active)

    @active.setter
    def active(self, v):
        self.s.active = int(bool(v))


def pcg_hash(x):
    return int(_lib().hor_pcg_hash(C.c_uint32(x & 0xFFFFFFFF)))


def to_float01(u):
    return float(_lib().hor_to_float01(C.c_uint32(u & 0xFFFFFFFF)))


def frand_kat(seed, which):
    state = C.c_uint32(0)
    out = (C.c_float * 4)()
    _lib().hor_frand_kat(C.c_uint32(seed & 0xFFFFFFFF), C.byref(state), out, C.c_int(which))
    return int(state.value), [float(out[i]) for i in range(max(which, 1))]


def round_literal(x):
    return float(_lib().hor_round_literal(C.c_float(x)))


def math1(fn, x):
    return float(_lib().hor_math1(C.c_int(fn), C.c_float(x)))


def math1d(fn, x):
    """The binary64 kernel behind math1(fn, .), before the final rounding to binary32."""
    return float(_lib().hor_math1d(C.c_int(fn), C.c_double(x)))


def math2(fn, x, y):
    return float(_lib().hor_math2(C.c_int(fn), C.c_float(x), C.c_float(y)))


def k2_indirect(meta_rows):
    """vfx_indirect.wgsl restated; meta_rows: list of [capacity, alive_count, max_update, max_spawn, write_index]."""
    meta = np.array(meta_rows, dtype=np.uint32)
    n = len(meta)
    prefix = np.zeros(n, dtype=np.uint32)
    inst = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    pong = np.zeros(n, dtype=np.uint32)
    _lib().hor_k2_indirect(C.c_uint32(n), C.c_void_p(meta.ctypes.data), C.c_uint32(meta.shape[1]), C.c_void_p(prefix.ctypes.data),
                           C.c_void_p(inst.ctypes.data), C.c_void_p(pong.ctypes.data))
    return meta, prefix, inst, pong


def k3_prefix_sum(prefix, batches):
    """vfx_prefix_sum.wgsl restated; batches: list of (offset, count)."""
    ps = np.array(prefix, dtype=np.uint32)
    off = np.array([b[0] for b in batches], dtype=np.uint32)
    cnt = np.array([b[1] for b in batches], dtype=np.uint32)
    tot = np.zeros(len(batches), dtype=np.uint32)
    dx = np.zeros(len(batches), dtype=np.uint32)
    _lib().hor_k3_prefix_sum(C.c_uint32(len(batches)), C.c_void_p(off.ctypes.data), C.c_void_p(cnt.ctypes.data), C.c_void_p(ps.ctypes.data),
                             C.c_void_p(tot.ctypes.data), C.c_void_p(dx.ctypes.data))
    return ps, tot, dx


def find_location(prefix, offset, count, index):
    ps = np.array(prefix, dtype=np.uint32)
    out = np.zeros(3, dtype=np.uint32)
    _lib().hor_find_location(C.c_void_p(ps.ctypes.data), C.c_uint32(offset), C.c_uint32(count), C.c_uint32(index), C.c_void_p(out.ctypes.data))
    return tuple(int(x) for x in out)


def omp_threads():
    return int(_lib(True).hor_omp_threads())


# ---- tuned CPU port of the lowered streaming update (cpu_soa.c): bench.py's cpu_baseline -------------------
HCS_AGE_TICK, HCS_VEL_SCALE, HCS_VEL_ADD, HCS_EULER = 1, 2, 3, 4
_SOA = {}


class _HcsOp(C.Structure):
    _fields_ = [("op", C.c_uint32), ("v", C.c_float * 3)]


def _host_tag():
    """cpu_soa.c is compiled with -march=native: one library per host CPU model (the in-tree .so files travel to the GPU box)."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    return hashlib.sha1(model.encode()).hexdigest()[:10]


def build_cpu_soa():
    path = os.path.join(_DIR, f"libhanabi_cpu_soa_{_host_tag()}.so")
    src = os.path.join(_DIR, "cpu_soa.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-std=gnu11", "-O3", "-march=native", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                               "-Wall", "-shared", src, "-o", path])
    return path


def _soa():
    if "lib" not in _SOA:
        lib = C.CDLL(build_cpu_soa())
        lib.hcs_threads.restype = C.c_int
        lib.hcs_set_threads.argtypes = [C.c_int]
        lib.hcs_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        lib.hcs_update.restype = C.c_uint64
        lib.hcs_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
        _SOA["lib"] = lib
    return _SOA["lib"]


class CpuSoaEffect:
    """Packed SoA planes (position, velocity, age, lifetime + one alive byte per slot) updated by cpu_soa.c."""

    def __init__(self, position, velocity, age, lifetime, alive):
        self._lib = _soa()
        n = len(age)
        self.n = n
        # parallel first-touch copies: pages land on the NUMA node of the thread that updates them
        self.pos, self.vel = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        self.age, self.life, self.alive = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.uint8)
        for dst, src, bpp in ((self.pos, position, 12), (self.vel, velocity, 12), (self.age, age, 4), (self.life, lifetime, 4), (self.alive, alive, 1)):
            s = np.ascontiguousarray(src, dtype=dst.dtype).reshape(dst.shape)
            self._lib.hcs_copy(dst.ctypes.data, s.ctypes.data, n, bpp)

    @staticmethod
    def threads():
        return int(_soa().hcs_threads())

    @staticmethod
    def set_threads(n):
        _soa().hcs_set_threads(int(n))

    def update(self, ops):
        """ops: [(HCS_*, (v0, v1, v2))]; returns the number of particles that died."""
        arr = (_HcsOp * len(ops))()
        for i, (op, v) in enumerate(ops):
            arr[i].op = op
            vv = list(np.asarray(v, dtype=np.float32).reshape(-1)) + [0.0, 0.0, 0.0]
            for c in range(3):
                arr[i].v[c] = float(vv[c])
        return int(self._lib.hcs_update(self.pos.ctypes.data, self.vel.ctypes.data, self.age.ctypes.data, self.life.ctypes.data,
                                        self.alive.ctypes.data, self.n, C.byref(arr), len(ops)))
