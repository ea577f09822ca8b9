"""TEST INFRASTRUCTURE: plays an effect from the WGSL text `generate_wgsl` emits (host/wgsl.cpp), with the interpreter of interp.py.

What the text does not contain - the bodies of vfx_init.wgsl / vfx_update.wgsl around the template slots, the PRNG of vfx_common.wgsl, the
indirect / dead lists - is restated here from the reference files directly (file:line below), in the canonical serial thread order of
SURVEY.md section 8(c): threads in increasing global id. Independent of oracle/ and of the product: nothing is imported from either.
"""
import numpy as np

from .interp import BOOL, F32, I32, U32, Interp, Mat4, Parser, Particle, Ptr, Struct, concretise

TAU = np.array([6.283185307179586476925286766559], F32)   # vfx_common.wgsl:261


# ---- PRNG (vfx_common.wgsl:263-343), vectorised over lanes; `seed` is a per-lane private variable ---------------------------------------
def pcg_hash(x):
    x = x.astype(np.uint64)
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return (((word >> 22) ^ word) & 0xFFFFFFFF).astype(U32)


def to_float01(u):
    return ((u & U32(0x007FFFFF)) | U32(0x3F800000)).view(F32) - F32(1.0)


def _advance(it, new_seed):
    """seed = ... under the lane mask (a call inside an `if` or after a `return` must not disturb the other lanes)"""
    g = it.globals
    g["seed"] = np.where(it.active(), new_seed, g["seed"])


def h_frand(it):
    s1 = pcg_hash(it.globals["seed"])
    _advance(it, s1)
    return to_float01(pcg_hash(s1))


def _frand_n(it, n):
    s = it.globals["seed"]
    cols = []
    for _ in range(n):
        s = pcg_hash(s)
        cols.append(to_float01(s))
    _advance(it, s)
    return np.stack(cols, axis=1)


def h_frand4(it):
    r0 = pcg_hash(it.globals["seed"])
    r1 = pcg_hash(r0)
    r2 = pcg_hash(r1)
    _advance(it, r2)
    x = to_float01(r0)
    y = to_float01(((r0 & U32(0xFF000000)) >> U32(8)) | (r1 & U32(0x0000FFFF)))
    z = to_float01(((r1 & U32(0xFFFF0000)) >> U32(8)) | (r2 & U32(0x000000FF)))
    w = to_float01(r2 >> U32(8))
    return np.stack([x, y, z, w], axis=1)


def _uniform(draw):
    def f(it, a, b):
        a, b = concretise(a, np.zeros(1, F32)), concretise(b, np.zeros(1, F32))
        r = draw(it)
        if a.ndim == 1 and r.ndim == 2:
            a, b = a[:, None], b[:, None]
        return a + r * (b - a)
    return f


def _normal(draw_v):
    def f(it, mean, std_dev):
        mean, std_dev = concretise(mean, np.zeros(1, F32)), concretise(std_dev, np.zeros(1, F32))
        u = h_frand(it)
        v = draw_v(it)
        with np.errstate(all="ignore"):
            r = np.sqrt(F32(-2.0) * np.log(u)).astype(F32)
            c = np.cos(TAU * v if v.ndim == 1 else TAU[:, None] * v).astype(F32)
        if v.ndim == 2:
            r = r[:, None]
        return mean + std_dev * r * c
    return f


HOOKS = {
    "frand": h_frand, "frand2": lambda it: _frand_n(it, 2), "frand3": lambda it: _frand_n(it, 3), "frand4": h_frand4,
    "rand_uniform_f": _uniform(h_frand), "rand_uniform_vec2": _uniform(lambda it: _frand_n(it, 2)),
    "rand_uniform_vec3": _uniform(lambda it: _frand_n(it, 3)), "rand_uniform_vec4": _uniform(h_frand4),
    "rand_normal_f": _normal(h_frand), "rand_normal_vec2": _normal(lambda it: _frand_n(it, 2)),
    "rand_normal_vec3": _normal(lambda it: _frand_n(it, 3)), "rand_normal_vec4": _normal(h_frand4),
}

ELEM_OF = {"Float": F32, "Int": I32, "Uint": U32, "Bool": BOOL}


class WgslEffect:
    """One effect instance driven by its WGSL text. attrs: [(name, numpy dtype, components)] in layout order; props: {name: default array}."""

    def __init__(self, wgsl, attrs, capacity, props=None, slot_base=0):
        self.w, self.capacity, self.slot_base = wgsl, capacity, slot_base
        self.attr_specs = attrs
        self.attrs = {name: np.zeros((capacity,) if n == 1 else (capacity, n), dt) for name, dt, n in attrs}
        self.props = {k: np.atleast_1d(np.asarray(v)) for k, v in (props or {}).items()}
        # effect_cache.rs:298-323: dead[i] = i; the dead list is a stack whose top is row alive_count
        self.list = np.zeros(0, U32)
        self.dead = np.arange(capacity, dtype=U32)
        self.alive_count = 0
        self.particle_counter = 0
        self.max_update = self.dead_count = self.spawned = 0
        self.ref_write_index = 0
        self.init_main = Parser(wgsl["init_code"]).parse_stmts_until_eof()
        self.init_fns = Parser(wgsl["init_extra"]).parse_functions()
        self.sim_space = Parser(wgsl["init_sim_space_transform"]).parse_stmts_until_eof()
        self.age = Parser(wgsl["age_code"]).parse_stmts_until_eof()
        self.reap = Parser(wgsl["reap_code"]).parse_stmts_until_eof()
        self.update_main = Parser(wgsl["update_code"]).parse_stmts_until_eof()
        self.update_fns = Parser(wgsl["update_extra"]).parse_functions()
        self.events = {}       # channel -> list of parent slots appended this frame (append_spawn_events_N, src/lib.rs:976-993)

    def set_property(self, name, value):
        self.props[name] = np.atleast_1d(np.asarray(value))

    def _globals(self, n, slots, seed, dt, time, transform, counter0=None):
        xf = np.asarray(transform if transform is not None else [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], F32).reshape(3, 4)
        # vfx_init.wgsl:157-164: transpose(mat4x4(row0, row1, row2, (0,0,0,1))): column j = (row0[j], row1[j], row2[j], [j == 3])
        cols = [np.array([[xf[0, j], xf[1, j], xf[2, j], 1.0 if j == 3 else 0.0]], F32) for j in range(4)]
        sim = Struct(delta_time=np.array([dt], F32), time=np.array([time], F32), virtual_delta_time=np.array([dt], F32), virtual_time=np.array([time], F32),
                     real_delta_time=np.array([dt], F32), real_time=np.array([time], F32))
        props = Struct(**{k: (v[None, :] if len(v) > 1 else v) for k, v in self.props.items()})    # a vector property: one row for all lanes
        pidx = (slots.astype(np.uint64) + self.slot_base).astype(U32)
        g = {"sim_params": sim, "tau": TAU, "properties": [props], "properties_array_index": np.array([0], U32), "transform": Mat4(cols),
             "particle_index": pidx, "seed": pcg_hash(pidx ^ U32(seed)),   # vfx_init.wgsl:152 / vfx_update.wgsl:138
             "effect_metadata": Ptr(Struct(base_child_index=np.array([0], U32)))}    # `let effect_metadata = &effect_metadatas[..]`
        if counter0 is not None:
            g["particle_counter"] = (counter0 + np.arange(n, dtype=np.uint64)).astype(U32)
        return g

    # ---- vfx_init.wgsl:101-196 -------------------------------------------------------------------------------------------------------------
    def init_pass(self, dt, spawn_count, seed, time=0.0, transform=None, parent=None, parent_events=None):
        gpu_spawned = parent is not None
        request = len(parent_events) if gpu_spawned else int(spawn_count)
        n = min(request, self.capacity - self.alive_count)       # max_spawn = capacity - alive_count
        self.spawned = n
        if n == 0:
            return
        slots = self.dead[self.alive_count:self.alive_count + n].copy()    # thread i pops dead row alive_count + i
        g = self._globals(n, slots, seed, dt, time, transform, counter0=self.particle_counter)
        particle = Particle({name: np.zeros((n,) if k == 1 else (n, k), dtp) for name, dtp, k in self.attr_specs})   # var particle = Particle();
        g["particle"] = particle
        if gpu_spawned:    # vfx_init.wgsl:166-171
            ps = np.asarray(parent_events[:n], dtype=np.int64)
            g["parent_particle"] = Particle({name: arr[ps] for name, arr in parent.attrs.items()})
            g["parent_particle_index"] = ps.astype(U32)
        it = Interp(n, g, self.init_fns, HOOKS)
        it.exec_block(self.init_main)
        if "prev" in particle.fields:
            particle.fields["prev"] = np.full(n, 0xFFFFFFFF, U32)
        if "next" in particle.fields:
            particle.fields["next"] = np.full(n, 0xFFFFFFFF, U32)
        if not gpu_spawned and not self.w["consume_gpu_spawn_events"]:
            it.exec_block(self.sim_space)
        for name, arr in particle.fields.items():
            self.attrs[name][slots] = np.broadcast_to(arr, self.attrs[name][slots].shape)
        self.list = np.concatenate([self.list, slots])
        self.alive_count += n
        self.particle_counter = (self.particle_counter + n) & 0xFFFFFFFF

    def _append_events(self, ch, slots):
        def f(it, _base_child_index, particle_index, count):
            count = np.broadcast_to(concretise(count), (it.n,))
            m = it.active()
            ev = self.events.setdefault(ch, [])
            # serial order: event e of the frame is the e-th (row, repeat) pair in list order; calls happen once per statement for all lanes
            for row in np.flatnonzero(m):
                ev.extend([int(slots[row])] * int(count[row]))
            return None
        return f


# ---- vfx_update.wgsl:105-167 ---------------------------------------------------------------------------------------------------------------
def _update_pass(self, dt, seed, time=0.0, transform=None):
    n = self.alive_count
    self.max_update = n
    self.events = {}
    self.ref_write_index ^= 1
    if n == 0:
        self.dead_count = 0
        return
    slots = self.list.copy()
    g = self._globals(n, slots, seed, dt, time, transform)
    particle = Particle({name: arr[slots].copy() for name, arr in self.attrs.items()})
    g["particle"] = particle
    it = Interp(n, g, self.update_fns, dict(HOOKS))
    for ch in range(4):
        it.hooks[f"append_spawn_events_{ch}"] = self._append_events(ch, slots)
    it.scopes.append({})
    for s in self.age + self.reap + self.update_main:      # {{AGE_CODE}} {{REAP_CODE}} {{UPDATE_CODE}} share main()'s scope
        it.exec(s)
    is_alive = np.broadcast_to(it.scopes[-1]["is_alive"], (n,)).copy()
    # {{WRITEBACK_CODE}}: every attribute but PREV / NEXT, for every thread (dead or alive) (lib.rs:1266-1281)
    for name, arr in particle.fields.items():
        if name in ("prev", "next"):
            continue
        self.attrs[name][slots] = np.broadcast_to(arr, self.attrs[name][slots].shape)
    # vfx_update.wgsl:148-166 under serial thread order: stable compaction; the k-th casualty lands on dead row n - 1 - k
    casualties = slots[~is_alive]
    for k, slot in enumerate(casualties):
        self.dead[n - 1 - k] = slot
    self.list = slots[is_alive]
    self.alive_count = int(is_alive.sum())
    self.dead_count = n - self.alive_count


WgslEffect.update_pass = _update_pass
