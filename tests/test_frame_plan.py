"""The proofs hnb_simulate's launch sequence rests on (bevy_hanabi_amd/csrc/hnb_plan.h), one by one, on synthetic program state and
frame inputs - no device. Each proof is a pure function (facts fixed at program creation, the history it carries,
this frame's inputs) -> decision; tests/cpu_plan builds the same header for the host. What the DEVICE does with a decision (and that a
wrong one raises HnbEffectMetadata::fault) is the business of the -m gpu tests; here: every premise, and every way it can be violated."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


class Row(C.Structure):
    _fields_ = [("simulated", C.c_uint32), ("has_parent", C.c_uint32), ("spawn_count", C.c_uint32), ("event_capacity", C.c_uint32), ("ublock", C.c_uint32 * 8)]


class MergeRow(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("independent", "total_chunks", "init_blocks", "init_len", "update_len", "wide_file", "update_streams", "age_cohort")]


@pytest.fixture(scope="module")
def lib():
    path = os.path.join(ROOT, "tests", "cpu_plan", "libcpu_plan.so")
    lib = C.CDLL(path)
    lib.cpl_skip_new.restype = C.c_void_p
    lib.cpl_skip_new.argtypes = [C.c_int, C.c_uint32]
    lib.cpl_skip_free.argtypes = [C.c_void_p]
    lib.cpl_skip_mark_dirty.argtypes = [C.c_void_p]
    lib.cpl_skip_last_dirty.argtypes = [C.c_void_p]
    lib.cpl_skip_last_dirty.restype = C.c_uint32
    lib.cpl_skip_step.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Row), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    lib.cpl_ribbon_new.restype = C.c_void_p
    lib.cpl_ribbon_new.argtypes = [C.c_int] * 4 + [C.c_uint32] * 4
    lib.cpl_ribbon_free.argtypes = [C.c_void_p]
    lib.cpl_ribbon_sorted.argtypes = [C.c_void_p]
    lib.cpl_ribbon_host_write.argtypes = [C.c_void_p, C.c_int]
    lib.cpl_ribbon_step.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Row), C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    lib.cpl_horizon_usable.argtypes = [C.c_int, C.c_uint32, C.POINTER(Row), C.c_uint32]
    lib.cpl_slot_init.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(Row), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.cpl_init_grid.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    lib.cpl_init_grid.restype = C.c_uint32
    lib.cpl_event_grid.argtypes = [C.c_uint32, C.c_uint32]
    lib.cpl_event_grid.restype = C.c_uint32
    lib.cpl_stream_hints.argtypes = [C.c_uint64, C.c_uint32]
    lib.cpl_store_hints.argtypes = [C.c_uint64, C.c_uint32]
    lib.cpl_merge.argtypes = [C.POINTER(MergeRow), C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int32)]
    lib.cpl_partition_init_passes.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    lib.cpl_split_uncovered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    lib.cpl_set_lookup_new.restype = C.c_void_p
    lib.cpl_set_lookup_free.argtypes = [C.c_void_p]
    lib.cpl_set_lookup_reset_tried.argtypes = [C.c_void_p]
    lib.cpl_set_lookup_due.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64]
    return lib


def rows(*specs):
    """spec: dict(simulated=1, has_parent=0, spawn=0, evcap=0, u={operand: float or raw bits (int)})"""
    arr = (Row * len(specs))()
    for r, sp in zip(arr, specs):
        r.simulated, r.has_parent, r.spawn_count, r.event_capacity = sp.get("simulated", 1), sp.get("has_parent", 0), sp.get("spawn", 0), sp.get("evcap", 0)
        for k, v in sp.get("u", {}).items():
            r.ublock[k] = v if isinstance(v, int) else bits(v)
    return arr, len(specs)


DT = 1.0 / 60.0
NONE = 0xFFFFFFFF


# ---- prove_skip_lists -------------------------------------------------------------------------------------------------------------------
class SkipRunner:
    def __init__(self, lib, eligible=1):
        self.lib, self.h, self.f = lib, lib.cpl_skip_new(eligible, 3), 0

    def frame(self, *specs, tag=NONE, bound=0.0, option=1):
        arr, n = rows(*specs)
        d = self.lib.cpl_skip_step(self.h, self.f, arr, n, tag, bits(bound), option)
        self.f += 1
        return bool(d)


def test_skip_lists_premises_one_by_one(lib):
    tick = {"u": {3: DT}}
    r = SkipRunner(lib)
    r.frame(dict(spawn=1000, **tick))                            # F0: burst
    r.frame(tick)                                                # F1
    assert r.frame(tick, tag=0, bound=0.5)                       # F2: bound of frame 0 = 0.5 s, ticks of frames 1..2 = 2/60: list-free
    assert r.frame(tick, tag=1, bound=0.5 - DT)
    assert not r.frame(tick, tag=1, bound=2.9 * DT)              # F4: ticks of frames 2..4 = 3/60 > bound: the lists run
    assert r.frame(tick, tag=1, bound=4.01 * DT)                 # F5: 4 ticks just fit
    assert not r.frame(tick, tag=NONE, bound=10.0)               # nothing published
    assert not r.frame(tick, tag=7, bound=10.0)                  # F7: a tag that is not in the past
    assert not r.frame(tick, tag=6, bound=10.0, option=0)        # the option is off
    assert not r.frame(dict(spawn=1, **tick), tag=7, bound=10.0)  # F9: a spawn - this frame needs its lists ...
    assert lib.cpl_skip_last_dirty(r.h) == 9
    assert not r.frame(tick, tag=8, bound=10.0)                  # F10: ... and a bound from before the spawn does not cover the new particle
    assert r.frame(tick, tag=9, bound=10.0)                      # F11: the bound computed in the spawn's frame does
    assert not r.frame(tick, tag=11, bound=float("nan"))         # a NaN bound
    assert not r.frame(tick, tag=11, bound=-1.0)                 # a negative one (sign bit: not a valid bound)
    # an unsimulated instance is ignored, a child instance (spawned by events the host cannot count) ends it
    assert r.frame(tick, dict(simulated=0, spawn=5, u={3: 123.0}), tag=12, bound=10.0)
    assert not r.frame(tick, dict(has_parent=1, **tick), tag=13, bound=10.0)
    # instances ticking differently: no common clock
    assert not r.frame(tick, {"u": {3: 2 * DT}}, tag=14, bound=10.0)
    assert lib.cpl_skip_last_dirty(r.h) == r.f - 1
    # a negative or NaN tick says nothing about the future - and dirties the history (a later bound must postdate it)
    r2 = SkipRunner(lib)
    r2.frame(dict(spawn=10, **tick)); r2.frame(tick)
    assert not r2.frame({"u": {3: -DT}}, tag=1, bound=10.0) and lib.cpl_skip_last_dirty(r2.h) == 2
    assert not r2.frame({"u": {3: float("nan")}}, tag=2, bound=10.0) and lib.cpl_skip_last_dirty(r2.h) == 3
    assert not r2.frame(tick, tag=2, bound=10.0) and r2.frame(tick, tag=4, bound=10.0)
    # a host write / (un)freeze / new instance marks the history dirty: bounds from before it are void
    lib.cpl_skip_mark_dirty(r2.h)
    assert not r2.frame(tick, tag=5, bound=10.0) and lib.cpl_skip_last_dirty(r2.h) == 6
    assert r2.frame(tick, tag=6, bound=10.0)
    # not eligible: never
    r3 = SkipRunner(lib, eligible=0)
    r3.frame(tick); r3.frame(tick)
    assert not r3.frame(tick, tag=1, bound=10.0)


def test_skip_lists_trusts_a_bound_for_64_frames_and_the_tick_ring_holds_them(lib):
    r = SkipRunner(lib)
    tick = {"u": {3: 1e-4}}
    r.frame(dict(spawn=10, **tick))
    for f in range(1, 200):
        d = r.frame(tick, tag=100 if f > 100 else 0, bound=1.0)
        expect = (f <= 64) if f <= 100 else (f - 100 <= 64)
        assert d == expect, f
    # the sum over the ring is the sum of the ticks between tag and now: just below / just above the bound
    r = SkipRunner(lib)
    r.frame(dict(spawn=10, u={3: 0.25}))
    for _ in range(3):
        r.frame({"u": {3: 0.25}})
    assert r.frame({"u": {3: 0.25}}, tag=0, bound=1.0000021)          # frames 1..4: 1.0, x (1 + 1e-6) < bound
    assert not r.frame({"u": {3: 0.25}}, tag=1, bound=1.0)            # frames 2..5: 1.0 is not < 1.0


# ---- prove_ribbon_order -----------------------------------------------------------------------------------------------------------------------
TICK, AGE0, RID, LIFE = 1, 2, 3, 4


class RibbonRunner:
    def __init__(self, lib, provable=1, front_static=1, age_init_set=1, rid_set=1):
        self.lib = lib
        self.h = lib.cpl_ribbon_new(provable, front_static, age_init_set, rid_set, TICK, AGE0, RID, LIFE)

    def frame(self, *specs, capacity=4096, opt_skip=1, opt_suffix=1, opt_ring=1, sorted_after=True):
        arr, n = rows(*specs)
        out = (C.c_uint32 * 7)()
        self.lib.cpl_ribbon_step(self.h, capacity, arr, n, opt_skip, (opt_suffix & 1) | (2 if opt_ring else 0), out)
        if sorted_after:
            self.lib.cpl_ribbon_sorted(self.h)     # the frame ran its sort (or proved it unnecessary): the list is in key order again
        return dict(zip(("max_spawn", "values_ok", "front", "head_sorted", "rotate", "suffix", "ring"), (int(x) for x in out)))


def trail(spawn=8, tick=DT, age0=0.0, rid=7, life=1.5, **kw):
    return dict(spawn=spawn, u={TICK: tick, AGE0: age0, RID: rid, LIFE: life}, **kw)


def test_ribbon_proofs_hold_for_the_usual_trail(lib):
    r = RibbonRunner(lib)
    d = r.frame(trail())
    assert d["values_ok"] and not d["head_sorted"] and not d["rotate"]      # first frame: nothing sorted yet (history starts dirty)
    assert d["front"]                                                        # (vacuously: no older particle exists; min_tick starts at +inf)
    d = r.frame(trail())
    assert d == dict(max_spawn=8, values_ok=1, front=1, head_sorted=1, rotate=1, suffix=1, ring=1)
    d = r.frame(trail(spawn=0))
    assert d["front"] and d["head_sorted"] and not d["rotate"] and d["suffix"] and d["max_spawn"] == 0     # nothing to rotate without spawns
    d = r.frame(trail(spawn=100000), capacity=4096)
    assert d["max_spawn"] == 4096                                            # a request is capped by the capacity
    d = r.frame(trail(spawn=3), trail(spawn=9), dict(simulated=0, spawn=50, u={TICK: -1.0}))
    assert d["max_spawn"] == 9 and d["rotate"]                               # the largest request of a SIMULATED instance; frozen instances do not count
    d = r.frame(dict(has_parent=1, evcap=256, u={TICK: DT, AGE0: 0.0, RID: 7, LIFE: 1.5}))
    assert d["max_spawn"] == 256                                             # an effect with a parent: the event buffer's capacity


def test_ribbon_premise_violations(lib):
    # the options
    r = RibbonRunner(lib); r.frame(trail())
    d = r.frame(trail(), opt_skip=0)
    assert d["front"] and not d["head_sorted"] and not d["rotate"] and not d["suffix"]
    d = r.frame(trail(), opt_suffix=0)
    assert d["rotate"] and not d["suffix"]
    # a negative tick: values broken for good (negative ages outlive the frame), front broken for good
    r = RibbonRunner(lib); r.frame(trail()); assert r.frame(trail())["rotate"]
    d = r.frame(trail(tick=-DT))
    assert not d["values_ok"] and not d["front"] and not d["head_sorted"]
    d = r.frame(trail())
    assert not d["values_ok"] and not d["front"] and not d["rotate"] and not d["suffix"]
    # a NaN initial age likewise; a negative zero initial age is not +0: no front this frame, values still fine (sign bit set: not >= +0)
    r = RibbonRunner(lib); r.frame(trail())
    assert not r.frame(trail(age0=float("nan")))["values_ok"]
    r = RibbonRunner(lib); r.frame(trail())
    d = r.frame(trail(age0=0.25))
    assert d["values_ok"] and d["head_sorted"] and not d["front"] and not d["rotate"]        # spawns that do not start at +0: a partial sort, no rotation
    assert r.frame(trail())["rotate"]                                                        # ... this frame only (not sticky: the older ages are still >= their ticks)
    # a second RIBBON_ID value: front broken for good; the head stays provably sorted (the radix sort handles the spawns)
    r = RibbonRunner(lib); r.frame(trail()); r.frame(trail())
    d = r.frame(trail(rid=8))
    assert d["values_ok"] and d["head_sorted"] and not d["front"]
    assert not r.frame(trail(rid=7))["front"]
    # two instances with different ids in ONE frame
    r = RibbonRunner(lib); r.frame(trail())
    assert not r.frame(trail(rid=1), trail(rid=2))["front"]
    # instances ticking differently
    r = RibbonRunner(lib); r.frame(trail())
    assert not r.frame(trail(), trail(tick=2 * DT))["front"]
    # a lifetime that changes: rotation still fine, the suffix proof ends for good
    r = RibbonRunner(lib); r.frame(trail()); assert r.frame(trail())["suffix"]
    d = r.frame(trail(life=2.0))
    assert d["rotate"] and not d["suffix"]
    assert not r.frame(trail(life=1.5))["suffix"] and r.frame(trail(life=1.5))["rotate"]
    # a spawn that would die in its first frame (tick >= lifetime) misses the list: the rotation would move the wrong rows
    r = RibbonRunner(lib); r.frame(trail())
    assert not r.frame(trail(life=DT))["front"] and not r.frame(trail(life=float("inf")))["front"]
    # a zero tick: the spawns' keys equal the older particles' lower bound (bound > tick_now fails)
    r = RibbonRunner(lib); r.frame(trail())
    assert not r.frame(trail(tick=0.0))["front"]
    # ... and the smallest tick of ANY earlier frame enters the bound: after a zero-tick frame older particles may still have age == tick
    assert not r.frame(trail())["front"]
    # a tick so small against the current one that fl(min_tick + tick_now) == tick_now
    r = RibbonRunner(lib); r.frame(trail(tick=1e-12))
    assert not r.frame(trail(tick=1.0))["front"]
    # a host write: everything is dirty until the next sort; front broken for good; a write of AGE breaks the values too
    r = RibbonRunner(lib); r.frame(trail()); r.frame(trail())
    lib.cpl_ribbon_host_write(r.h, 0)
    d = r.frame(trail())
    assert d["values_ok"] and not d["head_sorted"] and not d["front"] and not d["rotate"]
    assert r.frame(trail())["head_sorted"] and not r.frame(trail())["front"]
    lib.cpl_ribbon_host_write(r.h, 1)
    assert not r.frame(trail())["values_ok"]
    # static facts missing
    r = RibbonRunner(lib, provable=0, front_static=0); r.frame(trail())
    assert r.frame(trail()) == dict(max_spawn=8, values_ok=0, front=0, head_sorted=0, rotate=0, suffix=0, ring=0)
    r = RibbonRunner(lib, provable=1, front_static=0); r.frame(trail())
    d = r.frame(trail())
    assert d["values_ok"] and d["head_sorted"] and not d["front"]
    # RIBBON_ID never set by the init (0 for every particle) and AGE never set (0): both fine
    r = RibbonRunner(lib, age_init_set=0, rid_set=0); r.frame(trail(age0=5.0, rid=1))
    assert r.frame(trail(age0=-3.0, rid=2))["rotate"]             # (the operands are not read)


# ---- horizon_usable, grids ---------------------------------------------------------------------------------------------------------------------
def test_horizons_need_finite_ticks(lib):
    ok, n = rows({"u": {3: DT}}, {"u": {3: 0.0}}, dict(simulated=0, u={3: float("inf")}))
    assert lib.cpl_horizon_usable(1, 3, ok, n) == 1 and lib.cpl_horizon_usable(0, 3, ok, n) == 0
    for bad in (float("inf"), float("-inf"), float("nan")):
        arr, n = rows({"u": {3: DT}}, {"u": {3: bad}})
        assert lib.cpl_horizon_usable(1, 3, arr, n) == 0
    arr, n = rows({"u": {3: -DT}})
    assert lib.cpl_horizon_usable(1, 3, arr, n) == 1        # negative but finite: the device clamps the clock's advance at zero


def test_init_and_event_grids(lib):
    g = lambda **k: lib.cpl_init_grid(k.get("simulated", 1), k.get("has_parent", 0), k.get("spawn", 0), k.get("evcap", 0), k.get("known", 0), k.get("events", 0),
                                      k.get("capacity", 1 << 24), 256, 4, k.get("big", 0), 256)
    assert g(spawn=0) == 0 and g(spawn=1) == 1 and g(spawn=256) == 1 and g(spawn=257) == 2
    assert g(spawn=1 << 24) == 65536 and g(spawn=1 << 24, big=1) == 16384          # a large burst: four groups of spawns per workgroup
    assert g(spawn=1 << 30, capacity=1000) == 4                                    # never more than the capacity allows
    assert g(spawn=500, simulated=0) == 0
    assert g(has_parent=1, spawn=999, evcap=256) == 1                              # the CPU count of a child effect is unused
    assert g(has_parent=1, evcap=1 << 23) == 256 * 8                               # event-driven: a bounded grid that strides
    assert g(has_parent=1, evcap=1 << 23, known=1, events=0) == 0                  # last frame's count has arrived: exact for zero
    assert g(has_parent=1, evcap=1 << 23, known=1, events=1000) == 4 and g(has_parent=1, evcap=512, known=1, events=100000) == 2
    assert g(has_parent=1, evcap=4096, simulated=0, known=1, events=50) == 0
    e = lib.cpl_event_grid
    assert e(0, 1) == 1 and e(256, 1) == 1 and e(16384, 1) == 1 and e(16385, 1) == 2 and e(1 << 23, 8) == 64 and e(1 << 23, 4096) == 1 and e(1 << 30, 1) == 64 and e(1000, 0) == 1


def test_streaming_hints_are_for_programs_bigger_than_the_infinity_cache(lib):
    h = lib.cpl_stream_hints
    assert h(1 << 24, 60) == 1            # the 16.7M firework: pos / vel / age in and out + lifetime = 60 B per slot, a gigabyte per frame
    assert h(1 << 23, 60) == 1            # force field 8.4M
    assert h(512 * 65536, 28) == 1        # 512 instances x 65,536
    assert h(1 << 22, 8) == 0             # the ribbon effect: age in and out, 4.19M slots: 67 MB per frame - the caches serve it better
    assert h(65536, 60) == 0 and h(0, 60) == 0
    assert h((256 << 20) // 68, 60) == 0 and h((256 << 20) // 68 + 1, 60) == 1     # the boundary: slots x (bytes + 8 B of list) > 256 MiB
    # the update's own plane stores: only when the WRITTEN planes exceed the cache by half (what fits is found there by the next frame)
    st = lib.cpl_store_hints
    assert st(1 << 24, 28) == 1           # firework 16.7M: position + velocity + age = 470 MB written per frame
    assert st(1 << 23, 28) == 0           # force field 8.4M: 235 MB - it fits, and the hint cost C3 22 % (profiles/r04p_ab_walk2.log)
    assert st(1 << 22, 4) == 0 and st(100_000_000, 28) == 1


# ---- plan_merged_launches --------------------------------------------------------------------------------------------------------------------------
def merge(lib, progs, option=1, timed=0):
    arr = (MergeRow * len(progs))()
    for r, p in zip(arr, progs):
        for k in ("independent", "total_chunks", "init_blocks", "init_len", "update_len", "wide_file", "update_streams", "age_cohort"):
            setattr(r, k, p.get(k, {"independent": 1, "total_chunks": 1, "init_blocks": 1, "init_len": 10, "update_len": 10}.get(k, 0)))
    out = (C.c_int32 * (2 * len(progs)))()
    lib.cpl_merge(arr, len(progs), option, timed, out)
    return [(out[2 * i], out[2 * i + 1]) for i in range(len(progs))]


def test_merged_launches(lib):
    stream, cohort, generic, wide = dict(update_streams=1), dict(update_streams=1, age_cohort=1), dict(), dict(wide_file=1)
    # two small independent programs share both launches; families: 0 stream, 1 stream + cohorts, 2 generic, 3 generic wide
    assert merge(lib, [stream, cohort]) == [(0, 0), (0, 1)]
    assert merge(lib, [stream, generic]) == [(0, 0), (0, 2)]                       # k_update_jobs serves all three narrow families in one launch
    assert merge(lib, [stream]) == [(-1, -1)]                                      # alone: its own specialised kernels
    assert merge(lib, [stream, cohort], option=0) == [(-1, -1)] * 2 and merge(lib, [stream, cohort], timed=1) == [(-1, -1)] * 2
    # the wide register file has launches of its own: worth it from two
    assert merge(lib, [stream, generic, wide]) == [(0, 0), (0, 2), (-1, -1)]
    assert merge(lib, [stream, wide, wide]) == [(-1, -1), (1, 3), (1, 3)]
    # not small / not independent / a long pass: stays out, and does not count towards the two
    big = dict(update_streams=1, total_chunks=17)
    assert merge(lib, [stream, big]) == [(-1, -1), (-1, -1)]
    assert merge(lib, [stream, cohort, big]) == [(0, 0), (0, 1), (-1, -1)]
    assert merge(lib, [stream, dict(update_streams=1, independent=0)]) == [(-1, -1)] * 2
    long_init = dict(update_streams=1, init_len=65)
    assert merge(lib, [stream, cohort, long_init]) == [(0, 0), (0, 1), (-1, 0)]      # the long init keeps its own launch, its short update shares
    long_update = dict(update_streams=1, update_len=65)
    assert merge(lib, [stream, cohort, long_update]) == [(0, 0), (0, 1), (0, -1)]
    # no spawn this frame: no init launch at all; too many spawns: its own
    assert merge(lib, [dict(update_streams=1, init_blocks=0), stream, cohort]) == [(-1, 0), (0, 0), (0, 1)]
    assert merge(lib, [dict(update_streams=1, init_blocks=65), stream, cohort]) == [(-1, 0), (0, 0), (0, 1)]
    # one init family member only: its init stays alone even if the updates share
    assert merge(lib, [dict(update_streams=1, init_blocks=0), stream]) == [(-1, 0), (-1, 0)]


# ---- partition_init_passes: which init passes stay in front of the fork behind the heavy program's init ------------------------------------
def _partition(lib, n, heavy, reads):
    """reads: pairs (child, parent) - an instance of program `child` has its parent in program `parent`"""
    m = np.zeros((n, n), dtype=np.uint8)
    for c, p in reads:
        m[c, p] = 1
    side = np.zeros(n, dtype=np.uint8)
    lib.cpl_partition_init_passes(m.ctypes.data, n, heavy, side.ctypes.data)
    return [int(x) for x in side]


def test_the_fork_behind_the_heavy_init(lib):
    # no heavy program: one stream
    assert _partition(lib, 3, -1, [(1, 0), (2, 0)]) == [0, 0, 0]
    # firework.rs: 0 rocket, 1 sparkle trail (child of the rocket), 2 trails (child of the rocket, heavy): the rocket's init precedes the
    # trails' init; the sparkles' init needs neither and goes behind the fork
    assert _partition(lib, 3, 2, [(1, 0), (2, 0)]) == [0, 1, 0]
    # independent programs beside a heavy one: all of them behind the fork
    assert _partition(lib, 4, 1, []) == [1, 0, 1, 1]
    # a program that reads the HEAVY program's particles at init must not run beside its update ...
    assert _partition(lib, 3, 0, [(1, 0)]) == [0, 0, 1]
    # ... and its OTHER parent stays in front with it (parents first, also across the fork), transitively
    assert _partition(lib, 5, 0, [(1, 0), (1, 2), (2, 3)]) == [0, 0, 0, 0, 1]
    # grandparents of the heavy program
    assert _partition(lib, 4, 3, [(3, 2), (2, 1)]) == [1, 0, 0, 0]
    # a chain among light programs only: both behind the fork, in their order
    assert _partition(lib, 3, 0, [(2, 1)]) == [0, 1, 1]


# ---- set_lookup_due: when a context looks its set module up -----------------------------------------------------------------------------------
def test_a_population_is_looked_up_once_after_two_frames(lib):
    h = lib.cpl_set_lookup_new()
    due = lambda pop, enabled=1, job=0, covered=0, n=3: bool(lib.cpl_set_lookup_due(h, enabled, job, covered, n, pop))
    assert not due(11)                       # first merged frame with this population: remembered
    assert due(11)                           # it stood: looked up - once
    assert not due(11) and not due(11)
    assert not due(12) and not due(13)       # an application that creates an effect per frame never generates a module source
    assert not due(13, job=1) and not due(13, covered=1) and not due(13, enabled=0) and not due(13, n=1)
    assert due(13)                           # (none of the refusals above consumed the population)
    assert not due(11) and due(11)           # back to an earlier population: looked up again (a cache entry may exist by now)
    lib.cpl_set_lookup_reset_tried(h)        # a background compilation finished: look again, even at the same population
    assert due(11)
    lib.cpl_set_lookup_free(h)


# ---- split_uncovered: a scene that has outgrown its set module ---------------------------------------------------------------------------------
def _split(lib, progs, loaded=1):
    """progs: (init_family, update_family, has_case, has_own_kernels) per program -> which stay out of the shared launches"""
    n = len(progs)
    dec = np.array([[p[0], p[1]] for p in progs], dtype=np.int32)
    case = np.array([p[2] for p in progs], dtype=np.uint8)
    own = np.array([p[3] for p in progs], dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint8)
    lib.cpl_split_uncovered(dec.ctypes.data, case.ctypes.data, own.ctypes.data, n, loaded, out.ctypes.data)
    return [int(x) for x in out]


def test_a_newcomer_stays_out_so_that_the_covered_programs_keep_their_set_kernels(lib):
    STREAM, COHORT, GENERIC, WIDE = 0, 1, 2, 3
    covered = [(0, STREAM, 1, 1)] * 3 + [(0, COHORT, 1, 1)] * 2 + [(-1, GENERIC, 1, 1)]         # six programs the loaded module knows
    assert _split(lib, covered + [(0, STREAM, 0, 1)]) == [0] * 6 + [1]                            # the 7th joined the scene: out, the six stay
    assert _split(lib, covered + [(0, STREAM, 0, 1)], loaded=0) == [0] * 7                         # no module: everybody shares the interpreters
    assert _split(lib, covered) == [0] * 6                                                         # everybody covered: nothing to do
    assert _split(lib, covered + [(0, STREAM, 0, 0)]) == [0] * 7                                   # a newcomer WITHOUT kernels of its own (HNB_JIT=0, specialisation pending) would be interpreted either way: it stays in
    assert _split(lib, covered + [(0, STREAM, 0, 1), (-1, GENERIC, 0, 1)]) == [0] * 8              # two newcomers against six: not a clear majority (< 4x), the launches stay shared
    assert _split(lib, covered + covered[:2] + [(0, STREAM, 0, 1), (-1, GENERIC, 0, 1)]) == [0] * 8 + [1, 1]   # ... against eight: out
    assert _split(lib, [(0, STREAM, 1, 1), (0, STREAM, 0, 1)]) == [0, 0]                            # one covered program is not a set
    assert _split(lib, covered + [(1, WIDE, 0, 1)]) == [0] * 7                                     # the wide register file never runs on the set kernels: not its concern
    assert _split(lib, covered + [(-1, -1, 0, 1)]) == [0] * 7                                      # a program that is not merged this frame anyway


def test_ring_frames_need_both_ribbon_proofs(lib):
    """A list is kept as a ring (nothing rewritten: hnb_kernels.hip.h "Ring lists") in exactly the frames in which the spawns provably sort in front AND the
    casualties are provably the last rows - or nothing spawns at all."""
    r = RibbonRunner(lib)
    r.frame(trail(spawn=8))                                       # the first frame: nothing sorted yet
    d = r.frame(trail(spawn=8))
    assert d["rotate"] and d["suffix"] and d["ring"]
    d = r.frame(trail(spawn=0))                                   # nothing spawns: the casualties still drop off the end
    assert not d["rotate"] and d["suffix"] and d["ring"]
    assert not r.frame(trail(spawn=8), opt_ring=0)["ring"]        # HNB_OPT_RING_LISTS off
    assert not r.frame(trail(spawn=8), opt_suffix=0)["ring"]      # no suffix proof: the casualties must be looked for, the list is rewritten
    d = r.frame(trail(spawn=8, life=2.5))                         # a lifetime that changed: older particles die out of order from now on
    assert d["rotate"] and not d["suffix"] and not d["ring"]
    r2 = RibbonRunner(lib)
    r2.frame(trail(spawn=8)); r2.frame(trail(spawn=8))
    d = r2.frame(trail(spawn=8, age0=0.5))                        # spawns that do not start at +0: they do not sort in front
    assert not d["rotate"] and not d["ring"]


# ---- plan_slot_init ---------------------------------------------------------------------------------------------------------------------
def _slot_init(lib, specs, capacity=1 << 20, chunks=256, eligible=1, option=1):
    arr, n = rows(*specs)
    out = (C.c_uint32 * 2)()
    lib.cpl_slot_init(eligible, option, capacity, chunks, arr, n, out)
    return int(out[0]), int(out[1])


def test_large_spawns_run_the_init_slot_major(lib):
    """hnb_kernels.hip.h "slot-major init": from an eighth of the program's slots on; k_spawn_mark unless every spawning instance asks for its whole capacity."""
    cap = 1 << 20
    assert _slot_init(lib, [dict(spawn=cap)]) == (1, 0)                     # a burst of `capacity`: every free slot, no marks
    assert _slot_init(lib, [dict(spawn=cap + 5)]) == (1, 0)                 # more than fits: capped, still every free slot
    assert _slot_init(lib, [dict(spawn=cap // 2)]) == (1, 1)                # half: marks (whether it fills up is the device's to say)
    assert _slot_init(lib, [dict(spawn=cap // 8)]) == (1, 1)
    assert _slot_init(lib, [dict(spawn=cap // 8 - 1)]) == (0, 0)            # below an eighth: row-major
    assert _slot_init(lib, [dict(spawn=0)]) == (0, 0)
    assert _slot_init(lib, [dict(spawn=cap)], eligible=0) == (0, 0)         # reads PARTICLE_COUNTER / ribbons / a parent particle
    assert _slot_init(lib, [dict(spawn=cap)], option=0) == (0, 0)           # HNB_OPT_SLOT_INIT off
    assert _slot_init(lib, [dict(spawn=cap, has_parent=1, evcap=cap)]) == (0, 0)
    assert _slot_init(lib, [dict(spawn=cap), dict(spawn=0)] * 4) == (1, 0)  # half of the instances burst: half of the slots
    assert _slot_init(lib, [dict(spawn=cap)] + [dict(spawn=0)] * 15) == (0, 0)   # one in sixteen
    assert _slot_init(lib, [dict(spawn=cap), dict(spawn=cap, simulated=0)]) == (1, 0)   # a frozen instance spawns nothing
    assert _slot_init(lib, [dict(spawn=cap), dict(spawn=100)]) == (1, 1)    # one partial request among them: marks
    assert _slot_init(lib, [dict(spawn=65536)], capacity=65536, chunks=16) == (0, 0)     # a small program: the merged launches' candidate
    assert _slot_init(lib, [dict(spawn=65536)] * 2, capacity=65536, chunks=16) == (1, 0)
    assert _slot_init(lib, [dict(spawn=3)], capacity=257, chunks=1, option=2) == (1, 1)  # HNB_OPT_SLOT_INIT = 2: wherever it is correct
    assert _slot_init(lib, [dict(spawn=0)], capacity=257, chunks=1, option=2) == (0, 0)
    assert _slot_init(lib, [dict(spawn=3)], capacity=257, chunks=1, option=2, eligible=0) == (0, 0)
