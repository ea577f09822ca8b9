"""Set modules (HNB_OPT_SET_MODULE; hnb_jit.h make_set_source, hnb_kernels.hip.h "SET MODULES"): the launches the small programs of a context
share run each program's SPECIALISED code - one hiprtc module per set of programs, a switch over the job's case - instead of the byte-code
interpreters. Same job tables, same launches, same results bit for bit.

Reference: one WGSL module per effect, compiled by the driver (EffectShaderSources::generate, src/lib.rs:805-1336); here the effects that share a
dispatch share a module.

CPU: the module source and its cache entry depend on the SET (not on order or multiplicity); the kernels' LDS / scratch.
GPU: small sets against the oracle in every mode; a program created after the module was built; which code served the launches.
"""
import os
import re
import struct
import subprocess
import time

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from helpers import Frame, GpuRunner, OracleRunner, assert_same_state, frame_seed


def _trio():
    return [effects.firework_trails(4096), effects.instancing(4096), effects.ribbon(4096)]


def _quartet():
    return _trio() + [effects.force_field(4096)]


def warm_sets():
    """The sets the GPU tests below create (tools/warm_jit_cache.py compiles them on the build box: hiprtc needs no GPU)."""
    return [_trio(), _quartet()]


def _entries(cache):
    return sorted(f for f in os.listdir(cache) if f.endswith(".hnbjit"))


def test_a_set_module_is_keyed_by_the_set_not_by_order_or_multiplicity(tmp_path, monkeypatch):
    monkeypatch.setenv("HNB_JIT_CACHE", str(tmp_path))
    a, b = bh.lower(effects.single_particle(16)), bh.lower(effects.instancing(4096))
    bh.jit_precompile_set([a, b])
    first = _entries(tmp_path)
    assert len(first) == 1
    stamp = os.stat(tmp_path / first[0]).st_mtime_ns
    bh.jit_precompile_set([b, a, b, a, a])          # the same set: the same entry, not rewritten
    assert _entries(tmp_path) == first and os.stat(tmp_path / first[0]).st_mtime_ns == stamp
    bh.jit_precompile_set([a])                      # fewer than two members: nothing to share a launch, nothing compiled
    assert _entries(tmp_path) == first
    big = bh.lower(effects.firework_trails(1 << 20))  # can never join a merged launch (more than 65,536 slots): skipped like hnb_simulate skips it
    bh.jit_precompile_set([a, b, big])
    assert _entries(tmp_path) == first

    # the kernels of the module: one LDS object however many programs (the streaming update's 24 KiB of staging), no scratch
    raw = open(tmp_path / first[0], "rb").read()
    hdr = struct.unpack_from("<8sIIQQQQQII", raw, 0)
    assert hdr[0].rstrip(b"\0") == b"HNBJIT2" and hdr[8] == 0        # no name expressions: the kernels are extern "C"
    code = tmp_path / "set.co"
    code.write_bytes(raw[len(raw) - hdr[6]:])
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(code)], capture_output=True, text=True, check=True).stdout
    kernels = {}
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        kernels[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count")}
    assert set(kernels) == {"hnb_set_init", "hnb_set_update"}
    assert kernels["hnb_set_update"]["group_segment_fixed_size"] <= 25 * 1024 and kernels["hnb_set_init"]["group_segment_fixed_size"] <= 1024
    for k in kernels.values():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0


def test_a_bad_blob_is_refused():
    with pytest.raises(bh.HanabiError):
        bh.jit_precompile_set([b"not a program", bh.lower(effects.single_particle(16))])


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------------
def _in_cache(assets):
    """CACHED finds what an earlier hnb_jit_precompile_set left (tools/warm_jit_cache.py on the build box; compiled here if the cache did not travel)."""
    bh.jit_precompile_set([bh.lower(a) for a in assets])


class Scene:
    def __init__(self, assets, set_module):
        self.ctx = bh.Context(0)
        self.ctx.set_option("set_module", set_module)
        self.runs, self.oracles = [], []
        for a in assets:
            self.add(a)
        self.f = 0

    def add(self, asset):
        self.runs.append(GpuRunner(asset, ctx=self.ctx))
        self.oracles.append(OracleRunner(asset))

    def step(self, n, check_every=8):
        for _ in range(n):
            f = self.f
            t = f / 60.0
            self.ctx.frame_begin(1 / 60.0, t)
            for i, (r, o) in enumerate(zip(self.runs, self.oracles)):
                cap = r.asset.capacity
                spawn = cap // 2 if f == 0 else (37 + 11 * i if f % 3 == 0 else 0)     # a burst, then a trickle: every frame of some effect has an init pass
                seed = frame_seed(f, base=0xBEEF00 + 97 * i)
                r.fx.set_frame(spawn, seed, None)
                o.step(Frame(1 / 60.0, spawn, seed, None, t))
            self.ctx.simulate()
            self.f += 1
            if self.f % check_every == 0:
                self.check()

    def check(self):
        for i, (r, o) in enumerate(zip(self.runs, self.oracles)):
            assert_same_state(o.state(), r.state(), f"effect {i} after frame {self.f}")

    def set_frames(self):
        out = []
        for r in self.runs:
            m = re.search(r"set module \(the program's specialised code behind the shared launch\): (\d+) frames", r.prog.kernel_info())
            out.append(int(m.group(1)) if m else 0)
        return out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["off", "cached", "compile"])
def test_gpu_a_small_set_is_the_oracle_in_every_mode(mode):
    if mode == "cached":
        _in_cache(_trio())
    sc = Scene(_trio(), {"off": 0, "cached": 1, "compile": 2}[mode])
    sc.step(120)            # (the oldest particles of the firework die within these frames: lists, sorts)
    sc.check()
    served = sc.set_frames()
    merged = ["merged launch" in r.prog.kernel_info() for r in sc.runs]
    assert all(merged), "the three small programs did not share their launches"
    if mode == "off":
        assert served == [0, 0, 0]
    else:                   # (cached: tools/warm_jit_cache.py compiled this set on the build box; a population is looked up on its second merged frame)
        assert min(served) >= 100, (served, sc.runs[0].prog.kernel_info())
    assert sum(o.state()["counters"]["alive_count"] for o in sc.oracles) > 1000


@pytest.mark.gpu
def test_gpu_a_program_created_later_runs_interpreted_until_its_set_has_a_module():
    _in_cache(_trio())
    _in_cache(_quartet())
    sc = Scene(_trio(), 1)
    sc.step(24)
    before = sc.set_frames()
    assert min(before) >= 20
    sc.add(effects.force_field(4096))            # the set of four: in the cache as well (warm_sets), found once it has stood for two merged frames
    sc.step(24)
    after = sc.set_frames()
    assert after[3] >= 20 and min(a - b for a, b in zip(after[:3], before)) >= 20
    sc.add(effects.single_particle(16))          # a set of five nobody compiled: CACHED never compiles on the frame path ...
    sc.step(24)
    grown = sc.set_frames()
    # ... and (round 5) the four it knows do NOT fall back to the interpreters because a fifth joined: the newcomer runs its own specialised kernels in its
    # own launches (plan::split_uncovered) and the shared launches stay on the set kernels
    assert grown[4] == 0 and min(g - a for g, a in zip(grown[:4], after)) >= 20, (after, grown)
    assert "kept out of the shared launches" in sc.runs[4].prog.kernel_info() and "kept out" not in sc.runs[0].prog.kernel_info()
    assert "no cache entry for this set of 5 programs" in sc.runs[0].prog.kernel_info()
    sc.check()


@pytest.mark.gpu
def test_gpu_a_context_of_one_program_has_no_set():
    sc = Scene(_trio()[:1], 2)
    sc.step(16)
    assert sc.set_frames() == [0] and "merged launch" not in sc.runs[0].prog.kernel_info()


@pytest.mark.gpu
def test_gpu_background_compilation_hands_the_module_over(tmp_path, monkeypatch):
    """HNB_SET_MODULE_BACKGROUND with an empty cache: the frames go on (interpreted, equal to the oracle) while a thread of the library compiles the set;
    the module serves the launches from the frame that finds it ready."""
    monkeypatch.setenv("HNB_JIT_CACHE", str(tmp_path))
    sc = Scene([effects.instancing(4096), effects.single_particle(16)], 3)
    sc.step(4)
    assert "in the background" in sc.runs[0].prog.kernel_info()
    t0, served = time.time(), [0, 0]
    while min(served) == 0 and time.time() - t0 < 180:
        sc.step(8)
        served = sc.set_frames()
        if min(served) == 0:
            time.sleep(0.25)
    assert min(served) > 0, ("no module after 180 s", sc.runs[0].prog.kernel_info())
    before = sc.set_frames()
    sc.step(24)
    sc.check()
    assert min(a - b for a, b in zip(sc.set_frames(), before)) >= 20
    assert "(compiled)" in sc.runs[0].prog.kernel_info()
    assert len(_entries(tmp_path)) == 3      # two programs' own kernels and their set
    sc.ctx.close()


@pytest.mark.gpu
def test_gpu_a_set_that_does_not_compile_is_tried_once_and_says_why(tmp_path, monkeypatch):
    """a failed background build reset the lookup, the next merged frame found nothing in the cache and started the identical
    compilation again - a CPU thread in hiprtc for the life of the context, the compiler's message overwritten by "compiling ... in the background".
    Now the population is tried once, the frames stay on the interpreters (equal to the oracle) and the log keeps the error."""
    monkeypatch.setenv("HNB_JIT_CACHE", str(tmp_path))
    sc = Scene([effects.instancing(4096), effects.single_particle(16)], 3)      # (the programs' own kernels are compiled here, before the option below exists)
    monkeypatch.setenv("HNB_JIT_EXTRA", "-DHNB_CHUNK=)")                         # a compiler option that breaks every translation unit compiled from now on
    sc.step(4)
    assert "in the background" in sc.runs[0].prog.kernel_info()
    t0, info = time.time(), ""
    while "failed" not in info and time.time() - t0 < 120:
        sc.step(4)
        time.sleep(0.25)
        info = sc.runs[0].prog.kernel_info()
    assert "the background compilation failed" in info and "set module builds that failed: 1" in info, info
    sc.step(40)                               # many more merged frames: no second attempt
    time.sleep(1.0)
    sc.step(8)
    info = sc.runs[0].prog.kernel_info()
    assert "set module builds that failed: 1" in info and "in the background" not in info, info
    assert sc.set_frames() == [0, 0]
    sc.check()
    monkeypatch.delenv("HNB_JIT_EXTRA")
    sc.ctx.close()


@pytest.mark.gpu
def test_gpu_a_context_destroyed_while_its_set_compiles(tmp_path, monkeypatch):
    monkeypatch.setenv("HNB_JIT_CACHE", str(tmp_path))
    sc = Scene([effects.ribbon(4096), effects.single_particle(16)], 3)
    sc.step(4)
    assert "in the background" in sc.runs[0].prog.kernel_info()
    sc.check()
    sc.ctx.close()                           # waits for the compilation (hiprtc cannot be interrupted): no crash, and the entry is in the cache afterwards
    assert len(_entries(tmp_path)) == 3


@pytest.mark.gpu
def test_gpu_asynchronous_specialisation_of_single_programs(tmp_path, monkeypatch):
    """HNB_OPT_JIT_ASYNC with an empty cache: hnb_program_create returns at once, the program runs on the ahead-of-time / interpreter kernels (equal to
    the oracle) until the context's compilation thread has its own, and goes on equal to the oracle afterwards; a program destroyed while it waits."""
    monkeypatch.setenv("HNB_JIT_CACHE", str(tmp_path))
    ctx = bh.Context(0)
    ctx.set_option("jit_async", 1)
    ctx.set_option("scene_merge", 0)          # (every program on its own launches: what is being swapped is the program's own kernels)
    t0 = time.time()
    assets = [effects.firework_trails(4096), effects.force_field(4096), effects.instancing(4096)]
    runs = [GpuRunner(a, ctx=ctx) for a in assets]
    oracles = [OracleRunner(a) for a in assets]
    assert all("specialisation pending" in r.prog.kernel_info() for r in runs)
    doomed = GpuRunner(effects.ribbon(4096), ctx=ctx)   # queued behind the three: destroyed before its turn (or during it)
    doomed.fx.destroy()
    doomed.prog.destroy()

    def frames(n, f0):
        for f in range(f0, f0 + n):
            ctx.frame_begin(1 / 60.0, f / 60.0)
            for i, (r, o) in enumerate(zip(runs, oracles)):
                spawn = r.asset.capacity // 2 if f == 0 else (29 + 7 * i if f % 2 == 0 else 0)
                seed = frame_seed(f, base=0xA5A500 + 31 * i)
                r.fx.set_frame(spawn, seed, None)
                o.step(Frame(1 / 60.0, spawn, seed, None, f / 60.0))
            ctx.simulate()
        for i, (r, o) in enumerate(zip(runs, oracles)):
            assert_same_state(o.state(), r.state(), f"effect {i} after frame {f0 + n - 1}")
        return f0 + n

    f = frames(6, 0)
    while any("pending" in r.prog.kernel_info() for r in runs) and time.time() - t0 < 180:
        f = frames(4, f)
        time.sleep(0.2)
    infos = [r.prog.kernel_info().split("\n")[0] for r in runs]
    assert all("pending" not in s and "init=jit" in s for s in infos), infos
    frames(40, f)
    ctx.close()
