"""Pins the oracle (and the host-side mirror of the CPU pieces of the path) on every golden
vector the reference's own tests hold for this path. All of them are
integer / control-plane vectors: no reference test asserts a particle float, so float parity
stays "unpinned" (see oracle/hanabi_oracle.c header and DESIGN.md).

The vectors are transcribed from the cited reference tests; nothing here reads /root/reference.
"""
import json
import os

import numpy as np
import pytest

import bevy_hanabi_amd as bh
import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- A1: PRNG (vfx_common.wgsl:260-368). Derived KATs (hand-evaluated) ---------------

PCG_KATS = {0: 0x07BB2FE2, 1: 0xA8BEEA3C, 2: 0x7A7ECC88, 42: 0x48F432FF, 0xFFFFFFFF: 0xE62A4902}


def test_pcg_hash_kats():
    for x, want in PCG_KATS.items():
        assert oracle.pcg_hash(x) == want


def test_frand_kats():
    # slot 0, spawner seed 0 -> initial seed pcg_hash(0 ^ 0)
    seed0 = oracle.pcg_hash(0)
    state, (v,) = oracle.frand_kat(seed0, 1)
    assert state == 0x30BE035E and np.float32(v) == np.float32(0.7320672273635864)
    state, v3 = oracle.frand_kat(seed0, 3)
    assert state == 0x8D324821
    assert [np.float32(x) for x in v3] == [np.float32(0.4844777584075928), np.float32(0.7320672273635864), np.float32(0.3928261995315552)]


def test_to_float01_range_and_bit_trick():
    # bitcast((u & 0x007fffff) | 0x3f800000) - 1.0: 23 mantissa bits, [0, 1)
    assert oracle.to_float01(0) == 0.0
    assert oracle.to_float01(0xFF800000) == 0.0
    assert oracle.to_float01(0x007FFFFF) == float(np.float32(1.0) - np.float32(2.0 ** -23))
    assert oracle.to_float01(0x00400000) == 0.5


# ---- A10: EffectSpawner::tick sequences (spawn.rs:1044-1287) ---------------------------------------

def _both_spawners(settings, count, dur, period, cycles, starts_active=True, emit_on_start=True):
    """The product's C++ EffectSpawner and the oracle's C restatement, driven in lock-step."""
    host = bh.EffectSpawner(settings)
    orc = oracle.OracleSpawner(count, dur, period, cycles, starts_active, emit_on_start)
    rng = bh.Pcg32()

    class Both:
        def tick(self, dt):
            a, b = host.tick(dt, rng), orc.tick(dt)
            assert a == b, f"host spawner {a} != oracle spawner {b}"
            return a

        def reset(self):
            host.reset()
            orc.reset()

        def set_active(self, v):
            host.active = v
            orc.active = v

    return Both(), host


INF = float("inf")


def test_spawner_new_3_3_10_2():
    # spawn.rs:1044-1085
    s, host = _both_spawners(bh.SpawnerSettings.new(3.0, 3.0, 10.0, 2), 3.0, 3.0, 10.0, 2)
    assert s.tick(2.0) == 2
    assert host.active and host.cycle_time() == 2.0 and host.cycle_spawn_duration() == 3.0 and host.cycle_period() == 10.0
    assert host.cycle_ratio() == np.float32(0.2) and host.cycle_spawn_count() == 3.0 and host.completed_cycle_count() == 0
    assert s.tick(5.0) == 1
    assert host.cycle_time() == 7.0 and host.cycle_ratio() == np.float32(0.7) and host.completed_cycle_count() == 0
    assert s.tick(8.0) == 3
    assert host.cycle_time() == 5.0 and host.cycle_ratio() == 0.5 and host.completed_cycle_count() == 1
    assert s.tick(10.0) == 0
    assert host.active and host.completed_cycle_count() == 2
    assert s.tick(0.1) == 0
    assert host.active and host.completed_cycle_count() == 2


def test_spawner_once():
    # spawn.rs:1146-1156
    st = bh.SpawnerSettings.once(5.0)
    assert st.is_once()
    s, host = _both_spawners(st, 5.0, 0.0, INF, 1)
    assert host.active
    assert s.tick(0.001) == 5
    assert s.tick(100.0) == 0


def test_spawner_once_reset():
    # spawn.rs:1158-1169
    st = bh.SpawnerSettings.once(5.0)
    assert st.is_once() and st.starts_active()
    s, _ = _both_spawners(st, 5.0, 0.0, INF, 1)
    s.tick(1.0)
    s.reset()
    assert s.tick(1.0) == 5


def test_spawner_once_start_inactive():
    # spawn.rs:1171-1211
    st = bh.SpawnerSettings.once(5.0).with_starts_active(False)
    assert st.is_once() and not st.starts_active()
    s, host = _both_spawners(st, 5.0, 0.0, INF, 1, starts_active=False)
    assert not host.has_completed()
    assert s.tick(1.0) == 0 and not host.has_completed()
    s.set_active(True)
    assert s.tick(1.0) == 5 and host.active and host.has_completed()
    assert s.tick(1.0) == 0 and host.active and host.has_completed()
    s.reset()
    assert host.active and not host.has_completed()
    assert s.tick(1.0) == 5 and host.active and host.has_completed()


def test_spawner_rate():
    # spawn.rs:1213-1225
    st = bh.SpawnerSettings.rate(5.0)
    assert not st.is_once() and st.is_forever()
    s, _ = _both_spawners(st, 5.0, 1.0, 1.0, 0)
    assert s.tick(1.01) == 5
    assert s.tick(0.4) == 2


def test_spawner_rate_active():
    # spawn.rs:1227-1243
    s, host = _both_spawners(bh.SpawnerSettings.rate(5.0), 5.0, 1.0, 1.0, 0)
    s.tick(1.01)
    s.set_active(False)
    assert not host.active
    assert s.tick(0.4) == 0
    s.set_active(True)
    assert host.active
    assert s.tick(0.4) == 2


def test_spawner_rate_accumulate():
    # spawn.rs:1245-1254: 13 ticks of 1/60 s at 5 particles/s -> exactly one particle
    s, _ = _both_spawners(bh.SpawnerSettings.rate(5.0), 5.0, 1.0, 1.0, 0)
    assert sum(s.tick(1.0 / 60.0) for _ in range(13)) == 1


def test_spawner_burst():
    # spawn.rs:1256-1269
    st = bh.SpawnerSettings.burst(5.0, 2.0)
    assert not st.is_once() and st.is_forever()
    s, _ = _both_spawners(st, 5.0, 0.0, 2.0, 0)
    assert s.tick(1.0) == 5
    assert s.tick(4.0) == 10
    assert s.tick(0.1) == 0


def test_spawner_with_active():
    # spawn.rs:1271-1287
    st = bh.SpawnerSettings.rate(5.0).with_starts_active(False)
    s, host = _both_spawners(st, 5.0, 1.0, 1.0, 0, starts_active=False)
    assert not host.active
    assert s.tick(1.0) == 0
    s.set_active(False)
    assert s.tick(1.0) == 0
    s.set_active(True)
    assert host.active
    assert s.tick(1.0) == 5


def test_tick_spawners_once_32_at_16ms():
    # spawn.rs:1350-1489 (tick_spawners under App): once(32) ticked with 16 ms spawns 32, then nothing
    s, _ = _both_spawners(bh.SpawnerSettings.once(32.0), 32.0, 0.0, INF, 1)
    assert s.tick(0.016) == 32
    assert s.tick(0.016) == 0


def test_c1_rate_1000_first_frame():
    # gpu_tests/single_particle.rs:37-45 @ dt = 1/60: 1000/60 = 16.67 -> 16 particles, remainder carried
    s, host = _both_spawners(bh.SpawnerSettings.rate(1000.0), 1000.0, 1.0, 1.0, 0)
    assert s.tick(1.0 / 60.0) == 16
    assert s.tick(1.0 / 60.0) == 17  # 0.67 + 16.67


def test_spawner_settings_period_validation():
    # spawn.rs:1087-1143: try_new errors and new() panics
    U = bh.CpuValue.Uniform
    for period in (U(-1.0, 1.0), U(0.0, 0.0)):
        with pytest.raises(bh.SpawnerSettingsError, match="[Pp]eriod"):
            bh.SpawnerSettings.try_new(3.0, 1.0, period, 0)
        with pytest.raises(bh.PanicError):
            bh.SpawnerSettings.new(3.0, 1.0, period, 0)
    for period in (U(0.0, INF), U(INF, INF)):
        with pytest.raises(bh.SpawnerSettingsError, match="[Ii]nfinite"):
            bh.SpawnerSettings.try_new(3.0, 1.0, period, 0)
    with pytest.raises(bh.PanicError):
        bh.SpawnerSettings.new(3.0, 1.0, U(0.0, INF), 0)


def test_cpu_value_range():
    # spawn.rs:1026-1042
    assert list(bh.CpuValue.Single(1.0).range()) == [1.0, 1.0]
    assert list(bh.CpuValue.Uniform(1.0, 3.0).range()) == [1.0, 3.0]
    assert list(bh.CpuValue.Uniform(3.0, 1.0).range()) == [1.0, 3.0]


# ---- A7 / A8 / location search: integer control-plane vectors ------------------------------------------

def test_k3_prefix_sum_vector():
    # shader_contract_tests.rs:200-341, CPU twin headless_batching_tests.rs:74-107
    ps, totals, dispatch_x = oracle.k3_prefix_sum([10, 5, 8, 6], [(0, 3), (3, 1)])
    assert ps.tolist() == [0, 10, 15, 0]
    assert totals.tolist() == [23, 6]
    assert dispatch_x.tolist() == [1, 1]


def test_find_location_vectors():
    # headless_batching_tests.rs:110-150: packed index -> (effect_index, base_particle, update_index)
    assert oracle.find_location([0, 10, 15], 0, 3, 0) == (0, 0, 0)
    assert oracle.find_location([0, 10, 15], 0, 3, 10) == (1, 10, 0)
    assert oracle.find_location([0, 10, 15], 0, 3, 20) == (2, 15, 5)
    # shader_contract_tests.rs:391-522: indices [100,109,110,114,115,122] with base 100 -> effects [0,0,1,1,2,2]
    got = [oracle.find_location([0, 10, 15], 0, 3, i - 100)[0] for i in (100, 109, 110, 114, 115, 122)]
    assert got == [0, 0, 1, 1, 2, 2]


def test_k2_indirect_vector():
    # shader_contract_tests.rs:1254-1486: caps (200,5), alive (130,1), write idx (0,1)
    meta, prefix, instance_count, render_pong = oracle.k2_indirect([[200, 130, 0, 0, 0], [5, 1, 0, 0, 1]])
    assert prefix.tolist() == [130, 1]
    assert meta[:, 2].tolist() == [130, 1]      # max_update
    assert meta[:, 3].tolist() == [70, 4]       # max_spawn
    assert instance_count.tolist() == [0, 0]
    assert render_pong.tolist() == [1, 0]
    # routing vector shader_contract_tests.rs:526-884: alive (7,5) -> prefix [7,5]
    _, prefix, _, _ = oracle.k2_indirect([[16, 7, 0, 0, 0], [16, 5, 0, 0, 0]])
    assert prefix.tolist() == [7, 5]


def test_k4_nothing_dies_vector():
    # shader_contract_tests.rs:888-1229: POSITION-only asset, 2 instances alive (2,1) -> instance_count (2,1),
    # alive rows stay [0,1] and [0] (stable compaction of an identity list)
    from bevy_hanabi_amd import effects
    from helpers import Frame, OracleRunner
    for alive in (2, 1):
        r = OracleRunner(effects.single_particle(4))
        r.step(Frame(1 / 60, alive, 1))
        r.step(Frame(1 / 60, 0, 2))
        st = r.state()
        assert st["counters"]["instance_count"] == alive and st["counters"]["alive_count"] == alive
        assert st["alive"].tolist() == list(range(alive))


# ---- A11: literal formatting (lib.rs:1924-1950) ------------------------------------------------------

LITERALS = [(1.0, "1."), (-1.0, "-1."), (1.5, "1.5"), (0.5, "0.5"), (0.12345678, "0.123457")]


def _wgsl_text_value(txt):
    return float(np.float32(float(txt + "0" if txt.endswith(".") else txt)))


@pytest.mark.parametrize("x,text", LITERALS)
def test_literal_rounding_matches_to_wgsl_string(x, text):
    want = _wgsl_text_value(text)
    assert oracle.round_literal(x) == want
    assert bh.round_literal_f32(x) == want


def test_literal_rounding_small_and_signed_zero():
    # "{:.6}" prints |x| < 5e-7 as 0.000000 -> "0." / "-0."
    for f in (oracle.round_literal, bh.round_literal_f32):
        assert f(4e-7) == 0.0 and not np.signbit(np.float32(f(4e-7)))
        assert f(-4e-7) == 0.0 and np.signbit(np.float32(f(-4e-7)))
        assert f(6e-7) == float(np.float32(1e-6))
        assert f(123456.789) == float(np.float32(123456.789))  # already exact at 6 decimals in f32


# ---- C1: gpu_tests/single_particle.rs:37-45, value-checked here (the reference only checks it runs) ----

def test_c1_single_particle_oracle():
    from bevy_hanabi_amd import effects
    from helpers import Frame, OracleRunner
    r = OracleRunner(effects.single_particle(16))
    r.step(Frame(1 / 60, 16, 0))
    st = r.state()
    assert st["counters"]["alive_count"] == 16
    assert (st["attrs"]["position"].view(np.float32) == np.array([0.1, 0.2, 0.3], dtype=np.float32)).all()
    assert (st["attrs"]["size3"].view(np.float32) == np.float32(10)).all()
    # capacity = 1 variant of BASELINE.json: the burst is capped by max_spawn
    r = OracleRunner(effects.single_particle(1))
    r.step(Frame(1 / 60, 16, 0))
    assert r.state()["counters"]["alive_count"] == 1


# ---- committed fixture: oracle state hashes of the configs, generated by tests/golden/make_golden.py ----

def test_oracle_matches_committed_fixtures():
    path = os.path.join(HERE, "golden", "oracle_states.json")
    want = json.load(open(path))
    from golden.make_golden import compute_all
    got = compute_all()
    assert got == want
