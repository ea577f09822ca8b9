"""hnb_effect_check / hnb_effect_compare (include/hanabi_amd.h "Verification on the device") and what bench.py builds on them: the gate on
the state the TIMED frames of a churn configuration left at full size. A healthy effect passes; a deliberately broken
host-side proof (HNB_OPT_TEST_BREAK_PROOF: "nothing can die" claimed for every frame that spawns nothing) is noticed by every leg."""
import argparse
import os
import sys

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from helpers import Frame, GpuRunner, OracleRunner, assert_same_state, frame_seed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

PLAIN = {"horizon": 0, "age_cohort": 0, "cull_lifetime": 0, "skip_lists": 0, "stream_hints": 0, "overlap_updates": 0,
         "suffix_proof": 0, "ring_lists": 0, "alternate": 0, "transpose": 0, "scene_merge": 0}


@pytest.fixture(autouse=True)
def _test_hooks(monkeypatch):
    monkeypatch.setenv("HNB_ENABLE_TEST_HOOKS", "1")   # HNB_OPT_TEST_BREAK_PROOF is refused without it


def test_the_test_hook_is_refused_outside_a_test_process(monkeypatch):
    monkeypatch.delenv("HNB_ENABLE_TEST_HOOKS")
    c = bh.Context(0)
    with pytest.raises(bh.HanabiError):
        c.set_option("test_break_proof", 1)
    c.set_option("test_break_proof", 0)    # switching it OFF is always allowed
    c.close()


def _ctx(options):
    c = bh.Context(0)
    for k, v in options.items():
        c.set_option(k, v)
    return c


def _frames(cap):
    fr = [Frame(1 / 60, cap, frame_seed(0))] + [Frame(1 / 60, 0, frame_seed(f), time=f / 60) for f in range(1, 56)]          # burst, flight, the die-off begins
    fr += [Frame(1 / 60, 900 if f % 2 else 0, frame_seed(f), time=f / 60) for f in range(56, 80)]                           # spawns into freed slots while it goes on
    return fr


def test_a_healthy_effect_passes_and_equals_its_plain_replay_and_the_oracle():
    cap = 6 * 4096 + 77
    asset = effects.firework_trails(cap)
    fast, plain = _ctx({}), _ctx(PLAIN)
    g, p, o = GpuRunner(asset, ctx=fast), GpuRunner(asset, ctx=plain), OracleRunner(asset)
    for i, fr in enumerate(_frames(cap)):
        for x in (g, p, o):
            x.step(fr)
        if i in (0, 30, 52, 60, 79):
            c = g.fx.check()
            assert c["ok"] == 1 and c["capacity"] == cap and c["alive_count"] == g.fx.alive_count(), c
            assert p.fx.check()["ok"] == 1
            d = g.fx.compare(p.fx)
            assert d["equal"] == 1 and d["first_section"] == -1, d
    assert 0 < g.fx.alive_count() < cap
    assert_same_state(o.state(), g.state(), "fast path")
    assert_same_state(o.state(), p.state(), "plain path")
    # compare notices a single flipped bit in a plane, a swapped pair of list rows is found by check (the alive bytes no longer agree ... no: by compare)
    pos = p.fx.read_attr(bh.Attribute.POSITION.id).view(np.uint32).copy()
    slot = int(p.fx.alive_list()[5])
    pos[slot, 1] ^= 1
    p.fx.write_attr(bh.Attribute.POSITION.id, pos)
    d = g.fx.compare(p.fx)
    assert d["equal"] == 0 and d["attr_diffs"] == 1 and d["first_section"] == 2 + bh.Attribute.POSITION.id and d["first_index"] == slot * 3 + 1, d
    with pytest.raises(bh.HanabiError):
        other = GpuRunner(effects.firework_trails(cap + 1), ctx=plain)
        g.fx.compare(other.fx)                      # different layouts (capacity)
    fast.close()
    plain.close()


def test_a_broken_proof_is_noticed_by_check_and_by_compare():
    cap = 6 * 4096 + 77
    asset = effects.firework_trails(cap)
    broken, plain = _ctx({"test_break_proof": 1}), _ctx(PLAIN)
    g, p = GpuRunner(asset, ctx=broken), GpuRunner(asset, ctx=plain)
    for fr in _frames(cap)[:56]:     # through the first deaths, none of whose frames spawn: the hook skips their list kernels
        g.step(fr)
        p.step(fr)
    assert p.fx.check()["ok"] == 1 and p.fx.alive_count() < cap
    c = g.fx.check()
    assert c["ok"] == 0 and c["fault"] == 1 and c["alive_byte_mismatches"] > 0, c     # particles died in frames whose lists were skipped: the lists still name them
    d = g.fx.compare(p.fx)
    assert d["equal"] == 0 and d["counter_diffs"] > 0, d
    broken.close()
    plain.close()


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _args(capacity):
    return argparse.Namespace(capacity=capacity, instances=None, backend="nccl", force_device=None, comm=False)


@pytest.mark.parametrize("name", ["c2_mixed", "c2_dieoff", "c5", "c2_events"])
def test_the_bench_gate_on_the_timed_state_passes_on_every_churn_configuration(name):
    bench = _bench()
    args = _args(1 << 17)
    D = bench.Dist(args)
    w = bench.Workload(name, args, D)
    for _ in range(150 if name != "c2_dieoff" else 2 * bench.DIEOFF_END + 60):
        w.step()
    r = bench.parity_timed_state(w, args, D)
    assert r["ok"] and all(c["ok"] for c in r["checks"]) and all(d["equal"] for d in r["diffs"]) and len(r["diffs"]) == len(w.fxs), r
    w.close()


def test_the_bench_gate_refuses_a_state_a_broken_proof_produced():
    bench = _bench()
    args = _args(1 << 17)
    D = bench.Dist(args)
    w = bench.Workload("c2_dieoff", args, D, options={"test_break_proof": 1})
    for _ in range(bench.DIEOFF_FIRST + 12):       # into the die-off: frames without spawns in which particles die
        w.step()
    r = bench.parity_timed_state(w, args, D)
    assert not r["ok"] and r["problems"], r
    assert any("invariants" in p for p in r["problems"]) and any("plain replay" in p for p in r["problems"]), r["problems"]
    w.close()
    # ... and the line built from such a record carries no number (tests/test_bench_line.py builds the rest of that case without a GPU)
