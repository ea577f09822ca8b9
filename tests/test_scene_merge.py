"""A scene of many small effects in ONE context: hnb_simulate serves the small programs' init and update passes with merged launches
(k_init_jobs / k_update_stream_jobs / k_update_generic_jobs: job tables, interpreter instantiations; DESIGN.md "Kernels"). Every effect
must come out exactly as it does alone - against the oracle, bit for bit, every attribute plane, both lists and every counter - and the
merged launches must really have been used.

Reference: one compute pass per effect batch (src/render/mod.rs:6975-7370); here the passes of independent programs share launches.
"""
import os

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import reference_examples as rx
from helpers import GpuRunner, OracleRunner, assert_same_state
from test_reference_examples import SINGLE, Player

pytestmark = pytest.mark.gpu


class SceneRunner:
    """All entries in one context, one simulate() per frame."""

    def __init__(self, entries, merge):
        self.ctx = bh.Context(0)
        self.ctx.set_option("scene_merge", 1 if merge else 0)
        self.runners = [GpuRunner(e.asset, ctx=self.ctx) for e in entries]

    def step(self, frames):
        """frames[i]: the Frame of entry i, or None (not simulated this frame)"""
        first = next(fr for fr in frames if fr is not None)
        self.ctx.frame_begin(first.dt, first.time)
        for r, fr in zip(self.runners, frames):
            if fr is None:
                r.fx.set_simulated(False)
                continue
            r.fx.set_simulated(True)
            for k, v in fr.props.items():
                r.fx.set_property(k, v)
            r.fx.set_frame(fr.spawn, fr.seed, fr.transform)
        self.ctx.simulate()


def _scene_entries():
    out = []
    for name in SINGLE:
        for index, entry in enumerate(rx.catalog()[name]):
            out.append((name, index, entry))
    return out


def test_merged_launches_leave_every_effect_as_the_oracle_has_it():
    entries = _scene_entries()
    assert len(entries) >= 20
    scene = SceneRunner([e for _, _, e in entries], merge=True)
    oracles = [OracleRunner(e.asset) for _, _, e in entries]
    players = [Player(e, i) for _, i, e in entries]
    n_frames = 150
    for f in range(n_frames):
        frames = [p.frame(f) for p in players]
        if all(fr is None for fr in frames):
            continue
        for orc, fr in zip(oracles, frames):
            if fr is not None:
                orc.step(fr)
        scene.step(frames)
        if f % 10 == 9 or f == n_frames - 1:
            for (name, index, _), orc, run in zip(entries, oracles, scene.runners):
                assert_same_state(orc.state(), run.state(), f"scene {name}[{index}] frame {f}")
    merged = [r for r in scene.runners if "merged launch" in r.prog.kernel_info()]
    assert len(merged) >= len(entries) // 2, "the small programs of the scene did not take the merged launches"
    assert sum(orc.state()["counters"]["alive_count"] for orc in oracles) > 1000


def test_merge_on_and_off_agree_bit_for_bit():
    entries = _scene_entries()
    a = SceneRunner([e for _, _, e in entries], merge=True)
    b = SceneRunner([e for _, _, e in entries], merge=False)
    pa = [Player(e, i) for _, i, e in entries]
    pb = [Player(e, i) for _, i, e in _scene_entries()]    # (a fresh catalog: some drives carry state)
    for f in range(90):
        fa, fb = [p.frame(f) for p in pa], [p.frame(f) for p in pb]
        a.step(fa)
        b.step(fb)
    for (name, index, _), ra, rb in zip(entries, a.runners, b.runners):
        assert_same_state(rb.state(), ra.state(), f"{name}[{index}] merged vs per-program launches")
    assert not any("merged launch" in r.prog.kernel_info() for r in b.runners)
