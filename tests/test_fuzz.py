"""Differential fuzzing of the lowering and the program interpreters: seeded random assets (fuzz_assets.py),
oracle vs the product, bit-exact. CPU: the host build of the product interpreters (tests/cpu_vm); GPU: the real
path through the C ABI, with specialised kernels and with the interpreters."""
import pytest

import bevy_hanabi_amd as bh
from fuzz_assets import random_asset, random_frames
from helpers import CpuVmRunner, GpuRunner, OracleRunner, run_script

CPU_SEEDS = list(range(60))
GPU_SEEDS = list(range(100, 124))


def _run(seed, make_runner):
    asset = random_asset(seed)
    try:
        blob = bh.lower(asset)
    except (bh.ExprError, bh.ShaderGenerateError) as e:
        # register pressure of a deep random tree: a lowering error, never a wrong result; the oracle agrees it is an asset
        pytest.skip(f"seed {seed}: {e}")
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
