"""The N > 1 path on CPU: two gloo processes, each simulating its capacity slab (through the
oracle, since there is no GPU here), the alive counters all-reduced exactly as bench.py does on
RCCL. Checks that the union of the slabs is bit-identical to one single-device run."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects, sharding


def test_slab_plan_and_spawn_split():
    assert sharding.slab_plan(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sharding.slab_plan(1 << 24, 8)[3] == (3 << 21, 1 << 21)
    assert sharding.split_spawn(10, [3, 3, 2, 2]) == [3, 3, 2, 2]
    assert sharding.split_spawn(5, [3, 3, 2, 2]) == [3, 2, 0, 0]
    assert sharding.split_spawn(100, [3, 0, 2]) == [3, 0, 2]
    assert sharding.instance_plan(10, 4) == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    with pytest.raises(ValueError):
        sharding.slab_plan(8, 0)
    assert sharding.allreduce_alive([5, 7]) == [5, 7]  # no process group: identity


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


TOTAL = 6000
FRAMES = 70


def _script(total_free_by_rank=None):
    from helpers import frame_seed
    # burst of the whole capacity, die-off, a partial re-spawn that only fits the first slab(s)
    out = [(TOTAL, frame_seed(0))] + [(0, frame_seed(f)) for f in range(1, 60)] + [(1000, frame_seed(60))]
    out += [(0, frame_seed(f)) for f in range(61, FRAMES)]
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import Frame, OracleRunner
    plan = sharding.slab_plan(TOTAL, world)
    slot_base, cap = plan[rank]
    r = OracleRunner(effects.firework_trails(cap), slot_base=slot_base)
    import torch
    for f, (spawn, seed) in enumerate(_script()):
        # every rank derives the same split from the all-gathered free capacities
        free = torch.zeros(world, dtype=torch.int64)
        free[rank] = cap - r.fx.alive_count()
        dist.all_reduce(free)
        mine = sharding.split_spawn(spawn, free.tolist())[rank]
        r.step(Frame(1 / 60, mine, seed, time=f / 60))
    total_alive = sharding.allreduce_alive([r.fx.alive_count()])[0]
    st = r.state()
    q.put((rank, slot_base, cap, total_alive, st["alive"], {k: v for k, v in st["attrs"].items()}, st["counters"]["alive_count"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slabs_equal_single_device_run():
    from helpers import Frame, OracleRunner
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-device reference over the burst + die-off part (the re-spawn reuses slots in
    # last-killed-first order per slab, which a single list would order differently: §8e only
    # promises union == single run for bursts that fill capacity, so compare up to frame 59)
    single = OracleRunner(effects.firework_trails(TOTAL))
    shards = [OracleRunner(effects.firework_trails(c), slot_base=b) for b, c in sharding.slab_plan(TOTAL, world)]
    for f, (spawn, seed) in enumerate(_script()[:60]):
        single.step(Frame(1 / 60, spawn, seed, time=f / 60))
        split = sharding.split_spawn(spawn, [s.asset.capacity - s.fx.alive_count() for s in shards])
        for s, n in zip(shards, split):
            s.step(Frame(1 / 60, n, seed, time=f / 60))
    ref = single.state()
    union_alive = np.concatenate([s.state()["alive"] + b for s, (b, _) in zip(shards, sharding.slab_plan(TOTAL, world))])
    np.testing.assert_array_equal(ref["alive"], union_alive)
    for name, plane in ref["attrs"].items():
        union = np.concatenate([s.state()["attrs"][name] for s in shards])
        np.testing.assert_array_equal(plane, union, err_msg=name)
    assert 0 < ref["counters"]["alive_count"] < TOTAL  # the die-off is in progress: compaction was exercised

    # the two gloo ranks agree on the all-reduced counter, and it is the sum of their slabs
    totals = {t[3] for t in results}
    assert len(totals) == 1
    assert totals.pop() == sum(t[6] for t in results)
    assert [t[1] for t in results] == [0, TOTAL // 2]


# ---- BASELINE config 4: sharding BY INSTANCE (bench.py --config c4) ---------------------------------------------------
N_INST, INST_CAP, INST_FRAMES = 12, 300, 6


def _instance_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from helpers import Frame, OracleRunner
    mine = sharding.instance_plan(N_INST, world)[rank]          # instance i lives on rank i mod world, as in bench.py
    runners = {g: OracleRunner(effects.instancing(INST_CAP)) for g in mine}
    for f in range(INST_FRAMES):
        for g, r in runners.items():   # per-instance seed and transform depend on the GLOBAL instance index only
            r.step(Frame(1 / 60, INST_CAP if f == 0 else 0, bench.instance_seed(f, g), np.array(bench.instance_transform(g), np.float32), time=f / 60))
    alive_local = [runners[g].fx.alive_count() if g in runners else 0 for g in range(N_INST)]
    alive_total = sharding.allreduce_alive(alive_local)          # the only collective: per-effect alive counters
    q.put((rank, mine, alive_total, {g: r.state()["attrs"]["position"] for g, r in runners.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_instance_sharding_equals_single_process():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from helpers import Frame, OracleRunner
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_instance_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == list(range(0, N_INST, 2)) and results[1][1] == list(range(1, N_INST, 2))
    assert results[0][2] == results[1][2] == [INST_CAP] * N_INST   # every rank sees every instance's counter after the all-reduce
    got = {}
    for _, _, _, planes in results:
        got.update(planes)
    for g in range(N_INST):   # one process simulating all instances gives the same particles
        r = OracleRunner(effects.instancing(INST_CAP))
        for f in range(INST_FRAMES):
            r.step(Frame(1 / 60, INST_CAP if f == 0 else 0, bench.instance_seed(f, g), np.array(bench.instance_transform(g), np.float32), time=f / 60))
        np.testing.assert_array_equal(r.state()["attrs"]["position"], got[g], err_msg=f"instance {g}")


def test_bench_helpers_agree_with_the_oracle_and_the_profiles():
    """bench.py restates the PCG hash for its seed lists; its roofline fraction is built from MOVED bytes (a counter pass of the same run,
    or the committed record if - and only if - it was taken from this very kernel build), never from the algorithmic 68 B."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import oracle
    for f in (0, 1, 7, 1000, 123456):
        assert bench.frame_seed(f) == oracle.pcg_hash(0xC0FFEE + f)
        assert bench.instance_seed(f, 0) == bench.frame_seed(f) and bench.instance_seed(f, 5) != bench.frame_seed(f)
    assert {"c2", "c2_lean", "c2_mixed", "c2_dieoff", "c2_reburst", "c2_interop", "c2_view", "c2_events", "c3", "c4", "c5"} == set(bench.CONFIGS)
    assert bench.CONFIGS["c2"]["capacity"] == 16_777_216 and bench.CONFIGS["c2"]["bytes_per_update"] == 68
    assert bench.CONFIGS["c3"]["capacity"] == 8_388_608 and bench.CONFIGS["c5"]["capacity"] == 4_194_304 and bench.CONFIGS["c4"]["instances"] * 8 == 4096
    assert bench.frame_dt(36) == bench.DT and bench.frame_dt(10_000) * 10_000 < bench.MIN_LIFETIME
    # a recorded traffic table is only accepted for the kernel sources it was measured on
    stamp = bench.kernel_source_stamp()
    assert len(stamp) == 16 and bench.load_recorded_traffic("0" * 16)[0] is None
    table = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    rec, src = bench.load_recorded_traffic(table["kernel_source_stamp"])
    assert rec == table["configs"] and table["kernel_source_stamp"] in src
    # the fraction: moved bytes / kernel time / 8 TB/s; the 68 B figure only ever appears under "algorithmic"
    n = 1 << 24
    res = {"ms_per_step": 0.14, "roofline": {"kernel_ms_avg": 0.13, "updates_per_launch": float(n), "algorithmic": {}}}
    bench.attach_roofline(res, "c2", {"bytes_per_launch": 48.0 * n, "fetch_size_kib": 1.0, "write_size_kib": 1.0, "frame_bytes": 49.0 * n}, "unit test")
    r = res["roofline"]
    assert abs(r["achieved"] - 48.0 * n / 0.13e-3 / 1e9) < 1e-6 and r["frac"] == r["achieved"] / 8000.0 and r["frac"] < 1.0
    assert abs(r["algorithmic"]["ratio_moved_to_algorithmic"] - 48 / 68) < 1e-12 and r["whole_step"]["frac"] < 1.0
    res = {"ms_per_step": 0.14, "roofline": {"kernel_ms_avg": 0.13, "updates_per_launch": float(n), "algorithmic": {}}}
    bench.attach_roofline(res, "c2", None, "model")
    assert res["roofline"]["traffic"] is None and res["roofline"]["moved_bytes_per_update"] == bench.CONFIGS["c2"]["model_bytes"] and res["roofline"]["frac"] < 1.0
