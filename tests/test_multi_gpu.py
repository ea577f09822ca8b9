"""The multi-GPU host path that needs no Python and no second process: hnb_comm_* over RCCL and the
single-process, thread-per-context driver examples/multi_gpu.c. One GPU is what the test box has: two contexts share device 0 (RCCL
refuses a communicator with one device twice, so the library sums their counters through the host - the same entry point), each driven
by its own thread; their slabs' union must equal one effect of twice the capacity, bit for bit."""
import json
import os
import subprocess
import threading

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import build as hb
from bevy_hanabi_amd import effects, runtime
from helpers import A, Frame, OracleRunner, frame_seed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_symbols_and_failure_modes_without_a_gpu():
    lib = runtime.load_library()
    for s in ("hnb_comm_create_local", "hnb_comm_unique_id", "hnb_comm_create_rank", "hnb_comm_allreduce_alive", "hnb_comm_destroy", "hnb_comm_set_library"):
        assert hasattr(lib, s)
    import ctypes as C
    h = C.c_void_p()
    assert lib.hnb_comm_create_local(None, 0, C.byref(h)) == -1        # HNB_ERR_INVALID_ARG
    assert lib.hnb_comm_allreduce_alive(None, None, 0, None) == -1
    assert lib.hnb_comm_destroy(None) == 0
    assert lib.hnb_comm_set_library(None, 0x80) == -1                   # unknown flag
    assert os.path.exists(os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so"))   # (built by conftest where rccl.h is)
    assert os.path.exists(hb.build_examples())                          # the C99 driver builds against the public header alone


@pytest.mark.gpu
def test_collective_branch_runs_through_a_stand_in_library():
    """the RCCL branch of hnb_comm_* (CommInitAll, grouped AllReduce on the contexts' streams, read-back) had
    never executed. tests/fake_rccl is a host-memory stand-in with RCCL's signatures (it includes rccl.h) that accepts one device twice,
    selected with hnb_comm_set_library: the branch runs on the one-GPU box, two contexts, two effects each."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "run_fake_comm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    (a0, b0), (a1, b1) = out["local"]
    assert out["totals"] == [a0 + a1, b0 + b1] and 0 < a0 < 30000 and 0 < a1 < 12345
    assert out["totals2"] == [a0, b1]
    assert out["calls"] == [1, 4, 2, 2], out["calls"]   # one CommInitAll, 2 x 2 AllReduce calls, two completed group reductions, two CommDestroy
    assert out["late"] and "already loaded" in out["late"]
    assert out["ctx_destroy_while_held"] == -1


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_processes_reduce_through_hnb_comm_create_rank(tmp_path):
    """hnb_comm_create_rank (ncclGetUniqueId on rank 0 -> ncclCommInitRank on every rank -> grouped ncclAllReduce) had only ever
    run with ONE rank. Two PROCESSES, one rank each, both on device 0 (the stand-in library: the real librccl refuses a device twice), each with its
    capacity slab of one firework effect and a second effect of its own size: every rank sees the sum over both, at three points of the run, and
    a NULL entry counts 0 on the rank that passes it."""
    import sys
    from bevy_hanabi_amd import sharding
    id_file = str(tmp_path / "uid")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "run_fake_rank.py"), str(r), "2", id_file], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=540)
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["totals"] == outs[1]["totals"] and outs[0]["partial"] == outs[1]["partial"]
    assert outs[0]["totals"][0] == [100_000, 4096 * 3]                                         # frame 0: everybody alive, on both ranks together
    assert outs[0]["totals"][2] == [outs[0]["local"][0] + outs[1]["local"][0], outs[0]["local"][1] + outs[1]["local"][1]]
    assert 0 < outs[0]["totals"][2][0] < outs[0]["totals"][1][0] < 100_000                       # the die-off, seen by the collective as it goes
    assert outs[0]["partial"] == [outs[0]["local"][0], outs[0]["local"][1] + outs[1]["local"][1]]
    assert all(o["describe"].startswith("rccl ") and "libfake_rccl" in o["describe"] and o["describe"].endswith("ranks=2 local=1") for o in outs), outs
    # ... and the union of the two slabs is the one effect (the oracle's count)
    orc = OracleRunner(effects.firework_trails(100_000))
    for f in range(60):
        orc.step(Frame(1 / 60, 100_000 if f == 0 else 0, frame_seed(f), time=f / 60))
    assert orc.state()["counters"]["alive_count"] == outs[0]["totals"][2][0]
    assert sharding.slab_plan(100_000, 2) == [(0, 50_000), (50_000, 50_000)]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_the_real_librccl_reduces_a_one_context_communicator():
    """dlopen("librccl.so.1") -> ncclCommInitAll / ncclGetUniqueId + ncclCommInitRank -> grouped ncclAllReduce(ncclUint64,
    ncclSum) on the context's stream -> ncclCommDestroy had run zero times on hardware. HNB_COMM_LIB_SINGLE_RANK makes a communicator of one
    context take that branch: on the one-GPU box every call executes against the real library."""
    import sys
    env = dict(os.environ, NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG_FILE="/dev/stderr")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "real_rccl", "run_real_comm.py")], capture_output=True, text=True, timeout=540, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert out["describe_local"].startswith("rccl ") and "librccl" in out["describe_local"] and out["describe_local"].endswith("ranks=1 local=1"), out
    assert out["describe_rank"].startswith("rccl ") and out["describe_rank"].endswith("ranks=1 local=1"), out
    assert out["totals_local"] == out["alive"] and 0 < out["alive"][0] <= 30000 and 0 < out["alive"][1] <= 12345, out
    assert out["totals_partial"] == [out["alive"][1], 0], out
    assert out["totals_rank"] == out["alive_rank"] and out["alive_rank"][0] <= out["alive"][0], out
    assert out["librccl_mapped"], "librccl is not in the process's maps"


@pytest.mark.gpu
def test_two_contexts_two_threads_on_one_device_equal_one_effect():
    C_ = 30000   # (not a multiple of the 4096-slot chunk)
    frames = 70  # burst, flight, most of the die-off
    asset = effects.firework_trails(C_)
    blob = bh.lower(asset)
    ctxs = [bh.Context(0), bh.Context(0)]
    progs = [c.create_program(blob) for c in ctxs]
    slabs = [p.create_effect(slot_base=g * C_) for g, p in enumerate(progs)]
    errors = []

    def drive(g):
        try:
            for f in range(frames):
                ctxs[g].frame_begin(1 / 60, f / 60)
                slabs[g].set_frame(C_ if f == 0 else 0, frame_seed(f))
                ctxs[g].simulate()
            ctxs[g].synchronize()
        except Exception as e:   # surfaced below
            errors.append((g, e))

    threads = [threading.Thread(target=drive, args=(g,)) for g in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    comm = bh.Comm.local(ctxs)
    total = comm.allreduce_alive([[slabs[0]], [slabs[1]]])
    comm.destroy()
    # one effect of capacity 2C, by the oracle
    big = effects.firework_trails(2 * C_)
    orc = OracleRunner(big)
    for f in range(frames):
        orc.step(Frame(1 / 60, 2 * C_ if f == 0 else 0, frame_seed(f), time=f / 60))
    ref = orc.state()
    assert total == [ref["counters"]["alive_count"]] and 0 < total[0] < 2 * C_
    assert total[0] == slabs[0].alive_count() + slabs[1].alive_count()
    for a in (A.POSITION, A.VELOCITY, A.AGE, A.LIFETIME, A.COLOR):
        union = np.concatenate([fx.read_attr(a.id).view(np.uint32) for fx in slabs])
        np.testing.assert_array_equal(union, ref["attrs"][a.name], err_msg=a.name)
    glob = np.concatenate([fx.alive_list() + g * C_ for g, fx in enumerate(slabs)])
    np.testing.assert_array_equal(np.sort(glob), np.sort(ref["alive"]))
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_c99_thread_per_context_driver(tmp_path):
    exe = hb.build_examples()
    C_ = 20000
    blob_path = tmp_path / "trails.blob"
    blob_path.write_bytes(bh.lower(effects.firework_trails(C_)))
    warmup, steps, windows = 3, 5, 2
    p = subprocess.run([exe, str(blob_path), "0,0", str(warmup), str(steps), str(windows), str(1 / 60), str(tmp_path / "dump")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_ctx"] == 2 and out["devices"] == [0, 0] and out["alive_total"] == 2 * C_ and len(out["window_ms_per_step"]) == windows
    frames = 1 + warmup + steps * windows
    orc = OracleRunner(effects.firework_trails(2 * C_))
    for f in range(frames):
        orc.step(Frame(np.float32(1 / 60), 2 * C_ if f == 0 else 0, frame_seed(f), time=float(np.float32(f) * np.float32(1 / 60))))
    pos = np.concatenate([np.fromfile(str(tmp_path / f"dump.{g}.pos"), dtype=np.uint32).reshape(-1, 3) for g in range(2)])
    np.testing.assert_array_equal(pos, orc.state()["attrs"]["position"])
