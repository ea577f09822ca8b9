"""One RANK of a two-process communicator (tests/test_multi_gpu.py): hnb_comm_unique_id on rank 0 -> a file -> hnb_comm_create_rank on every
rank -> hnb_comm_allreduce_alive, through tests/fake_rccl/libfake_rccl.so (the real librccl refuses two ranks on one device; the stand-in
meets in shared memory). Every rank simulates its capacity slab of one firework effect on device 0.
    python run_fake_rank.py <rank> <n_ranks> <id file> [library]         prints one JSON line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects, runtime, sharding  # noqa: E402
from helpers import frame_seed  # noqa: E402

rank, n, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lib = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
runtime.comm_set_library(lib, duplicate_devices=True)
if rank == 0:
    uid = bh.Comm.unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        assert time.time() - t0 < 120, "rank 0 never published the unique id"
        time.sleep(0.01)
    uid = open(id_file, "rb").read()
assert len(uid) == 128
total_cap = 100_000
base, cap = sharding.slab_plan(total_cap, n)[rank]
ctx = bh.Context(0)
fx = ctx.create_program(bh.lower(effects.firework_trails(cap))).create_effect(slot_base=base)
other = ctx.create_program(bh.lower(effects.force_field(4096 * (rank + 1)))).create_effect()
comm = bh.Comm.rank(ctx, uid, rank, n)
totals = []
for f in range(60):
    ctx.frame_begin(1 / 60, f / 60)
    fx.set_frame(cap if f == 0 else 0, frame_seed(f))
    other.set_frame(4096 * (rank + 1) if f == 0 else 0, frame_seed(500 + f))
    ctx.simulate()
    if f in (0, 55, 59):
        totals.append(comm.allreduce_alive([[fx, other]]))     # (not synchronised: ordered behind the frames on the context's stream)
partial = comm.allreduce_alive([[fx if rank == 0 else None, other]])
desc = comm.describe()
local = [fx.alive_count(), other.alive_count()]
comm.destroy()
ctx.close()
print(json.dumps({"rank": rank, "totals": totals, "partial": partial, "local": local, "describe": desc}))
