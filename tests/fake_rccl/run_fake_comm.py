"""Runs in its OWN process (tests/test_multi_gpu.py): hnb_comm_set_library must precede the first hnb_comm_* use of a process.
Two contexts on device 0, the collective branch of hnb_comm_* (ncclCommInitAll -> grouped ncclAllReduce on the contexts' streams ->
read-back) through tests/fake_rccl/libfake_rccl.so. Prints one JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (first: see bench.py on which libamdhip64 owns the devices)

import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects, runtime  # noqa: E402
from helpers import frame_seed  # noqa: E402

fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
runtime.comm_set_library(fake, duplicate_devices=True)
caps = (30000, 12345)
ctxs = [bh.Context(0), bh.Context(0)]
fxs = []
for c, cap in zip(ctxs, caps):
    a = c.create_program(bh.lower(effects.firework_trails(cap))).create_effect()
    b = c.create_program(bh.lower(effects.force_field(cap // 2))).create_effect()
    fxs.append([a, b])
    for f in range(58):
        c.frame_begin(1 / 60, f / 60)
        a.set_frame(cap if f == 0 else 0, frame_seed(f))
        b.set_frame(cap // 2 if f == 0 else 0, frame_seed(100 + f))
        c.simulate()                              # (not synchronised: the all-reduce is ordered behind the frames on each context's stream)
comm = bh.Comm.local(ctxs)
totals = comm.allreduce_alive(fxs)
totals2 = comm.allreduce_alive([[fxs[0][0], None], [None, fxs[1][1]]])   # NULL entries count 0
local = [[fx.alive_count() for fx in row] for row in fxs]
late = None
try:
    runtime.comm_set_library(None)                # too late now
except bh.HanabiError as e:
    late = str(e)
rc_ctx = runtime.load_library().hnb_ctx_destroy(ctxs[0]._h)   # a communicator holds its contexts
comm.destroy()
calls = (C.c_int * 4)()
C.CDLL(fake).fake_rccl_calls(calls)
for c in ctxs:
    c.close()
print(json.dumps({"totals": totals, "totals2": totals2, "local": local, "calls": list(calls), "late": late, "ctx_destroy_while_held": rc_ctx}))
