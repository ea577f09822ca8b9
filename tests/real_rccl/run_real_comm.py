"""The REAL librccl behind hnb_comm_* on a one-GPU box: with HNB_COMM_LIB_SINGLE_RANK a communicator of one context is
built by ncclCommInitAll over one device (hnb_comm_create_local) / ncclCommInitRank with n_ranks = 1 (hnb_comm_create_rank), and
hnb_comm_allreduce_alive runs a grouped ncclAllReduce(ncclUint64, ncclSum) on the context's simulation stream behind its frames: every symbol,
enum value and the stream ordering execute for real. Run as its own process (the library choice precedes the first hnb_comm_* call); prints one
JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bevy_hanabi_amd as bh  # noqa: E402
from bevy_hanabi_amd import effects, runtime  # noqa: E402
from helpers import frame_seed  # noqa: E402


def main():
    runtime.comm_set_library(None, single_rank=True)   # librccl by name, as a production host gets it
    ctx = bh.Context(0)
    caps = (30000, 12345)
    fxs = [ctx.create_program(bh.lower(effects.firework_trails(c))).create_effect() for c in caps]
    out = {}
    comm = bh.Comm.local([ctx])
    out["describe_local"] = comm.describe()
    for f in range(60):   # burst, flight, the first deaths: the all-reduce is enqueued behind these frames without a synchronisation in between
        ctx.frame_begin(1 / 60, f / 60)
        for fx, c in zip(fxs, caps):
            fx.set_frame(c if f == 0 else 0, frame_seed(f))
        ctx.simulate()
    out["totals_local"] = comm.allreduce_alive([fxs])
    out["alive"] = [fx.alive_count() for fx in fxs]
    out["totals_partial"] = comm.allreduce_alive([[fxs[1], None]])
    comm.destroy()
    comm2 = bh.Comm.rank(ctx, bh.Comm.unique_id(), 0, 1)
    out["describe_rank"] = comm2.describe()
    for f in range(60, 66):
        ctx.frame_begin(1 / 60, f / 60)
        for fx in fxs:
            fx.set_frame(0, frame_seed(f))
        ctx.simulate()
    out["totals_rank"] = comm2.allreduce_alive([fxs])
    out["alive_rank"] = [fx.alive_count() for fx in fxs]
    comm2.destroy()
    maps = open("/proc/self/maps").read()
    out["librccl_mapped"] = sorted({ln.split()[-1] for ln in maps.splitlines() if "librccl" in ln})
    ctx.close()
    import ctypes
    ctypes.CDLL(None).fflush(None)               # (librccl prints its banner and warnings through C stdio into stdout: out, before the result line)
    print("\nRESULT " + json.dumps(out), flush=True)
    os._exit(0)                                  # (nothing the library prints at exit may follow it)


if __name__ == "__main__":
    main()
