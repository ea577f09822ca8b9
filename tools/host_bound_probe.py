"""Development tool (GPU box): is a configuration's frame bounded by the HOST's submission or by the GPU?

For each (capacity, options) it plays the bench's own frames (bench.Workload) and reports, per frame:
  wall          loop time with one synchronisation at the end (what bench.py's ms_per_step is)
  python        time in the binding outside hnb_simulate (frame_begin, set_frame, spawner arithmetic)
  in_simulate   time inside the hnb_simulate call
A frame whose wall does not move when the capacity shrinks 64-fold, and whose python + in_simulate adds up to the wall, waits for the host.

Usage: python tools/host_bound_probe.py [config=c5] [frames=2000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
full = bench.CONFIGS[name]["capacity"]


class OneRank:
    on, world, rank, device_index = False, 1, 0, 0


def play(capacity, options, label):
    args = argparse.Namespace(capacity=capacity, instances=None, steps=20, windows=1, warmup=5)
    w = bench.Workload(name, args, OneRank(), options=options)
    w.dt = bench.DT
    sim = w.ctx.simulate
    acc = [0]

    def timed_simulate():
        t = time.perf_counter_ns(); sim(); acc[0] += time.perf_counter_ns() - t

    for _ in range(bench.warmup_frames(name, 5)):
        w.step()
    w.ctx.synchronize()
    t0 = time.perf_counter_ns()
    for _ in range(frames):
        w.step()
    t_submit = time.perf_counter_ns() - t0
    w.ctx.synchronize()
    wall = (time.perf_counter_ns() - t0) / frames / 1e3
    w.ctx.simulate = timed_simulate
    t0 = time.perf_counter_ns()
    for _ in range(frames):
        w.step()
    t_loop = time.perf_counter_ns() - t0
    w.ctx.synchronize()
    wall2 = (time.perf_counter_ns() - t0) / frames / 1e3
    in_sim = acc[0] / frames / 1e3
    print(f"{label:34s} wall {wall:6.2f} us/frame (submit loop alone {t_submit / frames / 1e3:6.2f}); instrumented: wall {wall2:6.2f}, python {t_loop / frames / 1e3 - in_sim:5.2f}, "
          f"in hnb_simulate {in_sim:6.2f}; alive {w.alive()}", flush=True)
    w.close()


for cap in (full, full // 64):
    play(cap, None, f"{name} capacity {cap} (defaults)")
    if name == "c5":
        play(cap, {"ring_lists": 0}, f"{name} capacity {cap} ring_lists=0")
