"""Development tool (GPU box): a burst of the whole capacity into the dead list a die-off left (slots in death order: scattered spawns), against the
same burst into a fresh effect. Kernel times by HIP events (hnb_ctx_enable_kernel_timing). Usage: reburst_probe.py [capacity] ; HNB_CTX_OPTIONS selects
the variant (spawn_nursery=0: plane-granular stores)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
from bench import DT, frame_seed

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
ctx = bh.Context(0)
prog = ctx.create_program(bh.lower(effects.firework_trails(cap)))
fx = prog.create_effect()


def frame(f, spawn, timed=False):
    if timed:
        ctx.enable_kernel_timing(1)
    ctx.frame_begin(DT, f * DT)
    fx.set_frame(spawn, frame_seed(f))
    ctx.simulate()
    if timed:
        t = ctx.kernel_timing()
        ctx.enable_kernel_timing(0)
        return t
    return None


t_fresh = frame(0, cap, timed=True)
f = 1
while fx.alive_count() > 0:      # the die-off: lifetimes 0.8 .. 1.2 s, dead list in death order
    for _ in range(8):
        frame(f, 0)
        f += 1
t_re = frame(f, cap, timed=True)
f += 1
t_next = frame(f, 0, timed=True)
assert fx.alive_count() == cap
print(f"reburst cap={cap} options={os.environ.get('HNB_CTX_OPTIONS', '')!r}: fresh burst init {t_fresh['init_ms_avg']:.4f} update {t_fresh['update_ms_avg']:.4f} | "
      f"re-burst after {f - 2} frames: init {t_re['init_ms_avg']:.4f} update {t_re['update_ms_avg']:.4f} lists {t_re['compact_ms_avg']:.4f} "
      f"sum {t_re['init_ms_avg'] + t_re['update_ms_avg'] + t_re['compact_ms_avg']:.4f} | next frame update {t_next['update_ms_avg']:.4f}")
print(prog.kernel_info().split("\n")[0])
ctx.close()
