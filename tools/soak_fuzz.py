import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import bevy_hanabi_amd as bh
import test_fuzz as t
bad = 0
t0 = time.time()
import os
lo = int(os.environ.get('SOAK_LO', '100')); hi = int(os.environ.get('SOAK_HI', '160'))
for seed in range(lo, hi):
    for order in ("spawn", "slot"):
        print('world', seed, order, flush=True)
        try:
            t._world(seed, order)
        except AssertionError as e:
            bad += 1; print("WORLD MISMATCH", seed, order, str(e)[:200])
ctx = bh.Context(0)
for seed in range(400 + lo - 100, 400 + hi - 100):
    print('system', seed, flush=True)
    try:
        t._run_system(seed, ctx)
    except AssertionError as e:
        bad += 1; print("SYSTEM MISMATCH", seed, str(e)[:200])
ctx.close()
print("soak: %d mismatches in %.0f s" % (bad, time.time() - t0))
