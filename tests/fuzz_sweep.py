"""Differential fuzz sweep outside pytest: many seeds in one process, oracle vs the product.

    python tests/fuzz_sweep.py --backend gpu --jit 0 --typed --seeds 5000:5400
    python tests/fuzz_sweep.py --backend cpu_vm --seeds 0:2000

Prints one line per mismatch and a summary; exit code 1 if anything differed. The pytest suites run a fixed
subset of the same seeds (test_fuzz.py)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["gpu", "cpu_vm"], default="gpu")
    ap.add_argument("--jit", choices=["0", "1"], default="1")
    ap.add_argument("--typed", action="store_true")
    ap.add_argument("--abstract", action="store_true", help="i32 literals where floats / u32 are expected (WGSL AbstractInt)")
    ap.add_argument("--seeds", default="0:100")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--capacity", type=int, default=None, help="particles per effect (default: the generators' 300-400); >= 4096 covers completely alive chunks")
    args = ap.parse_args()
    os.environ["HNB_JIT"] = args.jit
    import bevy_hanabi_amd as bh
    from fuzz_assets import random_asset, random_frames, random_typed_asset
    from helpers import CpuVmRunner, GpuRunner, OracleRunner, run_script

    lo, hi = (int(x) for x in args.seeds.split(":"))
    ctx = bh.Context(0) if args.backend == "gpu" else None
    bad = skipped = 0
    t0 = time.time()
    for seed in range(lo, hi):
        gen = random_typed_asset if args.typed else random_asset
        asset = gen(seed, abstract=args.abstract) if args.capacity is None else gen(seed, args.capacity, abstract=args.abstract)
        try:
            bh.lower(asset)
        except bh.ShaderGenerateError as e:
            skipped += 1
            print(f"seed {seed}: not lowered: {e}")
            continue
        except bh.ExprError as e:   # a type error: the oracle has to refuse the asset too
            skipped += 1
            import oracle
            o = OracleRunner(asset)
            try:
                for fr in random_frames(seed, asset.capacity, n=args.frames):
                    o.step(fr)
                bad += 1
                print(f"seed {seed}: MISMATCH the lowering rejects ({e}) what the oracle runs")
            except oracle.OracleError:
                print(f"seed {seed}: rejected by both: {e}")
            continue
        runner = GpuRunner(asset, ctx=ctx) if ctx else CpuVmRunner(asset)
        try:
            run_script(runner, random_frames(seed, asset.capacity, n=args.frames), OracleRunner(asset), every=4)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:300]}")
        finally:
            if ctx:
                runner.prog.destroy()
    print(f"{hi - lo} seeds ({'typed' if args.typed else 'float'}, {args.backend}, jit={args.jit}): {bad} mismatches, {skipped} not lowered, "
          f"{time.time() - t0:.1f} s")
    if ctx:
        ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
