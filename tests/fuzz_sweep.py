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
    ap.add_argument("--scene", type=int, default=0, help="gpu: groups of this many consecutive seeds share ONE context and one simulate() per frame "
                                                         "(the merged launches of small programs: k_init_jobs / k_update_jobs)")
    args = ap.parse_args()
    os.environ["HNB_JIT"] = args.jit
    import bevy_hanabi_amd as bh
    from fuzz_assets import random_asset, random_frames, random_typed_asset
    from helpers import CpuVmRunner, GpuRunner, OracleRunner, run_script

    lo, hi = (int(x) for x in args.seeds.split(":"))
    if args.scene:
        return scene_sweep(args, lo, hi)
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


def scene_sweep(args, lo, hi):
    """Random assets, `--scene` of them per context, every frame one simulate(): each effect against its own oracle."""
    import bevy_hanabi_amd as bh
    from fuzz_assets import random_asset, random_frames, random_typed_asset
    from helpers import Frame, GpuRunner, OracleRunner, assert_same_state
    bad = skipped = merged = total = in_set = 0
    t0 = time.time()
    for g in range(lo, hi, args.scene):
        seeds, assets = [], []
        for seed in range(g, min(g + args.scene, hi)):
            gen = random_typed_asset if (args.typed or seed % 3 == 0) else random_asset
            asset = gen(seed, abstract=args.abstract) if args.capacity is None else gen(seed, args.capacity, abstract=args.abstract)
            try:
                bh.lower(asset)
            except (bh.ShaderGenerateError, bh.ExprError):
                skipped += 1
                continue
            seeds.append(seed); assets.append(asset)
        if len(assets) < 2:
            continue
        ctx = bh.Context(0)
        runners = [GpuRunner(a, ctx=ctx) for a in assets]
        oracles = [OracleRunner(a) for a in assets]
        scripts = [random_frames(seed, a.capacity, n=args.frames) for seed, a in zip(seeds, assets)]
        try:
            for f in range(args.frames):
                base = scripts[0][f]
                ctx.frame_begin(base.dt, base.time)
                for r, o, sc in zip(runners, oracles, scripts):
                    fr = Frame(base.dt, sc[f].spawn, sc[f].seed, sc[f].transform, time=base.time, props=sc[f].props)
                    for k, v in fr.props.items():
                        r.fx.set_property(k, v)
                    r.fx.set_frame(fr.spawn, fr.seed, fr.transform)
                    o.step(fr)
                ctx.simulate()
                if f % 4 == 3 or f == args.frames - 1:
                    for seed, r, o in zip(seeds, runners, oracles):
                        assert_same_state(o.state(), r.state(), f"seed {seed} (scene of {len(assets)}) frame {f}")
        except AssertionError as e:
            bad += 1
            print(f"seeds {seeds[0]}..{seeds[-1]}: MISMATCH {str(e)[:300]}")
        total += len(runners)
        merged += sum(1 for r in runners if "merged launch" in r.prog.kernel_info())
        in_set += sum(1 for r in runners if "set module (the program" in r.prog.kernel_info())   # (HNB_CTX_OPTIONS=set_module=2: compiled per scene)
        for r in runners:
            r.fx.destroy()
            r.prog.destroy()
        ctx.close()
    print(f"{hi - lo} seeds in scenes of {args.scene} (gpu, jit={args.jit}): {bad} mismatching scenes, {skipped} not lowered, "
          f"{merged} of {total} programs took the merged launches, {in_set} were served by a set module, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
