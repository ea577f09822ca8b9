"""Fill bevy_hanabi_amd/jit_cache/ with the specialised kernels of the programs the tests, the bench
and smoke() create. hiprtc needs no GPU, and the cache directory travels with the library, so a box
that runs these programs never has to compile them. Safe to skip: a missing entry is compiled on demand."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bevy_hanabi_amd as bh
    from bevy_hanabi_amd import effects

    assets = [effects.single_particle(16), effects.firework_trails(4096), effects.force_field(4096), effects.instancing(4096), effects.ribbon(4096)]
    assets += [effects.firework_rocket(), effects.firework_sparkle_trail(), effects.firework_trails_child()]
    from bevy_hanabi_amd import reference_examples
    for entries in reference_examples.catalog().values():
        assets += [e.asset for e in entries]
    try:
        from test_lowering_cpu import ZOO
        assets += [ZOO[k]() for k in sorted(ZOO)]
        from helpers import math_probe_asset
        assets.append(math_probe_asset(4096))
    except Exception as e:  # tests not present: product programs only
        print("warm_jit_cache: zoo skipped:", e)
    # entries are keyed by the generated source and the kernel headers: drop what older builds left behind
    cache = os.path.join(ROOT, "bevy_hanabi_amd", "jit_cache")
    if os.path.isdir(cache) and not os.environ.get("HNB_JIT_CACHE"):
        for f in os.listdir(cache):
            if f.endswith((".hsaco", ".names", ".hnbjit")):
                os.remove(os.path.join(cache, f))
    t0 = time.time()
    for a in assets:
        bh.jit_precompile(bh.lower(a))
    print(f"warm_jit_cache: {len(assets)} programs in {time.time() - t0:.1f} s")
    # Set modules (HNB_OPT_SET_MODULE): the scene of every single-entity example effect (bench.py's small_effects_scene, tests/test_scene_merge.py)
    # and the small sets of tests/test_set_module.py
    t0 = time.time()
    sets = []
    cat = reference_examples.catalog()
    sets.append([e.asset for name in sorted(cat) if all(x.parent is None for x in cat[name]) for e in cat[name]])
    try:
        from test_set_module import warm_sets
        sets += warm_sets()
    except Exception as e:
        print("warm_jit_cache: test sets skipped:", e)
    for members in sets:
        bh.jit_precompile_set([bh.lower(a) for a in members])
    print(f"warm_jit_cache: {len(sets)} set modules in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
