"""The boundary as a host application sees it: include/hanabi_amd.h is a C header, and examples/firework.cpp (the
reference's examples/firework.rs trails effect written against the C++ authoring mirror + the C ABI, no Python) builds,
fails loudly without a GPU and, on a GPU, produces what the Python binding produces for the same inputs."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "firework")


def _build_example():
    pkg = os.path.join(ROOT, "bevy_hanabi_amd")
    host = os.path.join(pkg, "csrc", "host")
    srcs = [os.path.join(ROOT, "examples", "firework.cpp")] + [os.path.join(host, f) for f in ("hanabi.cpp", "lowering.cpp", "wgsl.cpp")]
    deps = srcs + [os.path.join(host, "hanabi.hpp"), os.path.join(ROOT, "include", "hanabi_amd.h"), os.path.join(pkg, "libhanabi_amd.so")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + host] + srcs +
                              ["-L" + pkg, "-lhanabi_amd", "-Wl,-rpath," + pkg, "-o", EXE])
    return EXE


def test_header_is_plain_c99(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    src = tmp_path / "use_header.c"
    names = bh.runtime.ABI_SYMBOLS
    src.write_text('#include "hanabi_amd.h"\n' + "typedef void (*fn_t)(void);\nstatic const fn_t entry_points[] = {\n" + "".join(f"    (fn_t)&{n},\n" for n in names) + "};\n"
                   "int main(void) { return (int)(sizeof entry_points / sizeof entry_points[0]) - %d; }\n" % len(names))
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "use_header.o")])


def test_example_builds_and_fails_loudly_without_a_gpu():
    exe = _build_example()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe, "1000", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "hnb_ctx_create" in r.stderr   # no CPU fallback: the first call reports the missing device


@pytest.mark.gpu
def test_cpp_example_matches_the_python_binding():
    exe = _build_example()
    cap, frames = 50000, 90
    r = subprocess.run([exe, str(cap), str(frames)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [int(line.split()[1]) for line in r.stdout.strip().splitlines()]
    asset = effects.firework_trails(cap)
    ctx = bh.Context(0)
    prog = ctx.create_program(bh.lower(asset))
    fx = prog.create_effect()
    spawner, rng = bh.EffectSpawner(asset.spawner), bh.Pcg32()
    want = []
    for f in range(frames):
        ctx.frame_begin(np.float32(1.0) / np.float32(60.0), float(np.float32(f) * (np.float32(1.0) / np.float32(60.0))))
        fx.set_frame(spawner.tick(float(np.float32(1.0) / np.float32(60.0)), rng), (0x9E3779B9 * (f + 1)) & 0xFFFFFFFF)
        ctx.simulate()
        want.append(fx.alive_count())
    prog.destroy()
    ctx.close()
    assert got == want
    assert got[0] == cap and got[-1] == 0 and any(0 < a < cap for a in got)   # burst, die-off, empty
