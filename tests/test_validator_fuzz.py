"""hnb_program_validate is the memory-safety boundary of the C ABI: a program it accepts must not be able to address
anything outside its register files, parameter block, attribute table or planes. Mutated program blobs (bit flips in
instruction words, random operand bytes, header and attribute-table edits) are validated; every ACCEPTED mutant then runs
through the host build of the product interpreters with bounds-checked register files under AddressSanitizer
(tests/validator_fuzz/run_blobs.cpp). CPU only; the kernels execute the same vm_exec on the same operands."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from fuzz_assets import random_asset, random_typed_asset
from test_fuzz import uniform_heavy_asset, wide_asset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER_WORDS = 24   # HnbProgramHeader: 24 u32 (include/hanabi_amd.h)


def _driver(tmp_path_factory):
    out = os.path.join(ROOT, "tests", "validator_fuzz", "run_blobs")
    src = os.path.join(ROOT, "tests", "validator_fuzz", "run_blobs.cpp")
    deps = [src, os.path.join(ROOT, "tests", "cpu_vm", "cpu_vm.cpp")] + [os.path.join(ROOT, "bevy_hanabi_amd", "csrc", f) for f in ("hnb_vm.h", "hnb_math.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-ffp-contract=off",
                               "-Wno-unknown-pragmas", "-DHNB_VM_BOUNDS_CHECK", src, "-o", out])
    return out


def _bases():
    blobs = [bh.lower(random_asset(s, 64)) for s in range(6)] + [bh.lower(random_typed_asset(2000 + s, 64)) for s in range(10)]
    blobs += [bh.lower(wide_asset(12, 64)), bh.lower(uniform_heavy_asset(20, True, 64)), bh.lower(uniform_heavy_asset(20, False, 64))]
    return blobs


def _mutate(rng, blob):
    b = bytearray(blob)
    words = struct.unpack_from(f"<{HEADER_WORDS}I", b)
    uniform_len, init_len, update_len = words[8], words[9], words[10]
    attrs_off, uniform_off, init_off, update_off = words[14], words[16], words[17], words[18]
    n_ins = uniform_len + init_len + update_len
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.random()
        if kind < 0.45 and n_ins:        # a random bit of a random instruction
            i = uniform_off + int(rng.integers(n_ins)) * 8 + int(rng.integers(8))
            b[i] ^= 1 << int(rng.integers(8))
        elif kind < 0.75 and n_ins:      # a random operand / opcode byte
            i = uniform_off + int(rng.integers(n_ins)) * 8 + int(rng.integers(5))
            b[i] = int(rng.integers(256))
        elif kind < 0.9:                 # a header field other than magic / version / total_size / capacity
            f = int(rng.choice([4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21]))
            v = words[f]
            v = int(rng.choice([v + 1, max(v - 1, 0), v + 8, int(rng.integers(0, 300)), v ^ (1 << int(rng.integers(12)))]))
            struct.pack_into("<I", b, f * 4, v & 0xFFFFFFFF)
        else:                            # a byte of the attribute table
            n_attrs = words[5]
            if n_attrs:
                i = attrs_off + int(rng.integers(n_attrs * 8))
                b[i] = int(rng.integers(64))
    return bytes(b)


@pytest.mark.parametrize("seed", [1, 2])
def test_accepted_mutants_stay_in_bounds(seed, tmp_path, tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    driver = _driver(tmp_path_factory)
    rng = np.random.default_rng(seed)
    bases = _bases()
    accepted = rejected = 0
    for k in range(6000):
        m = _mutate(rng, bases[int(rng.integers(len(bases)))])
        try:
            bh.validate_program(m)
        except bh.HanabiError:
            rejected += 1
            continue
        with open(tmp_path / f"m{k:05d}.blob", "wb") as f:
            f.write(m)
        accepted += 1
    assert accepted > 300 and rejected > 300, (accepted, rejected)   # the mutations exercise both outcomes
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([driver, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    tail = "\n".join(r.stderr.splitlines()[-25:])
    assert r.returncode == 0, f"accepted {accepted} mutants; driver exit {r.returncode}\n{tail}"
    assert f"{accepted} blobs ran clean" in r.stdout
