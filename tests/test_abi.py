"""The drop-in boundary without a GPU: libhanabi_amd.so loads, exports every entry point that
include/hanabi_amd.h declares (and nothing in the header is missing from the binding), validates
program blobs, and fails loudly instead of falling back when no HIP device exists."""
import ctypes as C
import os
import re
import struct

import pytest
import torch

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hanabi_amd.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hnb_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = runtime.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/hanabi_amd.h but not exported"
    assert sorted(runtime.ABI_SYMBOLS) == names, "python binding list and header disagree"


def test_library_is_in_tree_and_has_no_torch_or_oracle_dependency():
    path = bh.build.runtime_lib_path()
    assert path.startswith(ROOT)
    needed = os.popen(f"readelf -d {path} 2>/dev/null").read()
    assert "libamdhip64" in needed
    assert "torch" not in needed and "oracle" not in needed and "python" not in needed


def test_version_string():
    assert b"gfx950" in runtime.load_library().hnb_version()


def test_validate_accepts_lowered_programs():
    for asset in (effects.single_particle(16), effects.firework_trails(1000), effects.force_field(1000), effects.instancing(1000), effects.ribbon(1000)):
        bh.validate_program(bh.lower(asset))


def test_validate_rejects_corrupt_blobs():
    blob = bytearray(bh.lower(effects.firework_trails(1000)))
    with pytest.raises(bh.HanabiError):
        bh.validate_program(bytes(blob[:40]))            # truncated header
    bad = bytearray(blob); bad[0:4] = b"XXXX"
    with pytest.raises(bh.HanabiError):
        bh.validate_program(bytes(bad))                  # wrong magic
    bad = bytearray(blob); struct.pack_into("<I", bad, 4, 999)
    with pytest.raises(bh.HanabiError):
        bh.validate_program(bytes(bad))                  # wrong version
    with pytest.raises(bh.HanabiError):
        bh.validate_program(bytes(blob[:-8]))            # code stream runs past the end
    with pytest.raises(bh.HanabiError):
        bh.validate_program(b"")


def test_validate_rejects_bad_instruction_operands():
    from bevy_hanabi_amd.runtime import load_library
    blob = bytearray(bh.lower(effects.firework_trails(1000)))
    # find the update stream (header: see HnbProgramHeader) and poison the first opcode
    hdr = struct.unpack_from("<24I", blob, 0)
    lib = load_library()
    ok = 0
    for off in range(len(blob) - 8, 64, -8):
        bad = bytearray(blob)
        bad[off] = 0xFE  # opcode byte of some instruction (or table data -> may stay valid)
        if lib.hnb_program_validate(bytes(bad), len(bad)) != 0:
            ok += 1
            break
    assert ok, "no corrupted opcode was rejected"
    assert hdr[0] == struct.unpack("<I", b"HNB2")[0]


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box without a GPU")
def test_no_device_is_a_loud_error_not_a_fallback():
    with pytest.raises(bh.HanabiError) as ei:
        bh.Context(0)
    assert ei.value.code in (runtime.HNB_ERR_NO_DEVICE, -5)
    assert b"" != runtime.load_library().hnb_last_error()


def test_null_arguments_are_rejected():
    lib = runtime.load_library()
    assert lib.hnb_ctx_create(0, None) != 0
    assert lib.hnb_simulate(None) != 0
    assert lib.hnb_frame_begin(None, None) != 0
    assert lib.hnb_effect_set_frame(None, 0, 0, None) != 0
    assert lib.hnb_ctx_destroy(None) == 0 and lib.hnb_program_destroy(None) == 0 and lib.hnb_effect_destroy(None) == 0


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under bevy_hanabi_amd/ may reference it."""
    pkg = os.path.join(ROOT, "bevy_hanabi_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(import oracle|from oracle)|#include[^\n]*oracle|libhanabi_oracle", src, flags=re.M), os.path.join(dp, f)
                if f == "hnb_comm.h":   # the one run-time library lookup of the product: RCCL, by name
                    assert "oracle" not in src and all("rccl" in m for m in re.findall(r'"([^"]*\.so[^"]*)"', src))
                else:
                    assert "dlopen" not in src, os.path.join(dp, f)


def test_jit_precompile_without_a_device(tmp_path, monkeypatch):
    """Kernel specialisation (hiprtc for gfx950) works on a box without a GPU and fills the cache; cache entries are
    self-describing and verified on load: a truncated, a corrupted and a foreign entry are all recompiled and replaced."""
    import bevy_hanabi_amd as bh
    from test_lowering_cpu import ZOO
    monkeypatch.setenv("HNB_JIT_CACHE", str(tmp_path / "cache"))
    cache = tmp_path / "cache"
    bh.jit_precompile(bh.lower(effects.firework_trails(2048)))          # init only (update has a pre-built kernel)
    entries = list(cache.glob("*.hnbjit"))
    assert len(entries) == 1
    assert (os.stat(cache).st_mode & 0o777) == 0o700                    # private directory
    name = sorted(ZOO)[0]
    bh.jit_precompile(bh.lower(ZOO[name]()))                            # a zoo program: init + update
    assert len(list(cache.glob("*.hnbjit"))) == 2
    bh.jit_precompile(bh.lower(effects.firework_trails(1 << 20)))       # capacity does not enter the key
    assert len(list(cache.glob("*.hnbjit"))) == 2
    good = entries[0].read_bytes()
    assert good[:7] == b"HNBJIT2" and b"_ZN3hnb" in good                # header + lowered kernel names + code object
    mtime = lambda: os.stat(entries[0]).st_mtime_ns
    t0 = mtime()
    bh.jit_precompile(bh.lower(effects.firework_trails(2048)))          # a valid entry is a hit: not rewritten
    assert mtime() == t0
    for bad in (good[: len(good) // 2],                                 # truncated
                good[:-16] + bytes(16),                                 # code object damaged
                good[:24] + bytes([good[24] ^ 1]) + good[25:]):         # built from something else (key hash differs)
        entries[0].write_bytes(bad)
        bh.jit_precompile(bh.lower(effects.firework_trails(2048)))
        fresh = entries[0].read_bytes()
        # replaced by a complete entry for the same key (magic, hiprtc version, both key hashes, key length: bytes 0..40). The code object itself is
        # not compared: with two kernels in the translation unit (k_init + k_init_slots, round 6) hiprtc's output is not bit-reproducible from one
        # compilation to the next inside a process (24 bytes of it differ; both objects are valid) ...
        assert fresh[:40] == good[:40] and len(fresh) == len(good) and fresh != bad
        t1 = mtime()
        bh.jit_precompile(bh.lower(effects.firework_trails(2048)))      # ... and the entry it wrote is a hit (header, length and code hash verified on load)
        assert mtime() == t1
    assert not list(cache.glob("*.tmp*"))
    with pytest.raises(bh.HanabiError):
        bh.jit_precompile(b"garbage")


def test_slab_layout_reaches_the_capacities_the_reference_allows():
    """EffectAsset::capacity is a u32 (src/asset.rs:391-415). Slab sections are addressed with 32-bit offsets in units of 256 bytes
    (every section is 256-byte aligned): 1 TiB per instance. The firework takes 65 B per slot: 2 G particles are 130 GB, within an
    MI355X's 288 GB - round 2 stopped at 4 GiB, 87.6M firework particles."""
    for cap in (80_000_000, 100_000_000, 200_000_000, 1_000_000_000, 2_000_000_000):
        bh.validate_program(bh.lower(effects.firework_trails(cap)))
    bh.validate_program(bh.lower(effects.ribbon(1_000_000_000)))


def test_batched_frame_inputs_are_declared():
    assert "hnb_program_set_frames" in declared_symbols() and "hnb_effect_index" in declared_symbols()


def test_integration_binding_lists_every_entry_point():
    """INTEGRATION.md's Rust `extern "C"` block mirrors the header one to one."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    bound = set(re.findall(r"pub fn (hnb_[a-z0-9_]+)\(", txt))
    host = open(os.path.join(ROOT, "include", "hanabi_amd_host.h")).read()
    host = re.sub(r"/\*.*?\*/", "", host, flags=re.S)
    host_syms = set(re.findall(r"\b(hnb_[a-z0-9_]+)\s*\(", host))
    assert bound == set(declared_symbols()) | host_syms   # device ABI (section 1) + host ABI (section 3)
