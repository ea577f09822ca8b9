"""The host-side C ABI (include/hanabi_amd_host.h, libhanabi_host.so): authoring + lowering for hosts that are not C++.

CPU tests: the header is plain C99 and every declared entry point is exported; every asset of the suite (the reference's
examples and the modifier / operator zoo), rebuilt call by call through the C ABI from its flat description, lowers to the
SAME program blob as through the C++ / Python mirror; the reference's failure modes come back as status codes; the spawner
reproduces the reference's tick sequences; examples/firework_c99.c (C99, no C++ anywhere) produces the firework program.
GPU test: that C program simulates the effect and its particles equal the oracle's.
"""
import ctypes as C
import os
import re
import shutil
import struct
import subprocess

import numpy as np
import pytest

import bevy_hanabi_amd as bh
from bevy_hanabi_amd import build as hb
from bevy_hanabi_amd import effects

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hanabi_amd_host.h")
EXE = os.path.join(ROOT, "examples", "firework_c99")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hnb_[a-z0-9_]+)\s*\(", txt)))


class Value(C.Structure):
    _fields_ = [("scalar_type", C.c_uint32), ("count", C.c_uint32), ("bits", C.c_uint32 * 4)]


class CpuValue(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("uniform", C.c_uint32)]


class SpawnerSettings(C.Structure):
    _fields_ = [("count", CpuValue), ("spawn_duration", CpuValue), ("period", CpuValue), ("cycle_count", C.c_uint32),
                ("starts_active", C.c_uint32), ("emit_on_start", C.c_uint32)]


class ModifierDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("attribute", C.c_uint32), ("e", C.c_uint32 * 7), ("dimension", C.c_uint32), ("kill_inside", C.c_uint32),
                ("condition", C.c_uint32), ("child_index", C.c_uint32), ("n_render_attrs", C.c_uint32), ("render_attrs", C.c_uint32 * 8)]


@pytest.fixture(scope="module")
def lib():
    l = C.CDLL(hb.build_host_lib())
    l.hnb_host_last_error.restype = C.c_char_p
    l.hnb_host_free.argtypes = [C.c_void_p]
    l.hnb_spawner_settings_new.argtypes = [CpuValue, CpuValue, CpuValue, C.c_uint32, C.POINTER(SpawnerSettings)]
    l.hnb_spawner_settings_once.argtypes = [CpuValue, C.POINTER(SpawnerSettings)]
    l.hnb_spawner_settings_rate.argtypes = [CpuValue, C.POINTER(SpawnerSettings)]
    l.hnb_spawner_settings_burst.argtypes = [CpuValue, CpuValue, C.POINTER(SpawnerSettings)]
    l.hnb_spawner_create.argtypes = [C.POINTER(SpawnerSettings), C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
    l.hnb_spawner_tick.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_uint32)]
    l.hnb_spawner_destroy.argtypes = [C.c_void_p]
    l.hnb_module_destroy.argtypes = [C.c_void_p]
    l.hnb_asset_destroy.argtypes = [C.c_void_p]
    for fn in ("hnb_module_lit", "hnb_module_attr", "hnb_module_parent_attr", "hnb_module_add_property", "hnb_module_prop", "hnb_module_builtin",
               "hnb_module_unary", "hnb_module_binary", "hnb_module_ternary", "hnb_module_cast", "hnb_asset_add_modifier", "hnb_asset_set_simulation_space",
               "hnb_asset_set_simulation_condition", "hnb_asset_set_motion_integration", "hnb_asset_set_prng_seed", "hnb_lower", "hnb_asset_serialize",
               "hnb_asset_particle_layout", "hnb_asset_set_name", "hnb_module_num_expressions"):
        getattr(l, fn).restype = C.c_int
    return l


def test_header_is_c99_and_every_symbol_is_exported(lib, tmp_path):
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/hanabi_amd_host.h but not exported by libhanabi_host.so"
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    src = tmp_path / "use_host_header.c"
    src.write_text('#include "hanabi_amd_host.h"\ntypedef void (*fn_t)(void);\nstatic const fn_t entry_points[] = {\n' + "".join(f"    (fn_t)&{n},\n" for n in names) +
                   "};\nint main(void) { return (int)(sizeof entry_points / sizeof entry_points[0]) - %d; }\n" % len(names))
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "x.o")])
    needed = os.popen(f"readelf -d {hb.host_lib_path()} 2>/dev/null").read()
    assert "amdhip" not in needed and "python" not in needed and "torch" not in needed   # authoring + lowering need no GPU runtime


def replay_through_c_abi(lib, asset):
    """Rebuild `asset` through the C entry points from its flat description (hanabi::serialize_asset): what a binding from
    another language does with its own Module / modifier lists. Returns (program blob, flat description) as the C ABI gives them."""
    raw = bh.serialize_asset(asset)
    hdr = struct.unpack_from("<12I", raw, 0)
    magic, _ver, cap, space, cond, motion, seed, n_exprs, n_props, n_init, n_update, n_render = hdr
    assert magic == 0x31534148
    sp = struct.unpack_from("<ffIffIffIIII", raw, 48)
    off = 96
    exprs = [struct.unpack_from("<14I", raw, off + 56 * i) for i in range(n_exprs)]
    off += 56 * n_exprs
    props = []
    for i in range(n_props):
        name = raw[off:off + 48].split(b"\0")[0]
        elem, count, b0, b1, b2, b3 = struct.unpack_from("<6I", raw, off + 48)
        props.append((name, elem, count, (b0, b1, b2, b3)))
        off += 72
    mods = [struct.unpack_from("<22I", raw, off + 88 * i) for i in range(n_init + n_update + n_render)]
    ok = lambda rc: (_ for _ in ()).throw(AssertionError(lib.hnb_host_last_error().decode())) if rc != 0 else None

    m = C.c_void_p()
    ok(lib.hnb_module_create(C.byref(m)))
    for name, elem, count, bits in props:
        v = Value(elem, count, (C.c_uint32 * 4)(*bits))
        h = C.c_uint32()
        ok(lib.hnb_module_add_property(m, name, C.byref(v), C.byref(h)))
    for i, (kind, op, a, b, c, vt_elem, vt_count, b0, b1, b2, b3, attr, prop, _pad) in enumerate(exprs):
        h = C.c_uint32()
        if kind == 0:
            ok(lib.hnb_module_builtin(m, op, vt_elem, vt_count, C.byref(h)))
        elif kind == 1:
            v = Value(vt_elem, vt_count, (C.c_uint32 * 4)(b0, b1, b2, b3))
            ok(lib.hnb_module_lit(m, C.byref(v), C.byref(h)))
        elif kind == 2:
            ok(lib.hnb_module_prop(m, prop, C.byref(h)))
        elif kind == 3:
            ok(lib.hnb_module_attr(m, attr, C.byref(h)))
        elif kind == 4:
            ok(lib.hnb_module_parent_attr(m, attr, C.byref(h)))
        elif kind == 5:
            ok(lib.hnb_module_unary(m, op, a, C.byref(h)))
        elif kind == 6:
            ok(lib.hnb_module_binary(m, op, a, b, C.byref(h)))
        elif kind == 7:
            ok(lib.hnb_module_ternary(m, op, a, b, c, C.byref(h)))
        elif kind == 8:
            ok(lib.hnb_module_cast(m, a, vt_elem, vt_count, C.byref(h)))
        else:
            raise AssertionError(f"expression kind {kind}")
        assert h.value == i + 1   # handles are 1-based indices in both worlds
    s = SpawnerSettings(CpuValue(sp[0], sp[1], sp[2]), CpuValue(sp[3], sp[4], sp[5]), CpuValue(sp[6], sp[7], sp[8]), sp[9], sp[10], sp[11])
    a = C.c_void_p()
    ok(lib.hnb_asset_create(cap, C.byref(s), m, C.byref(a)))
    ok(lib.hnb_module_destroy(m))
    ok(lib.hnb_asset_set_simulation_space(a, space))
    ok(lib.hnb_asset_set_simulation_condition(a, cond))
    ok(lib.hnb_asset_set_motion_integration(a, motion))
    ok(lib.hnb_asset_set_prng_seed(a, seed))
    for i, md in enumerate(mods):
        ctx = 1 if i < n_init else 2 if i < n_init + n_update else 4
        d = ModifierDesc()
        d.kind, d.attribute = md[0], md[1]
        for k in range(7):
            d.e[k] = md[2 + k]
        flags = md[9]
        d.kill_inside = 1 if flags & 4 else 0
        d.dimension, d.condition, d.child_index, d.n_render_attrs = md[10], md[11], md[12], md[13]
        for k in range(8):
            d.render_attrs[k] = md[14 + k]
        ok(lib.hnb_asset_add_modifier(a, ctx, C.byref(d)))
    out, size = C.c_void_p(), C.c_size_t()
    ok(lib.hnb_lower(a, C.byref(out), C.byref(size)))
    blob = C.string_at(out, size.value)
    lib.hnb_host_free(out)
    ok(lib.hnb_asset_serialize(a, C.byref(out), C.byref(size)))
    flat = C.string_at(out, size.value)
    lib.hnb_host_free(out)
    n = C.c_uint32()
    attrs = (C.c_uint32 * 39)()
    ok(lib.hnb_asset_particle_layout(a, attrs, 39, C.byref(n)))
    assert [attrs[i] for i in range(n.value)] == [x.id for x in asset.particle_layout()]
    ok(lib.hnb_asset_destroy(a))
    return blob, flat


def suite_assets():
    from test_lowering_cpu import ZOO
    out = {"single_particle": effects.single_particle(16), "firework_trails": effects.firework_trails(4096), "force_field": effects.force_field(4096),
           "instancing": effects.instancing(4096), "ribbon": effects.ribbon(4096), "rocket": effects.firework_rocket(),
           "sparkle_trail": effects.firework_sparkle_trail(), "trails_child": effects.firework_trails_child()}
    out.update({f"zoo/{k}": ZOO[k]() for k in sorted(ZOO)})
    return out


def test_every_suite_asset_lowers_identically_through_the_c_abi(lib):
    assets = suite_assets()
    assert len(assets) >= 20
    for name, asset in assets.items():
        blob, flat = replay_through_c_abi(lib, asset)
        assert flat == bh.serialize_asset(asset), name
        assert blob == bh.lower(asset), name
        bh.validate_program(blob)


def test_failure_modes_are_status_codes(lib):
    err = lambda: lib.hnb_host_last_error().decode()
    m = C.c_void_p()
    assert lib.hnb_module_create(C.byref(m)) == 0
    h = C.c_uint32()
    assert lib.hnb_module_unary(m, 0, 7, C.byref(h)) == -1 and "handle" in err()            # operand that is not in the module (the mirror panics)
    assert lib.hnb_module_unary(m, 99, 1, C.byref(h)) == -1
    bad = Value(7, 1, (C.c_uint32 * 4)(0, 0, 0, 0))
    assert lib.hnb_module_lit(m, C.byref(bad), C.byref(h)) == -1
    one = Value(1, 1, (C.c_uint32 * 4)(0x3F800000, 0, 0, 0))
    assert lib.hnb_module_lit(m, C.byref(one), C.byref(h)) == 0 and h.value == 1
    assert lib.hnb_module_prop(m, 3, C.byref(h)) == -1                                      # unknown property
    assert lib.hnb_module_attr(m, 1000, C.byref(h)) == -1
    s = SpawnerSettings()
    neg = CpuValue(-1.0, -1.0, 0)
    cnt = CpuValue(5.0, 5.0, 0)
    assert lib.hnb_spawner_settings_new(cnt, CpuValue(1, 1, 0), neg, 0, C.byref(s)) == -1 and "period" in err()   # spawn.rs:299-311 panics
    assert lib.hnb_spawner_settings_new(cnt, CpuValue(1, 1, 0), neg, 1, C.byref(s)) == 0                          # ... unless cycle_count == 1
    assert lib.hnb_spawner_settings_once(cnt, C.byref(s)) == 0 and s.cycle_count == 1 and s.emit_on_start == 1 and s.starts_active == 1
    a = C.c_void_p()
    assert lib.hnb_asset_create(100, C.byref(s), m, C.byref(a)) == 0
    d = ModifierDesc()
    d.kind, d.e[0] = 9, 1      # AccelModifier: update only
    assert lib.hnb_asset_add_modifier(a, 1, C.byref(d)) == -1                               # .init(AccelModifier) panics in the reference (asset.rs:482)
    d.e[0] = 42
    assert lib.hnb_asset_add_modifier(a, 2, C.byref(d)) == -8 and "#42" in err()            # an expression the module does not have
    d.e[0] = 1
    assert lib.hnb_asset_add_modifier(a, 2, C.byref(d)) == 0
    assert lib.hnb_asset_add_modifier(a, 3, C.byref(d)) == -1                               # one context at a time
    out, size = C.c_void_p(), C.c_size_t()
    assert lib.hnb_lower(a, C.byref(out), C.byref(size)) == -2 and "POSITION" in err().upper()   # ShaderGenerateError: no POSITION attribute (lib.rs:838-845)
    assert lib.hnb_lower(None, C.byref(out), C.byref(size)) == -1
    lib.hnb_asset_destroy(a)
    lib.hnb_module_destroy(m)


def test_spawner_reproduces_the_reference_sequences(lib):
    """spawn.rs:1044-1287 through the C ABI: new(3, 3, 10, 2) ticked 2, 5, 8, 10, 0.1 -> 2, 1, 3, 0, 0; rate(5): 1.01 -> 5, 0.4 -> 2."""
    def run(settings_call, ticks):
        s = SpawnerSettings()
        assert settings_call(C.byref(s)) == 0
        sp = C.c_void_p()
        assert lib.hnb_spawner_create(C.byref(s), 1, 2, C.byref(sp)) == 0
        out = []
        for dt in ticks:
            n = C.c_uint32()
            assert lib.hnb_spawner_tick(sp, dt, C.byref(n)) == 0
            out.append(n.value)
        lib.hnb_spawner_destroy(sp)
        return out
    cv = lambda x: CpuValue(x, x, 0)
    assert run(lambda p: lib.hnb_spawner_settings_new(cv(3), cv(3), cv(10), 2, p), [2.0, 5.0, 8.0, 10.0, 0.1]) == [2, 1, 3, 0, 0]
    assert run(lambda p: lib.hnb_spawner_settings_rate(cv(5), p), [1.01, 0.4]) == [5, 2]
    assert run(lambda p: lib.hnb_spawner_settings_once(cv(5), p), [0.001, 100.0]) == [5, 0]
    assert run(lambda p: lib.hnb_spawner_settings_burst(cv(5), cv(2), p), [1.0, 4.0, 0.1]) == [5, 10, 0]


def _build_c99_example():
    pkg = os.path.join(ROOT, "bevy_hanabi_amd")
    src = os.path.join(ROOT, "examples", "firework_c99.c")
    deps = [src, HEADER, os.path.join(ROOT, "include", "hanabi_amd.h"), hb.build_host_lib(), hb.runtime_lib_path()]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"), src,
                               "-L" + pkg, "-lhanabi_host", "-lhanabi_amd", "-Wl,-rpath," + pkg, "-o", EXE])
    return EXE


def test_c99_program_builds_the_firework_program(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists(hb.runtime_lib_path()):
        pytest.skip("needs gcc and the built runtime library")
    exe = _build_c99_example()
    for cap in (1000, 1 << 24):
        out = tmp_path / f"fw_{cap}.blob"
        subprocess.check_call([exe, "lower", str(cap), str(out)], stdout=subprocess.DEVNULL)
        assert out.read_bytes() == bh.lower(effects.firework_trails(cap))
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:   # no CPU fallback: the device ABI's first call reports the missing device
        r = subprocess.run([exe, "run", "1000", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "hnb_ctx_create" in r.stderr


@pytest.mark.gpu
def test_c99_program_simulates_what_the_oracle_computes(tmp_path):
    import oracle
    exe = _build_c99_example()
    cap, frames = 30000, 66
    dump = tmp_path / "pos.f32"
    r = subprocess.run([exe, "run", str(cap), str(frames), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got_alive = [int(line.split()[1]) for line in r.stdout.strip().splitlines()]
    asset = effects.firework_trails(cap)
    orc = oracle.OracleEffect(bh.serialize_asset(asset))
    spawner, rng = bh.EffectSpawner(asset.spawner), bh.Pcg32()
    dt = float(np.float32(1.0) / np.float32(60.0))
    want_alive = []
    for f in range(frames):
        orc.step(dt, spawner.tick(dt, rng), (0x9E3779B9 * (f + 1)) & 0xFFFFFFFF, time=float(np.float32(f) * np.float32(dt)))
        want_alive.append(orc.alive_count())
    assert got_alive == want_alive and 0 < got_alive[-1] < cap
    pos = np.fromfile(dump, dtype=np.uint32).reshape(cap, 3)
    np.testing.assert_array_equal(pos, orc.read_attr(2).view(np.uint32))


def test_emitted_wgsl_through_the_c_abi():
    """hnb_asset_wgsl: the text EffectShaderSources::generate would paste into the templates, one `// {{SLOT}}` section per template slot."""
    from bevy_hanabi_amd import _hanabi_host as hh
    lib = C.CDLL(hb.build_host_lib())
    lib.hnb_host_free.argtypes = [C.c_void_p]
    lib.hnb_asset_destroy.argtypes = [C.c_void_p]
    asset = effects.force_field(1000)
    ron = bh.to_ron(asset).encode()
    a = C.c_void_p()
    assert lib.hnb_asset_from_ron(ron, len(ron), C.byref(a)) == 0
    t, n = C.c_char_p(), C.c_size_t()
    assert lib.hnb_asset_wgsl(a, 0, C.byref(t), C.byref(n)) == 0
    text = C.string_at(t, n.value).decode()
    w = hh.generate_wgsl(asset)
    for slot, key in (("INIT_EXTRA", "init_extra"), ("INIT_CODE", "init_code"), ("UPDATE_CODE", "update_code"), ("REAP_CODE", "reap_code"), ("WRITEBACK_CODE", "writeback_code")):
        assert "// {{" + slot + "}}\n" + w[key] in text, slot
    assert "fn force_field_" in text and "let shell_factor = smoothstep(0., shell_half_thickness, abs(surface_dist));" in text
    assert "particle_buffer.particles[base_particle + particle_index].velocity = particle.velocity;" in text
    lib.hnb_asset_destroy(a)
