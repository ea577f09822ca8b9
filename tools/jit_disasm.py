"""Development tool: disassemble the kernels specialised (hiprtc) for one of the bench assets and count instruction classes.
Usage: python tools/jit_disasm.py c2|c3|c4|c5 [--dump]"""
import os, sys, struct, subprocess, tempfile, collections, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = tempfile.mkdtemp(prefix="hnbjit_")
os.environ["HNB_JIT_CACHE"] = d
import bevy_hanabi_amd as bh
from bevy_hanabi_amd import effects
asset = {"c2": effects.firework_trails, "c3": effects.force_field, "c4": effects.instancing, "c5": effects.ribbon}[sys.argv[1]](65536)
bh.jit_precompile(bh.lower(asset))
for f in os.listdir(d):
    raw = open(os.path.join(d, f), "rb").read()
    # CacheHeader: magic[8], u32 x2, u64 x3, u64 x2, u32 x2
    magic, maj, mino, ka, kb, kl, code_size, code_hash, n_names, names_bytes = struct.unpack_from("<8sIIQQQQQII", raw, 0)
    hdr = struct.calcsize("<8sIIQQQQQII")
    hdr = (hdr + 7) // 8 * 8
    code = raw[hdr + names_bytes: hdr + names_bytes + code_size]
    co = os.path.join(d, f + ".co")
    open(co, "wb").write(code)
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)[\s\S]*?\.name:\s+(\S+)[\s\S]*?\.private_segment_fixed_size:\s+(\d+)[\s\S]*?\.vgpr_count:\s+(\d+)", notes):
        print(f"{m.group(2)[:60]}: LDS {m.group(1)} B, scratch {m.group(3)} B, {m.group(4)} VGPRs")
    cur, counts = None, {}
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = m.group(1); counts[cur] = collections.Counter(); continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
        if m and cur: counts[cur][m.group(1)] += 1
    for k, c in counts.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:70]
        tot = sum(c.values())
        valu = sum(v for i, v in c.items() if i.startswith("v_"))
        f64 = sum(v for i, v in c.items() if "f64" in i)
        quarter = sum(v for i, v in c.items() if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_i64|rcp|rsq|sqrt|exp|log|sin|cos|div_scale|div_fmas|div_fixup|ldexp|frexp|trig)", i))
        mem = sum(v for i, v in c.items() if i.startswith(("global_", "flat_", "scratch_", "buffer_")))
        print(f"{name}: {tot} instrs, VALU {valu} (f64 {f64}, quarter-rate-ish {quarter}), vmem {mem}, salu {sum(v for i, v in c.items() if i.startswith('s_'))}")
        print("   top:", ", ".join(f"{i} {v}" for i, v in c.most_common(28)))
    if "--dump" in sys.argv:
        open("/tmp/jit_disasm.s", "w").write(txt); print("wrote /tmp/jit_disasm.s")
