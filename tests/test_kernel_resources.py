"""The compiled kernels use the LDS and scratch their sources declare, and no more.

Round 3 found a struct of pointer arrays indexed by a run-time parity, and a struct copy through possibly aliasing pointers, moved into LDS
by the compiler (40 B and 12 B per thread: 10 KiB and 3 KiB per workgroup); every launch of the affected small kernels (k_init, k_count_rows,
k_compact) took 25 us longer (profiles/r03j_kernels.log). Nothing fails when that happens, so the code object's own metadata is checked here.
"""
import os
import re
import subprocess

import pytest

from bevy_hanabi_amd import build as _build

LLVM = "/opt/rocm/lib/llvm/bin"

# LDS bytes the sources declare, per kernel-name prefix (the streaming update kernels: transpose buffers + per-wave words)
LDS_BUDGET = {
    "k_reset_lists": 0, "k_spawn_mark": 0, "k_init_slots": 0, "k_check_rows": 0, "k_compare_words": 0, "k_probe_placement": 0, "k_materialise_age": 0, "k_marker": 0, "k_gather_alive": 0, "k_sort_merge": 0,
    "k_init": 32, "k_init_jobs": 32, "k_update_generic_wide_jobs": 32, "k_update_jobs": 24704, "k_count_rows": 16, "k_count_rows_multi": 16, "k_compact": 16, "k_compact_multi": 16, "k_emit_count": 16, "k_order_count": 16,
    "k_update_slots_generic": 32, "k_sort_hist": 1024, "k_sort_fill": 1152, "k_sort_small": 5120, "k_sort_scatter": 5120, "k_sort_tile": 6280,
    "k_order_write": 32784, "k_emit_events": 49188, "k_update_slots_stream": 24672, "k_update_slots_stream_age": 64,
}
# kernels allowed to use scratch: the interpreters (a register file indexed by the instruction stream) and the 5/6-wave streaming variants
# (a handful of spilled words under their register budget)
SCRATCH_OK = re.compile(r"InterpCode|k_update_slots_stream|k_update_generic_wide_jobs")


def _kernels():
    lib = _build.runtime_lib_path()
    if not os.path.exists(lib) or not os.path.exists(f"{LLVM}/clang-offload-bundler"):
        pytest.skip("runtime library or LLVM tools not present")
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"hnb_co_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "gfx950.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    demangle = subprocess.run([f"{LLVM}/llvm-cxxfilt"], input=notes, capture_output=True, text=True).stdout if os.path.exists(f"{LLVM}/llvm-cxxfilt") else notes
    out, cur = [], {}
    for line in demangle.splitlines():
        m = re.match(r"\s+\.(group_segment_fixed_size|private_segment_fixed_size|name):\s+(.*)", line)
        if not m:
            continue
        cur[m.group(1)] = m.group(2).strip()
        if len(cur) == 3:
            out.append((cur["name"], int(cur["group_segment_fixed_size"]), int(cur["private_segment_fixed_size"])))
            cur = {}
    assert len(out) > 20, "no kernel metadata found"
    return out


def _short(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name


def test_lds_per_kernel_is_what_the_sources_declare():
    for name, lds, _ in _kernels():
        k = _short(name)
        assert k in LDS_BUDGET, f"kernel {name}: add its declared LDS size to LDS_BUDGET"
        assert lds <= LDS_BUDGET[k], f"{name}: {lds} B of LDS per workgroup, the sources declare {LDS_BUDGET[k]} (a private array promoted to LDS?)"


def test_scratch_only_where_expected():
    for name, _, scratch in _kernels():
        if scratch:
            assert SCRATCH_OK.search(name), f"{name}: {scratch} B of scratch per thread"
            assert scratch <= 640, f"{name}: {scratch} B of scratch per thread"
