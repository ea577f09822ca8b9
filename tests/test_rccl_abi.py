"""hnb_comm_decl.h - the RCCL declarations the product binds by hand (dlsym, no build-time dependency) - pinned against the installed
RCCL header: tests/rccl_abi/check_rccl_abi.cpp static_asserts every enum value and every function signature.
The negative cases prove that the check can fail."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bevy_hanabi_amd", "csrc")
TU = os.path.join(ROOT, "tests", "rccl_abi", "check_rccl_abi.cpp")
RCCL_H = "/opt/rocm/include/rccl/rccl.h"

pytestmark = pytest.mark.skipif(not os.path.exists(RCCL_H), reason="no RCCL header on this box (the GPU box has the same image: the check runs where the header is)")


def compile_against(decl_dir):
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + decl_dir, TU],
                          capture_output=True, text=True)


def test_hand_written_declarations_match_rccl_h():
    r = compile_against(CSRC)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("pattern,replacement,expect", [
    (r"kNcclUint64 = 5", "kNcclUint64 = 4", "ncclUint64"),
    (r"kNcclSum = 0", "kNcclSum = 1", "ncclSum"),
    (r"kNcclUniqueIdBytes = 128", "kNcclUniqueIdBytes = 64", "NCCL_UNIQUE_ID_BYTES"),
    (r"\(const void\*, void\*, size_t, int, int, ncclComm_t, hipStream_t\)", "(const void*, void*, int, int, int, ncclComm_t, hipStream_t)", "ncclAllReduce"),
    (r"\(ncclComm_t\*, int, ncclUniqueId, int\)", "(ncclComm_t*, int, ncclUniqueId*, int)", "ncclCommInitRank"),
])
def test_a_drifted_declaration_fails_the_check(pattern, replacement, expect):
    text = open(os.path.join(CSRC, "hnb_comm_decl.h")).read()
    broken, n = re.subn(pattern, replacement, text)
    assert n == 1
    d = tempfile.mkdtemp(prefix="hnb_rccl_abi_")
    try:
        open(os.path.join(d, "hnb_comm_decl.h"), "w").write(broken)
        r = compile_against(d)
        assert r.returncode != 0 and expect in r.stderr, r.stderr[-400:]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_the_product_uses_exactly_the_pinned_declarations():
    # hnb_comm.h takes its types from hnb_comm_decl.h and declares no RCCL value or signature of its own
    text = open(os.path.join(CSRC, "hnb_comm.h")).read()
    assert '#include "hnb_comm_decl.h"' in text
    assert "kNcclUint64 =" not in text and "typedef struct ncclComm" not in text
    for f in ("GetUniqueId", "CommInitRank", "CommInitAll", "CommDestroy", "AllReduce", "GroupStart", "GroupEnd", "GetErrorString"):
        assert re.search(rf"{f}_fn {f} = nullptr;", text), f
