import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Native pieces are built in-tree; on the GPU box the prebuilt .so files travel with the snapshot.
    from bevy_hanabi_amd import build as hb

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if os.path.exists(hipcc) or shutil.which("hipcc"):
        hb.build_runtime()
    hb.build_host()
    hb.build_host_lib()
    import oracle

    oracle.build()
    cvm_dir = os.path.join(ROOT, "tests", "cpu_vm")
    so = os.path.join(cvm_dir, "libcpu_vm.so")
    src = os.path.join(cvm_dir, "cpu_vm.cpp")
    deps = [src] + [os.path.join(ROOT, "bevy_hanabi_amd", "csrc", f) for f in ("hnb_vm.h", "hnb_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", src, "-o", so])
    # tests/cpu_plan: the frame-planning proofs of hnb_simulate (hnb_plan.h) compiled for the host
    cp_dir = os.path.join(ROOT, "tests", "cpu_plan")
    cp_so, cp_src = os.path.join(cp_dir, "libcpu_plan.so"), os.path.join(cp_dir, "cpu_plan.cpp")
    cp_dep = os.path.join(ROOT, "bevy_hanabi_amd", "csrc", "hnb_plan.h")
    if not os.path.exists(cp_so) or max(os.path.getmtime(cp_src), os.path.getmtime(cp_dep)) > os.path.getmtime(cp_so):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", cp_src, "-o", cp_so])
    # tests/fake_rccl: the stand-in collective library (built here, where the RCCL header is; the .so travels to the GPU box)
    fr_dir = os.path.join(ROOT, "tests", "fake_rccl")
    fr_so, fr_src = os.path.join(fr_dir, "libfake_rccl.so"), os.path.join(fr_dir, "fake_rccl.cpp")
    if os.path.exists("/opt/rocm/include/rccl/rccl.h") and (not os.path.exists(fr_so) or os.path.getmtime(fr_src) > os.path.getmtime(fr_so)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", fr_src, "-o", fr_so,
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])


    # tests/device_view: the HIP consumer kernel of the device-side output boundary
    hb.build_consumer()


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
