"""Build the native pieces of the package in-tree (hipcc for gfx950, g++ for the host library)."""
import os
import subprocess
import sys
import sysconfig

_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_DIR, "csrc")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-fPIC", "-shared", "-Wno-unused-value"]
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def runtime_lib_path():
    return os.path.join(_DIR, "libhanabi_amd.so")


def host_module_path():
    return os.path.join(_DIR, "_hanabi_host" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_runtime(force=False, verbose=False):
    """hanabi_amd.hip -> libhanabi_amd.so (HIP kernels + C ABI). hipcc cross-compiles without a GPU."""
    srcs = [os.path.join(_CSRC, f) for f in ("hanabi_amd.hip", "hnb_kernels.hip.h", "hnb_dev.h", "hnb_vm.h", "hnb_math.h")]
    srcs.append(os.path.join(_DIR, "..", "include", "hanabi_amd.h"))
    out = runtime_lib_path()
    if force or _newer(out, srcs):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + HIP_FLAGS + [srcs[0], "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=_CSRC)
    return out


def build_host(force=False, verbose=False):
    """hanabi:: host library + pybind11 module -> _hanabi_host*.so."""
    import pybind11

    hdir = os.path.join(_CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in ("pybind.cpp", "hanabi.cpp", "lowering.cpp")]
    deps = srcs + [os.path.join(hdir, "hanabi.hpp"), os.path.join(_DIR, "..", "include", "hanabi_amd.h")]
    out = host_module_path()
    if force or _newer(out, deps):
        cmd = ["g++"] + HOST_FLAGS + ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]] + srcs + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    return build_runtime(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
