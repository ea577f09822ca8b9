"""bevy_hanabi_amd — MI355X-native particle simulation hot path behind bevy_hanabi's effect API.

Layout:
  csrc/            HIP kernels + C ABI (libhanabi_amd.so), deterministic math, program interpreters
  csrc/host/       hanabi:: C++ mirror of the reference authoring API + lowering, exposed as `_hanabi_host`
  runtime.py       ctypes binding of include/hanabi_amd.h
  effects.py       the reference's example assets restated as SURVEY.md §8(d) configurations
"""
from . import build  # noqa: F401

try:
    from ._hanabi_host import *  # noqa: F401,F403
except ImportError:  # a fresh checkout: the host library (g++, no GPU toolchain needed) is built in-tree on first import
    try:
        build.build_host(verbose=True)
        from ._hanabi_host import *  # noqa: F401,F403
    except Exception as _e:  # pragma: no cover
        raise ImportError("bevy_hanabi_amd._hanabi_host could not be built: run `python -m bevy_hanabi_amd.build`") from _e
from . import _hanabi_host as host  # noqa: F401,E402

from .runtime import Comm, Context, Effect, EffectMetadata, HanabiError, Program, SimParams, jit_precompile, jit_precompile_set, validate_program  # noqa: F401,E402
