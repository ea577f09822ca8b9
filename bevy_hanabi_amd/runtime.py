"""ctypes binding of the C ABI (include/hanabi_amd.h): the only way Python reaches the GPU path.

There is no CPU fallback here: if libhanabi_amd.so is missing or no HIP device exists the
calls raise HanabiError.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

HNB_OK = 0
HNB_ERR_NO_DEVICE = -4
HNB_ERR_NOT_FOUND = -6

ATTR_COMPONENTS = [1, 1, 3, 3, 1, 1, 1, 4, 1, 1, 2, 3, 1, 1, 3, 3, 3, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 1, 1, 1, 1, 1]
ATTR_IS_FLOAT = [0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0] + [1] * 16 + [0] * 5

# every entry point include/hanabi_amd.h declares
ABI_SYMBOLS = [
    "hnb_last_error", "hnb_version", "hnb_ctx_create", "hnb_ctx_destroy", "hnb_ctx_set_stream", "hnb_ctx_synchronize",
    "hnb_program_create", "hnb_program_destroy", "hnb_program_validate", "hnb_effect_create", "hnb_effect_destroy",
    "hnb_effect_set_parent", "hnb_frame_begin", "hnb_effect_set_frame", "hnb_effect_set_property", "hnb_simulate",
    "hnb_effect_metadata", "hnb_effect_alive_count", "hnb_effect_read_attr", "hnb_effect_read_alive_list",
    "hnb_effect_read_dead_list", "hnb_effect_write_attr", "hnb_effect_sort_ribbons", "hnb_ctx_enable_kernel_timing",
    "hnb_ctx_kernel_timing", "hnb_program_kernel_info", "hnb_jit_precompile", "hnb_effect_set_simulated", "hnb_ctx_set_option", "hnb_program_set_frames", "hnb_effect_index",
    "hnb_ctx_profile_marker", "hnb_program_kernel_timing",
    "hnb_comm_create_local", "hnb_comm_unique_id", "hnb_comm_create_rank", "hnb_comm_allreduce_alive", "hnb_comm_destroy", "hnb_comm_set_library",
    "hnb_effect_device_view", "hnb_effect_materialise", "hnb_jit_precompile_set", "hnb_effect_check", "hnb_effect_compare", "hnb_comm_describe", "hnb_program_device_view",
]

# hnb_ctx_set_option (include/hanabi_amd.h): name -> option id
OPTIONS = {"list_order": 1, "alternate": 2, "skip_lists": 3, "age_cohort": 4, "cull_lifetime": 5, "horizon": 6, "transpose": 7, "scene_merge": 8,
           "suffix_proof": 9, "overlap_updates": 10, "stream_hints": 11, "set_module": 12, "jit_async": 13, "test_break_proof": 15, "ring_lists": 16, "slot_init": 17, "direct_upload": 18}
SET_MODULE_OFF, SET_MODULE_CACHED, SET_MODULE_COMPILE, SET_MODULE_BACKGROUND = 0, 1, 2, 3


class HanabiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


class SimParams(C.Structure):
    _fields_ = [("delta_time", C.c_float), ("time", C.c_float), ("virtual_delta_time", C.c_float), ("virtual_time", C.c_float),
                ("real_delta_time", C.c_float), ("real_time", C.c_float)]


class EffectMetadata(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("capacity", "alive_count", "max_update", "max_spawn", "indirect_write_index",
                                          "particle_counter", "instance_count", "dispatch_x", "dead_count", "spawned", "fault",
                                          "reserved")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


class DeviceMeta(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("alive_count", "particle_counter", "list_column", "max_update", "dead_count", "spawned",
                                          "indirect_write_index", "instance_count")]


class EffectCheck(C.Structure):
    """HnbEffectCheck (hnb_effect_check): the list / alive-byte / lifetime invariants of one effect, evaluated on the device."""
    _fields_ = [(n, C.c_uint32) for n in ("ok", "capacity", "alive_count", "bad_slots", "duplicate_slots", "alive_byte_mismatches", "alive_past_lifetime", "fault")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class EffectDiff(C.Structure):
    """HnbEffectDiff (hnb_effect_compare): two effects of one layout compared bit for bit on the device."""
    _fields_ = [("equal", C.c_uint32), ("counter_diffs", C.c_uint32), ("alive_list_diffs", C.c_uint64), ("dead_list_diffs", C.c_uint64), ("attr_diffs", C.c_uint64),
                ("first_section", C.c_int32), ("reserved", C.c_uint32), ("first_index", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


class DeviceAttr(C.Structure):
    _fields_ = [("attr", C.c_uint16), ("ncomp", C.c_uint8), ("scalar_type", C.c_uint8), ("stride_bytes", C.c_uint16), ("reserved", C.c_uint16),
                ("plane", C.c_void_p)]


class DeviceView(C.Structure):
    """HnbDeviceView: device pointers of an effect's planes, lists and metadata row (hnb_effect_device_view)."""
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("capacity", C.c_uint32), ("slot_base", C.c_uint32), ("n_attrs", C.c_uint32), ("reserved", C.c_uint32),
                ("stale_attr_mask", C.c_uint64), ("alive_list", C.c_void_p * 2), ("dead_list", C.c_void_p),
                ("meta", C.c_void_p), ("meta_next", C.c_void_p), ("attrs", DeviceAttr * 40)]

    def plane(self, attr_id):
        for i in range(self.n_attrs):
            if self.attrs[i].attr == int(attr_id):
                return self.attrs[i]
        raise KeyError(attr_id)


class ProgramAttr(C.Structure):
    _fields_ = [("attr", C.c_uint16), ("ncomp", C.c_uint8), ("scalar_type", C.c_uint8), ("stride_bytes", C.c_uint16), ("reserved", C.c_uint16),
                ("plane_off", C.c_uint64)]


class ProgramView(C.Structure):
    """HnbProgramView: every instance of a program behind one device-resident table (hnb_program_device_view)."""
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("capacity", C.c_uint32), ("n_instances", C.c_uint32), ("n_attrs", C.c_uint32), ("reserved", C.c_uint32),
                ("stale_attr_mask", C.c_uint64), ("slabs", C.c_void_p), ("meta", C.c_void_p), ("meta_next", C.c_void_p),
                ("alive_list_off", C.c_uint64 * 2), ("dead_list_off", C.c_uint64), ("attrs", ProgramAttr * 40)]


_lib = None


def load_library():
    """Load libhanabi_amd.so (in-tree). Raises if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("HNB_LIB") or _build.runtime_lib_path()  # HNB_LIB: kernel-variant A/B runs
        if not os.path.exists(path):
            raise HanabiError(-1, f"{path} not found: run `python -m bevy_hanabi_amd.build` (needs hipcc)")
        lib = C.CDLL(path)
        lib.hnb_last_error.restype = C.c_char_p
        lib.hnb_version.restype = C.c_char_p
        lib.hnb_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.hnb_ctx_destroy.argtypes = [C.c_void_p]
        lib.hnb_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.hnb_ctx_synchronize.argtypes = [C.c_void_p]
        lib.hnb_program_create.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.hnb_program_destroy.argtypes = [C.c_void_p]
        lib.hnb_program_validate.argtypes = [C.c_char_p, C.c_size_t]
        lib.hnb_effect_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.hnb_effect_destroy.argtypes = [C.c_void_p]
        lib.hnb_effect_set_parent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.hnb_frame_begin.argtypes = [C.c_void_p, C.POINTER(SimParams)]
        lib.hnb_effect_set_frame.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.hnb_effect_set_property.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32]
        lib.hnb_simulate.argtypes = [C.c_void_p]
        lib.hnb_effect_metadata.argtypes = [C.c_void_p, C.POINTER(EffectMetadata)]
        lib.hnb_effect_alive_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.hnb_effect_read_attr.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.hnb_effect_write_attr.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.hnb_effect_read_alive_list.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.hnb_effect_read_dead_list.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.hnb_effect_sort_ribbons.argtypes = [C.c_void_p]
        lib.hnb_ctx_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        lib.hnb_ctx_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        lib.hnb_effect_set_simulated.argtypes = [C.c_void_p, C.c_int]
        lib.hnb_ctx_set_option.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.hnb_program_set_frames.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.hnb_effect_index.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.hnb_ctx_profile_marker.argtypes = [C.c_void_p, C.c_uint32]
        lib.hnb_program_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        lib.hnb_comm_create_local.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p)]
        lib.hnb_comm_unique_id.argtypes = [C.c_void_p]
        lib.hnb_comm_create_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.hnb_comm_allreduce_alive.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint64)]
        lib.hnb_comm_destroy.argtypes = [C.c_void_p]
        lib.hnb_comm_set_library.argtypes = [C.c_char_p, C.c_uint32]
        lib.hnb_program_device_view.argtypes = [C.c_void_p, C.POINTER(ProgramView)]
        lib.hnb_comm_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.hnb_effect_check.argtypes = [C.c_void_p, C.POINTER(EffectCheck)]
        lib.hnb_effect_compare.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(EffectDiff)]
        lib.hnb_effect_device_view.argtypes = [C.c_void_p, C.POINTER(DeviceView)]
        lib.hnb_effect_materialise.argtypes = [C.c_void_p, C.c_uint64]
        lib.hnb_program_kernel_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.hnb_jit_precompile.argtypes = [C.c_char_p, C.c_size_t]
        lib.hnb_jit_precompile_set.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_uint32]
        _lib = lib
    return _lib


def _check(rc):
    if rc != HNB_OK:
        raise HanabiError(rc, load_library().hnb_last_error().decode())


def validate_program(blob: bytes):
    """Structural validation of a program blob; works without a GPU."""
    _check(load_library().hnb_program_validate(blob, len(blob)))


def jit_precompile(blob: bytes):
    """Compile and cache the kernels specialised for this program (hiprtc; needs no GPU)."""
    _check(load_library().hnb_jit_precompile(blob, len(blob)))


def jit_precompile_set(blobs):
    """Compile and cache the set module of these programs (HNB_OPT_SET_MODULE: the specialised code of the small programs of a context behind
    their shared launches; needs no GPU). Order and duplicates do not matter."""
    blobs = list(blobs)
    arr = (C.c_char_p * len(blobs))(*blobs)
    sizes = (C.c_size_t * len(blobs))(*[len(b) for b in blobs])
    _check(load_library().hnb_jit_precompile_set(arr, sizes, len(blobs)))


class Context:
    """One simulation context per GPU (`hnb_ctx_*`)."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.hnb_ctx_create(device_id, C.byref(self._h)))
        self._programs = []
        # A/B harness convenience (tools/, bench.py): HNB_CTX_OPTIONS="age_cohort=0,horizon=0" is applied to every context THIS BINDING creates.
        # The library itself reads no environment variable for its options (include/hanabi_amd.h, hnb_ctx_set_option).
        for kv in filter(None, os.environ.get("HNB_CTX_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), int(v))

    def close(self):
        if self._h:
            for p in list(self._programs):
                p._h = None
                for e in p._effects:
                    e._h = None
            self._lib.hnb_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(self._lib.hnb_ctx_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        _check(self._lib.hnb_ctx_synchronize(self._h))

    def set_list_order(self, order):
        """'spawn' (default: the serial-thread order) or 'slot' (lists kept in increasing slot order; programs created afterwards)."""
        _check(self._lib.hnb_ctx_set_option(self._h, 1, {"spawn": 0, "slot": 1}[order]))

    def set_option(self, option, value):
        """hnb_ctx_set_option by id or by name (OPTIONS): "age_cohort" (0 off / 1 lean stacks / 2 all; fixed in a program at creation; the one
        option that changes device-visible state: the AGE plane), "cull_lifetime", "horizon", "alternate", "skip_lists", "transpose",
        "scene_merge", "suffix_proof" (scheduling choices; results do not change)."""
        _check(self._lib.hnb_ctx_set_option(self._h, OPTIONS[option] if isinstance(option, str) else int(option), int(value)))

    def create_program(self, blob: bytes):
        return Program(self, blob)

    def frame_begin(self, dt, time=0.0, virtual_dt=None, virtual_time=None, real_dt=None, real_time=None):
        sp = SimParams(dt, time, dt if virtual_dt is None else virtual_dt, time if virtual_time is None else virtual_time,
                       dt if real_dt is None else real_dt, time if real_time is None else real_time)
        _check(self._lib.hnb_frame_begin(self._h, C.byref(sp)))

    def simulate(self):
        _check(self._lib.hnb_simulate(self._h))

    def enable_kernel_timing(self, every_n_frames=1):
        """0/False = off; n = bracket the kernels of every n-th simulated frame with HIP events."""
        _check(self._lib.hnb_ctx_enable_kernel_timing(self._h, int(every_n_frames)))

    def profile_marker(self, tag):
        """An empty kernel with `tag` workgroups on the simulation stream: a visible cut in rocprofv3's dispatch list."""
        _check(self._lib.hnb_ctx_profile_marker(self._h, int(tag)))

    def kernel_timing(self):
        u, c, i, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        _check(self._lib.hnb_ctx_kernel_timing(self._h, C.byref(u), C.byref(c), C.byref(i), C.byref(n)))
        return {"update_ms_avg": u.value, "compact_ms_avg": c.value, "init_ms_avg": i.value, "frames": n.value}


def comm_set_library(path, duplicate_devices=False, single_rank=False):
    """hnb_comm_set_library: which collective library hnb_comm_* binds; before the first Comm of the process.
    single_rank: a communicator of one context / one rank goes through the library too (HNB_COMM_LIB_SINGLE_RANK)."""
    _check(load_library().hnb_comm_set_library(None if path is None else os.fsencode(path), (1 if duplicate_devices else 0) | (2 if single_rank else 0)))


class Comm:
    """The alive-counter all-reduce over RCCL (`hnb_comm_*`): `Comm.local(contexts)` for one process with a context per GPU,
    `Comm.rank(ctx, unique_id, rank, n_ranks)` for one rank per process (`Comm.unique_id()` on rank 0)."""

    def __init__(self, handle, contexts):
        self._lib = load_library()
        self._h = handle
        self._contexts = list(contexts)

    @classmethod
    def local(cls, contexts):
        lib = load_library()
        arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
        h = C.c_void_p()
        _check(lib.hnb_comm_create_local(arr, len(contexts), C.byref(h)))
        return cls(h, contexts)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(load_library().hnb_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rank(cls, ctx, unique_id, rank, n_ranks):
        lib = load_library()
        h = C.c_void_p()
        _check(lib.hnb_comm_create_rank(ctx._h, C.c_char_p(unique_id), int(rank), int(n_ranks), C.byref(h)))
        return cls(h, [ctx])

    def allreduce_alive(self, effects_per_context):
        """effects_per_context: one list of effects (or None) per local context, all of the same length -> totals per effect."""
        n = len(effects_per_context[0])
        assert len(effects_per_context) == len(self._contexts) and all(len(e) == n for e in effects_per_context)
        flat = [None if fx is None else fx._h for row in effects_per_context for fx in row]
        arr = (C.c_void_p * len(flat))(*flat)
        out = (C.c_uint64 * n)()
        _check(self._lib.hnb_comm_allreduce_alive(self._h, arr, n, out))
        return [int(v) for v in out]

    def describe(self):
        """hnb_comm_describe: 'rccl <resolved library path> ranks=N local=M' or 'host-sum ranks=N local=M'."""
        buf = C.create_string_buffer(1024)
        _check(self._lib.hnb_comm_describe(self._h, buf, len(buf)))
        return buf.value.decode()

    def destroy(self):
        if self._h:
            self._lib.hnb_comm_destroy(self._h)
            self._h = None


class Program:
    def __init__(self, ctx: Context, blob: bytes):
        self._ctx = ctx
        self._lib = ctx._lib
        self._h = C.c_void_p()
        _check(self._lib.hnb_program_create(ctx._h, blob, len(blob), C.byref(self._h)))
        self._effects = []
        ctx._programs.append(self)

    def create_effect(self, slot_base=0):
        return Effect(self, slot_base)

    def set_frames(self, spawn_counts, seeds, transforms=None, first=0):
        """Per-frame inputs of instances [first, first + n) in one call (creation order)."""
        sc = np.ascontiguousarray(spawn_counts, dtype=np.uint32)
        sd = np.ascontiguousarray(seeds, dtype=np.uint32)
        assert sc.shape == sd.shape
        xf = None if transforms is None else np.ascontiguousarray(transforms, dtype=np.float32).reshape(len(sc), 12)
        _check(self._lib.hnb_program_set_frames(self._h, int(first), len(sc), sc.ctypes.data, sd.ctypes.data, None if xf is None else xf.ctypes.data))

    def kernel_timing(self):
        """Context.kernel_timing() restricted to this program's kernels."""
        u, c, i, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        _check(self._lib.hnb_program_kernel_timing(self._h, C.byref(u), C.byref(c), C.byref(i), C.byref(n)))
        return {"update_ms_avg": u.value, "compact_ms_avg": c.value, "init_ms_avg": i.value, "frames": n.value}

    def device_view(self):
        """hnb_program_device_view: all instances behind one device-resident table (no synchronisation)."""
        v = ProgramView()
        _check(self._lib.hnb_program_device_view(self._h, C.byref(v)))
        return v

    def kernel_info(self):
        """Which kernels run this program: 'init=jit|interp|none update=aot-stream:<name>|jit-stream|jit-generic|interp-*'."""
        buf = C.create_string_buffer(4096)
        _check(self._lib.hnb_program_kernel_info(self._h, buf, len(buf)))
        return buf.value.decode()

    def destroy(self):
        if self._h:
            for e in self._effects:
                e._h = None
            _check(self._lib.hnb_program_destroy(self._h))
            self._h = None
            self._ctx._programs.remove(self)


class Effect:
    def __init__(self, prog: Program, slot_base=0):
        self._prog = prog
        self._lib = prog._lib
        self._h = C.c_void_p()
        _check(self._lib.hnb_effect_create(prog._h, slot_base, C.byref(self._h)))
        prog._effects.append(self)
        self.capacity = self.metadata()["capacity"]

    def destroy(self):
        if self._h:
            _check(self._lib.hnb_effect_destroy(self._h))
            self._h = None
            self._prog._effects.remove(self)

    def index(self):
        """Current position of this instance in its program's tables (hnb_effect_index): destroying an instance moves the
        last one into its place."""
        out = C.c_uint32()
        _check(self._lib.hnb_effect_index(self._h, C.byref(out)))
        return out.value

    def set_simulated(self, simulated=True):
        """False freezes the instance (SimulationCondition::WhenVisible while not visible)."""
        _check(self._lib.hnb_effect_set_simulated(self._h, int(bool(simulated))))

    def set_parent(self, parent, channel, event_capacity=256):
        """EffectParent: this effect's init consumes the spawn events `parent` appends on `channel`
        (EmitSpawnEventModifier.child_index). 256 events per frame is the reference's hard-coded capacity."""
        _check(self._lib.hnb_effect_set_parent(self._h, parent._h, int(channel), int(event_capacity)))
        self._parent = parent

    def set_frame(self, spawn_count, seed, transform=None):
        xf = None
        if transform is not None:
            xf = np.ascontiguousarray(np.asarray(transform, dtype=np.float32).reshape(12))
        _check(self._lib.hnb_effect_set_frame(self._h, int(spawn_count), int(seed) & 0xFFFFFFFF, None if xf is None else xf.ctypes.data))

    def set_property(self, name, values):
        v = np.atleast_1d(np.asarray(values))
        words = v.astype(np.float32).view(np.uint32) if v.dtype.kind == "f" else v.astype(np.uint32)
        words = np.ascontiguousarray(words)
        _check(self._lib.hnb_effect_set_property(self._h, name.encode(), words.ctypes.data, len(words)))

    def apply_properties(self, effect_properties):
        """Push every value of an `EffectProperties` the program declares (the reference uploads the component's values
        every frame they changed, src/render/mod.rs property upload)."""
        for name, _default, value in effect_properties.properties():
            try:
                self.set_property(name, value)
            except HanabiError as e:
                if e.code != HNB_ERR_NOT_FOUND:   # a value for a property this effect's asset does not declare is not an error
                    raise

    def device_view(self):
        """hnb_effect_device_view: the effect's device pointers after the frames enqueued so far (no synchronisation)."""
        v = DeviceView()
        _check(self._lib.hnb_effect_device_view(self._h, C.byref(v)))
        return v

    def materialise(self, attr_ids):
        """hnb_effect_materialise: make these attributes' planes current for device-side readers (enqueued on the simulation stream)."""
        mask = 0
        for a in attr_ids:
            mask |= 1 << int(a)
        _check(self._lib.hnb_effect_materialise(self._h, mask))

    def check(self):
        """hnb_effect_check: list permutation, alive bytes, age < lifetime, fault flag - on the device; a dict with "ok"."""
        c = EffectCheck()
        _check(self._lib.hnb_effect_check(self._h, C.byref(c)))
        return c.as_dict()

    def compare(self, other):
        """hnb_effect_compare: this effect against another of the same layout on the same device, bit for bit; a dict with "equal"."""
        d = EffectDiff()
        _check(self._lib.hnb_effect_compare(self._h, other._h, C.byref(d)))
        return d.as_dict()

    def metadata(self):
        m = EffectMetadata()
        _check(self._lib.hnb_effect_metadata(self._h, C.byref(m)))
        return m.as_dict()

    def alive_count(self):
        n = C.c_uint32()
        _check(self._lib.hnb_effect_alive_count(self._h, C.byref(n)))
        return int(n.value)

    def read_attr(self, attr_id):
        attr_id = int(attr_id)
        out = np.zeros((self.capacity, ATTR_COMPONENTS[attr_id]), dtype=np.uint32)
        _check(self._lib.hnb_effect_read_attr(self._h, attr_id, out.ctypes.data, out.nbytes))
        return out.view(np.float32) if ATTR_IS_FLOAT[attr_id] else out

    def write_attr(self, attr_id, array):
        a = np.ascontiguousarray(array)
        _check(self._lib.hnb_effect_write_attr(self._h, int(attr_id), a.ctypes.data, a.nbytes))

    def alive_list(self):
        out = np.zeros(max(self.alive_count(), 1), dtype=np.uint32)
        _check(self._lib.hnb_effect_read_alive_list(self._h, out.ctypes.data, len(out)))
        return out[: self.alive_count()]

    def dead_list(self):
        n = self.capacity - self.alive_count()
        out = np.zeros(max(n, 1), dtype=np.uint32)
        _check(self._lib.hnb_effect_read_dead_list(self._h, out.ctypes.data, len(out)))
        return out[:n]
