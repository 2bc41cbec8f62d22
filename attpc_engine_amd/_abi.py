"""ctypes binding of the C ABI declared in ``include/attpc_engine.h``.

This is the *only* place the Python package touches native code.  There is no CPU
fallback: if ``libattpc_hip.so`` is missing or no HIP device is present every
engine call raises (``EngineUnavailable``) instead of silently computing on the host.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

MAX_STEPS = 8
MAX_ROWS = 4 + 2 * (MAX_STEPS - 1)
MAX_SPECIES = 16
MAX_SIM = 8
DEDX_EMIN = -30
DEDX_EMAX = 14
DEDX_SUB = 32
DEDX_NODES = (DEDX_EMAX - DEDX_EMIN) * DEDX_SUB + 1
NUM_TB = 512
TIME_SAMPLES = 10001
LONG_STEPS = 5

EX_GAUSSIAN, EX_UNIFORM, EX_TABLE = 0, 1, 2
POLAR_UNIFORM, POLAR_ARBITRARY = 0, 1

ABI_VERSION = 3  # ATTPC_ABI_VERSION of include/attpc_engine.h this binding was written against

OK, E_INVALID, E_NODEVICE, E_HIP, E_CAPACITY, E_NOTCONFIGURED, E_DATALOSS = 0, 1, 2, 3, 4, 5, 6

_dp = C.POINTER(C.c_double)


class ExcitationDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("table_len", C.c_int32),
        ("p0", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
        ("table_x", _dp), ("table_cdf", _dp),
    ]


class PolarDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("table_len", C.c_int32),
        ("cos_min", C.c_double), ("cos_max", C.c_double), ("bin_width", C.c_double),
        ("angles", _dp), ("cdf", _dp),
    ]


class KinDesc(C.Structure):
    _fields_ = [
        ("n_steps", C.c_int32), ("sample_limit", C.c_int32),
        ("beam_energy", C.c_double),
        ("masses", C.c_double * MAX_ROWS),
        ("excitation", ExcitationDesc * MAX_STEPS),
        ("polar", PolarDesc * MAX_STEPS),
        ("has_target", C.c_int32), ("eloss_len", C.c_int32),
        ("rho_sigma", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
        ("eloss", _dp),
    ]


class SpeciesDesc(C.Structure):
    _fields_ = [("Z", C.c_int32), ("A", C.c_int32), ("mass", C.c_double), ("dedx", _dp)]


class DetDesc(C.Structure):
    _fields_ = [
        ("length", C.c_double), ("efield", C.c_double), ("bfield", C.c_double),
        ("density", C.c_double), ("diffusion", C.c_double), ("fano_factor", C.c_double),
        ("w_value", C.c_double),
        ("mpgd_gain", C.c_int64),
        ("micromegas_edge", C.c_int32), ("windows_edge", C.c_int32),
        ("pad_lut", C.POINTER(C.c_int16)),
        ("lut_n", C.c_int32), ("lut_lo", C.c_int32),
        ("n_species", C.c_int32), ("ode_substeps", C.c_int32),
        ("species", SpeciesDesc * MAX_SPECIES),
        ("longitudinal_diffusion", C.c_double),
        ("long_weights", C.c_double * 5),
        ("mc_diffusion", C.c_int32),
        ("reserved_ext", C.c_int32),
        ("path_step", C.c_double),
    ]


class EventLayout(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int32), ("n_sim", C.c_int32),
        ("indices", C.c_int32 * MAX_SIM),
        ("species_of_row", C.c_int32 * MAX_ROWS),
    ]


class CloudOut(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("offsets", C.POINTER(C.c_int64)),
        ("points", _dp),
        ("labels", C.POINTER(C.c_int64)),
        ("event_points", C.POINTER(C.c_int64)),
    ]


class SpyralDesc(C.Structure):
    _fields_ = [
        ("response", _dp), ("pad_centers", _dp), ("pad_sizes", _dp),
        ("n_pads", C.c_int32), ("windows_edge", C.c_int32), ("micromegas_edge", C.c_int32),
        ("reserved", C.c_int32),
        ("length", C.c_double), ("adc_threshold", C.c_double),
    ]


class RunStats(C.Structure):
    _fields_ = [
        ("n_events", C.c_uint64), ("n_points", C.c_uint64), ("n_track_samples", C.c_uint64),
        ("n_sample_limit", C.c_uint64), ("n_lds_overflow", C.c_uint64), ("n_failed", C.c_uint64),
        ("charge_checksum", C.c_uint64), ("key_checksum", C.c_uint64),
        ("ms_kinematics", C.c_double), ("ms_tracks", C.c_double), ("ms_scatter", C.c_double),
        ("launches_kinematics", C.c_uint32), ("launches_tracks", C.c_uint32),
        ("launches_scatter", C.c_uint32), ("n_inconsistent", C.c_uint32),
        ("n_lone_buckets", C.c_uint64), ("n_buffer_growths", C.c_uint64),
        ("n_tracks_capped", C.c_uint64), ("device_bytes", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


class EngineUnavailable(RuntimeError):
    """The HIP library or a HIP device is missing -- there is no CPU fallback."""


class DataLossError(RuntimeError):
    """A run finished but part of an event's charge is missing from its cloud
    (``stats.n_failed`` / ``stats.n_inconsistent`` != 0, status ATTPC_E_DATALOSS)."""


def dptr(arr: np.ndarray | None):
    if arr is None:
        return None
    assert arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(_dp)


def iptr(arr: np.ndarray | None, ctype):
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.POINTER(ctype))


def library_path() -> Path:
    env = os.environ.get("ATTPC_HIP_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "_lib" / "libattpc_hip.so"


# every symbol include/attpc_engine.h declares
EXPORTED_SYMBOLS = (
    "attpc_version", "attpc_device_count", "attpc_ctx_create", "attpc_ctx_destroy",
    "attpc_last_error", "attpc_set_chunk_events", "attpc_sync", "attpc_kin_configure",
    "attpc_kin_run", "attpc_kin_calculate", "attpc_decay_calculate", "attpc_det_configure", "attpc_det_run",
    "attpc_sim_run", "attpc_det_tracks", "attpc_spyral_rows", "attpc_spyral_configure", "attpc_sim_run_spyral",
    "attpc_set_option", "attpc_host_alloc", "attpc_host_free", "attpc_det_scatter", "attpc_unpack_rows",
    "attpc_unpack_spyral_rows", "attpc_det_run_spyral", "attpc_sim_hint_next", "attpc_unpack_rows8",
)

_lib = None


def load_library() -> C.CDLL:
    """Load libattpc_hip.so and declare prototypes (no device is touched here)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise EngineUnavailable(
            f"HIP engine library not found at {path}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "attpc_engine_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    ctxp = C.c_void_p
    lib.attpc_version.restype = C.c_int32
    if lib.attpc_version() != ABI_VERSION:
        raise EngineUnavailable(
            f"{path} implements ABI version {lib.attpc_version()}, this package needs {ABI_VERSION}: rebuild it "
            "(`python -c 'import __graft_entry__ as g; g.build(force=True)'`)")
    lib.attpc_device_count.restype = C.c_int32
    lib.attpc_ctx_create.argtypes = [C.c_int32, C.POINTER(ctxp)]
    lib.attpc_ctx_destroy.argtypes = [ctxp]
    lib.attpc_last_error.argtypes = [ctxp]
    lib.attpc_last_error.restype = C.c_char_p
    lib.attpc_set_chunk_events.argtypes = [ctxp, C.c_int32]
    lib.attpc_sync.argtypes = [ctxp]
    lib.attpc_kin_configure.argtypes = [ctxp, C.POINTER(KinDesc)]
    lib.attpc_kin_run.argtypes = [
        ctxp, C.c_uint64, C.c_uint64, C.c_uint64, _dp, _dp, C.POINTER(C.c_int32),
        C.POINTER(C.c_uint32),
    ]
    lib.attpc_kin_calculate.argtypes = [
        ctxp, C.c_uint64, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int32),
    ]
    lib.attpc_decay_calculate.argtypes = [
        ctxp, C.c_uint64, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, C.POINTER(C.c_int32),
    ]
    lib.attpc_det_configure.argtypes = [ctxp, C.POINTER(DetDesc)]
    lib.attpc_det_run.argtypes = [
        ctxp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(EventLayout), _dp, _dp,
        C.POINTER(CloudOut), C.POINTER(RunStats),
    ]
    lib.attpc_sim_run.argtypes = [
        ctxp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(EventLayout), _dp, _dp,
        C.POINTER(C.c_int32), C.POINTER(CloudOut), C.POINTER(RunStats),
    ]
    lib.attpc_sim_run_spyral.argtypes = lib.attpc_sim_run.argtypes
    lib.attpc_det_run_spyral.argtypes = lib.attpc_det_run.argtypes
    lib.attpc_sim_hint_next.argtypes = [ctxp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(EventLayout)]
    lib.attpc_spyral_configure.argtypes = [ctxp, C.POINTER(SpyralDesc)]
    lib.attpc_det_tracks.argtypes = [
        ctxp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(EventLayout), _dp, _dp, C.c_int64,
        _dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
    ]
    lib.attpc_det_scatter.argtypes = [
        ctxp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(EventLayout), _dp, C.POINTER(C.c_int32),
        C.POINTER(CloudOut), C.POINTER(RunStats),
    ]
    lib.attpc_set_option.argtypes = [ctxp, C.c_char_p, C.c_int64]
    lib.attpc_host_alloc.argtypes = [ctxp, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.attpc_host_free.argtypes = [ctxp, C.c_void_p]
    lib.attpc_unpack_rows.argtypes = [C.c_void_p, C.c_int64, _dp, C.POINTER(C.c_int64), C.c_int32]
    lib.attpc_unpack_rows8.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_uint64, C.c_uint64, _dp,
                                       C.POINTER(C.c_int64), C.c_int32]
    lib.attpc_unpack_spyral_rows.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                             C.c_double, _dp, C.POINTER(C.c_int64), C.c_int32]
    lib.attpc_spyral_rows.argtypes = [
        ctxp, C.c_int64, _dp, _dp, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_double, _dp,
    ]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:  # default -> int32 status
            fn.restype = C.c_int32
    _lib = lib
    return lib


class Context:
    """One engine context == one HIP device (single-threaded, like the reference objects)."""

    def __init__(self, device: int | None = None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("ATTPC_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        handle = C.c_void_p()
        status = self.lib.attpc_ctx_create(int(device), C.byref(handle))
        if status != OK:
            raise EngineUnavailable(
                f"attpc_ctx_create(device={device}) failed with status {status} "
                "(no HIP device?). attpc_engine_amd has no CPU fallback."
            )
        self.handle = handle
        self.device = device
        self._keepalive: list = []  # host arrays referenced by descriptors during configure

    def check(self, status: int, what: str) -> None:
        if status != OK:
            msg = self.lib.attpc_last_error(self.handle)
            text = msg.decode() if msg else ""
            if status == E_CAPACITY:
                raise BufferError(f"{what}: output capacity too small ({text})")
            if status == E_INVALID:
                raise ValueError(f"{what}: {text}")
            if status == E_DATALOSS:
                raise DataLossError(f"{what}: {text}")
            raise RuntimeError(f"{what} failed with status {status}: {text}")

    def set_option(self, name: str, value: int) -> None:
        """Tuning / test switches of the context (``attpc_set_option``)."""
        self.check(self.lib.attpc_set_option(self.handle, name.encode(), int(value)), f"attpc_set_option({name})")

    def pinned_empty(self, shape, dtype=np.float64) -> np.ndarray:
        """Uninitialised numpy array in page-locked host memory (``attpc_host_alloc``): device-to-host
        copies into it run at the PCIe rate.  The memory is returned when the array (and every view of
        it) is garbage collected, or with the context."""
        import weakref

        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n_bytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr = C.c_void_p()
        self.check(self.lib.attpc_host_alloc(self.handle, max(1, n_bytes), C.byref(ptr)), "attpc_host_alloc")
        raw = (C.c_char * max(1, n_bytes)).from_address(ptr.value)
        arr = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        lib, handle_ref, address = self.lib, weakref.ref(self), ptr.value

        def release():
            ctx = handle_ref()
            if ctx is not None and getattr(ctx, "handle", None):
                lib.attpc_host_free(ctx.handle, C.c_void_p(address))

        weakref.finalize(raw, release)
        return arr

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.attpc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
