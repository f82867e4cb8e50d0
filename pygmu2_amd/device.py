"""
ctypes binding of libpygmu_hip.so (include/pygmu_hip.h) and the device-array type that
backs Snippet payloads.

There is no CPU fallback: if the shared library is missing, or no MI355X-class device can
be initialised, every call that needs the device raises RuntimeError.
"""

from __future__ import annotations

import atexit
import ctypes as C
import os
import threading

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# PGX_LIB_PATH: load another build of the same library (kernel experiments under experiments/)
LIB_PATH = os.environ.get("PGX_LIB_PATH") or os.path.join(_PKG_DIR, "libpygmu_hip.so")

_lib = None
_lib_lock = threading.Lock()
_initialised = False

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_i32_p = C.POINTER(C.c_int32)

# (name, restype, argtypes).  Device pointers are passed as c_void_p integers.
_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_D = C.c_double
_F = C.c_float
_Z = C.c_size_t

_SIGNATURES = [
    ("pgx_abi_version", _I, []),
    ("pgx_last_error", C.c_char_p, []),
    ("pgx_device_count", _I, [C.POINTER(_I)]),
    ("pgx_init", _I, [_I]),
    ("pgx_shutdown", _I, []),
    ("pgx_device_name", _I, [C.c_char_p, _Z]),
    ("pgx_stream_handle", _P, []),
    ("pgx_stream_sync", _I, []),
    ("pgx_stream_fork", _I, []),
    ("pgx_stream_select", _I, [_I]),
    ("pgx_stream_join", _I, []),
    ("pgx_stream_is_forked", _I, []),
    ("pgx_stream_detach", _I, []),
    ("pgx_stream_fork_after", _I, [_P]),
    ("pgx_stream_wait_detached", _I, []),
    ("pgx_stream_is_detached", _I, []),
    ("pgx_malloc", _I, [C.POINTER(_P), _Z]),
    ("pgx_free", _I, [_P]),
    ("pgx_pool_trim", _I, []),
    ("pgx_memset", _I, [_P, _I, _Z]),
    ("pgx_memcpy_h2d", _I, [_P, _P, _Z]),
    ("pgx_memcpy_d2h", _I, [_P, _P, _Z]),
    ("pgx_memcpy_d2d", _I, [_P, _P, _Z]),
    ("pgx_host_malloc", _I, [C.POINTER(_P), _Z]),
    ("pgx_host_free", _I, [_P]),
    ("pgx_d2h_begin", _I, [_P, _P, _Z, C.POINTER(_L)]),
    ("pgx_d2h_wait", _I, [_L]),
    ("pgx_d2h_query", _I, [_L, C.POINTER(_I)]),
    ("pgx_d2h_fence", _I, [_L]),
    ("pgx_event_create", _I, [C.POINTER(_P)]),
    ("pgx_event_destroy", _I, [_P]),
    ("pgx_event_record", _I, [_P]),
    ("pgx_event_elapsed_ms", _I, [_P, _P, C.POINTER(_F)]),
    ("pgx_selftest_sincos", _I, [_P, _P, _P, _L]),
    ("pgx_selftest_tanh", _I, [_P, _P, _L]),
    ("pgx_fill", _I, [_P, _L, _F]),
    ("pgx_ramp", _I, [_P, _F, _F, _L, _I]),
    ("pgx_ramp_blocks", _I, [_P, _L, _L, _I, _L]),
    ("pgx_dirac", _I, [_P, _L, _L, _I]),
    ("pgx_window_copy", _I, [_P, _L, _L, _I, _P, _L, _L, _I, _I]),
    ("pgx_extract_channel", _I, [_P, _P, _L, _I, _I]),
    ("pgx_sine_render", _I, [_P, _L, _I, _L, _L, _I, _D, _P]),
    ("pgx_sine_gain_render", _I, [_P, _L, _I, _L, _L, _I, _D, _P, _F]),
    ("pgx_sine_stateful", _I, [_P, _L, _I, _D, _P, _P, _P, _P, _P]),
    ("pgx_gain_const", _I, [_P, _P, _L, _F]),
    ("pgx_gain_vec", _I, [_P, _P, _P, _L, _I, _I]),
    ("pgx_mix_n", _I, [_P, C.POINTER(_P), _I, _L]),
    ("pgx_mix_batch", _I, [_P, _P, _L, _I, _L]),
    ("pgx_gain_mix_batch", _I, [_P, _P, _L, _P, _L, _I, _L, _I, _I]),
    ("pgx_biquad_workspace_bytes", _Z, [_I, _L, _I, _L]),
    ("pgx_biquad_table_doubles", _Z, []),
    ("pgx_biquad_tables", _I, [_P, _P, _I]),
    ("pgx_biquad_const", _I, [_P, _L, _P, _L, _I, _L, _I, _P, _P, _L, _P, _P]),
    ("pgx_biquad_sine_supported", _I, [_L, _L]),
    ("pgx_biquad_sine", _I, [_P, _L, _L, _D, _D, _D, _D, _P, _P, _L, _P, _P]),
    ("pgx_biquad_varying", _I, [_P, _P, _L, _I, _D, _P, _P, _P, _D, _D, _P, _P]),
    ("pgx_scan2_workspace_bytes", _Z, [_L, _I]),
    ("pgx_convolve_fft_size", _L, [_L]),
    ("pgx_convolve_fft_spectrum_bytes", _Z, [_L, _I]),
    ("pgx_convolve_fft_workspace_bytes", _Z, [_L, _L, _I, _L]),
    ("pgx_convolve_fft_prepare", _I, [_P, _P, _L, _I, _L]),
    ("pgx_convolve_fft", _I, [_P, _P, _L, _I, _P, _L, _I, _I, _L, _P, _P, _I]),
    ("pgx_channel_adapt", _I, [_P, _P, _L, _I, _I]),
    ("pgx_pan", _I, [_P, _P, _L, _I, C.c_float, _P, _I]),
    ("pgx_mono_mean", _I, [_P, _P, _L, _I]),
    ("pgx_loop", _I, [_P, _P, _L, _L, _I, _L, _L, _L]),
    ("pgx_window_workspace_bytes", _Z, [_L, _I, _L]),
    ("pgx_window", _I, [_P, _P, _L, _I, _L, _I, _I, _P]),
    ("pgx_dynamics", _I, [_P, _P, _P, _L, _I, _I, _P]),
    ("pgx_gate_stateful", _I, [_P, _L, _D, _D, _D, _D, _P, _P, _P, _P]),
    ("pgx_interp_lookup", _I, [_P, _P, _L, _L, _I, _L, _L, _D, _P, _I, _I, _D, _D]),
    ("pgx_index_range", _I, [_P, _P, _L, _L]),
    ("pgx_stream_range", _I, [_P, _I, _P, _L]),
    ("pgx_piecewise", _I, [_P, _L, _L, _I, _P, _P, _I, _I, _I, _I]),
    ("pgx_f32_to_pcm16", _I, [_P, _P, _L]),
    ("pgx_pcm16_to_f32", _I, [_P, _P, _L]),
    ("pgx_svf", _I, [_P, _P, _L, _I, _D, _P, _P, _P, _D, _P, _P, _P]),
    ("pgx_envelope_scratch_bytes", _Z, [_L, _I]),
    ("pgx_envelope", _I, [_P, _P, _L, _I, _D, _D, _I, _I, _L, _P, _P]),
    ("pgx_transform", _I, [_P, _P, _L, _P, _I]),
    ("pgx_blitsaw", _I, [_P, _L, _I, _L, _I, _D, _P, _P, _L, _P, _L, _P, _L, _P, _P, _P]),
    ("pgx_blitsaw_workspace_bytes", _Z, [_I, _L, _I]),
    ("pgx_supersaw_sum", _I, [_P, _L, _I, _I, _L, _I, _P, _P, _P, _L]),
    ("pgx_blitsaw_biquad_bank", _I, [_P, _L, _I, _L, _D, _P, _P, _P, _P]),
    ("pgx_blitsaw_biquad_wide", _I, [_P, _L, _I, _L, _P, _P, _P, _P, _P, _P, _L]),
    ("pgx_blitsaw_biquad_wide_segments", _I, [_I, _L, _L]),
    ("pgx_blitsaw_biquad_wide_seg", _I, [_P, _L, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L]),
    ("pgx_voice_tiles_max_warm", _L, []),
    ("pgx_voice_tiles_table_bytes", _Z, [_I]),
    ("pgx_voice_tiles_tables", _I, [_P, _P, _P, _P, _I]),
    ("pgx_voice_tiles_workspace_bytes", _Z, [_I, _L, _L]),
    ("pgx_voice_tiles_entries", _I, [_P, _I, _I, _L, _P, _P, _L, _L]),
    ("pgx_voice_tiles", _I, [_P, _I, _L, _P, _P, _P, _P, _P, _P, _L, _L, _P, _I, _I]),
    ("pgx_supersaw_bank", _I, [_P, _L, _I, _I, _L, _I, _D, _P, _P, _P]),
    ("pgx_supersaw_bank_segments", _I, [_I, _L]),
    ("pgx_supersaw_bank_seg", _I, [_P, _L, _I, _I, _L, _I, _D, _P, _P, _P, _P, _P]),
    ("pgx_supersaw_bank_table_bytes", _Z, [_I, _I]),
    ("pgx_supersaw_bank_tables", _I, [_P, _I, _I, _D, _P]),
    ("pgx_supersaw_wide_table_bytes", _Z, [_I, _I]),
    ("pgx_supersaw_wide_tables", _I, [_P, _I, _I, _D, _P]),
    ("pgx_supersaw_wide_segments", _I, [_I, _I, _L]),
    ("pgx_supersaw_wide", _I, [_P, _L, _I, _I, _L, _I, _P, _P, _P, _P]),
    ("pgx_ladder", _I, [_P, _L, _P, _L, _I, _L, _I, _D, _P, _P, _P, _P, _P, _L, _L, _P]),
    ("pgx_ladder_workspace_bytes", _Z, [_I, _L, _I, _L]),
    ("pgx_comb", _I, [_P, _L, _P, _L, _I, _L, _I, _D, _P, _I, _I, _P, _P, _D, _L, _P, _L, _L, _I, _P, _P]),
    ("pgx_comb_workspace_bytes", _Z, [_I, _L, _I, _I, _I]),
    ("pgx_periodic_gate", _I, [_P, _L, _I, _L, _L, _P]),
    ("pgx_periodic_trigger", _I, [_P, _L, _L, _L, _L, _F]),
    ("pgx_adsr_workspace_bytes", _Z, [_I, _L]),
    ("pgx_adsr_gated", _I, [_P, _L, _P, _L, _I, _L, _P, _P, _P]),
    ("pgx_adsr_gated_periodic", _I, [_P, _L, _I, _L, _L, _P, _P, _P, _P, _I]),
    ("pgx_adsr_gated_periodic_to", _I, [_P, _L, _I, _L, _L, _P, _P, _P, _P, _P, _I]),
    ("pgx_adsr_triggered", _I, [_P, _L, _P, _L, _I, _L, _L, _P, _P, _P]),
    ("pgx_convolve_workspace_bytes", _Z, [_L, _L, _I]),
    ("pgx_convolve", _I, [_P, _P, _L, _I, _P, _L, _I, _I, _P, _P]),
    ("pgx_comm_unique_id_bytes", _Z, []),
    ("pgx_comm_unique_id", _I, [_P, _Z]),
    ("pgx_comm_init", _I, [_I, _I, _P, _Z]),
    ("pgx_comm_info", _I, [C.POINTER(_I), C.POINTER(_I)]),
    ("pgx_comm_destroy", _I, []),
    ("pgx_comm_quiesce", _I, [_I]),
    ("pgx_comm_abandoned", _I, []),
    ("pgx_comm_fold_check", _I, [_L]),
    ("pgx_comm_stats", _I, [C.POINTER(_L), C.POINTER(_L), C.POINTER(_L)]),
    ("pgx_allreduce_sum", _I, [_P, _P, _Z, C.POINTER(_L)]),
    ("pgx_allreduce_wait", _I, [_L]),
    ("pgx_allreduce_scalar_host", _I, [C.POINTER(_D), _I]),
]

EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

# numpy mirrors of the parameter structs in include/pygmu_hip.h
SINE_PARAMS = np.dtype([("w", "<f8"), ("amp", "<f8"), ("phase0", "<f8")])
SINE_STATEFUL_PARAMS = np.dtype([("freq", "<f8"), ("amp", "<f8"), ("phase", "<f8"),
                                 ("phase_is_stream", "<i4"), ("pad", "<i4")])
BIQUAD_VAR_PARAMS = np.dtype([("freq", "<f8"), ("q", "<f8"), ("gain_db", "<f8"),
                              ("mode", "<i4"), ("pad", "<i4")])
DYNAMICS_PARAMS = np.dtype([("mode", "<i4"), ("soft", "<i4"), ("stereo_link", "<i4"), ("wide_makeup", "<i4"),
                            ("threshold", "<f4"), ("slope", "<f4"), ("neg_slope", "<f4"), ("half_knee", "<f4"),
                            ("two_knee", "<f4"), ("knee", "<f4"), ("knee_lo", "<f4"), ("knee_hi", "<f4"),
                            ("gate_range", "<f4"), ("makeup", "<f4"), ("gate_range_d", "<f8"), ("makeup_d", "<f8")])
BLITSAW_PARAMS = np.dtype([("freq", "<f8"), ("amp", "<f8"), ("leak", "<f8"), ("m", "<f8")])
LADDER_PARAMS = np.dtype([("freq", "<f8"), ("resonance", "<f8"), ("drive", "<f8"),
                          ("passband_gain", "<f8"), ("oversample", "<i4"), ("mode", "<i4")])
COMB_PARAMS = np.dtype([("feedback", "<f8"), ("delay", "<i4"), ("buffer_len", "<i4")])
TRANSFORM_OP = np.dtype([("code", "<i4"), ("pad", "<i4"), ("p0", "<f8"), ("p1", "<f8")])
GATE_PARAMS = np.dtype([("dt", "<f8"), ("phase", "<f8"), ("duty", "<f8")])
ADSR_PARAMS = np.dtype([("attack_dvdt", "<f8"), ("decay_dvdt", "<f8"), ("release_dvdt", "<f8"),
                        ("sustain_level", "<f8"), ("sustain_samples", "<i8")])


def load_library():
    """dlopen libpygmu_hip.so and declare every prototype.  Does not touch the GPU."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP render library has not been built. "
                "Run `python -m pygmu2_amd.build` (needs hipcc); there is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in _SIGNATURES:
            fn = getattr(lib, name)      # AttributeError -> a missing export is a build bug
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class PgxError(RuntimeError):
    pass


class PgxOutOfMemory(PgxError, MemoryError):
    """PGX_ERR_NOMEM: the pool could not get the block from HIP (look_ahead falls back to block-by-block on it)."""


def check(code: int, what: str = "") -> None:
    """Map a pgx_status to a Python exception (ValueError for bad arguments, as the
    reference raises from render(); RuntimeError otherwise)."""
    if code == 0:
        return
    msg = load_library().pgx_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}" if what else msg
    if code == -1:
        raise ValueError(text)
    if code == -4:
        raise PgxOutOfMemory(f"{text} (pgx status {code})")
    raise PgxError(f"{text} (pgx status {code})")


def default_device_index() -> int:
    for var in ("PYGMU_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v != "":
            return int(v)
    return 0


def ensure_init(device: int | None = None):
    """Initialise the library on `device` (default: $PYGMU_DEVICE, $LOCAL_RANK, else 0)."""
    global _initialised
    lib = load_library()
    if not _initialised:
        dev = default_device_index() if device is None else int(device)
        n = C.c_int(0)
        rc = lib.pgx_device_count(C.byref(n))
        if rc != 0 or n.value == 0:
            raise RuntimeError(
                "pygmu2_amd needs an AMD GPU (MI355X / gfx950): no HIP device is visible. "
                "There is no CPU fallback for the render path.")
        check(lib.pgx_init(dev % n.value if device is None else dev), "pgx_init")
        _initialised = True
        global _exit_hook
        if not _exit_hook:
            atexit.register(_shutdown_at_exit)
            _exit_hook = True
    return lib


_exit_hook = False


def shutdown(at_exit: bool = False) -> None:
    """Orderly end of the library: wait for the streams, then pgx_shutdown() -- communicator, streams, events and
    every pooled device / pinned block are released while the HIP runtime (and a profiler's tool library riding on
    it) is still fully alive.  Registered with atexit by the first ensure_init(): without it the process reached
    C++ static teardown with three streams, 64+ events and the pools still open, and under rocprofv3 that ended in
    a SIGSEGV inside __cxa_finalize after the results were written (gpurun_out/r2o_ss.log, round 2).  Buffers that
    Python still holds are harmless afterwards: pgx_free / pgx_host_free return quietly once the library is down.
    Safe to call twice; ensure_init() brings the library up again.

    A rank whose peer died mid-collective (or whose ranks fell out of step) holds collectives that never complete:
    nothing is synchronised before pgx_comm_quiesce has said, within its deadline, that the communicator is idle.
    If it is not, the communicator is abandoned; at interpreter exit the process then ends at once with status 70
    (os._exit: the HIP runtime's own teardown would wait for the stuck queues), otherwise RuntimeError."""
    global _initialised
    if not _initialised or _lib is None:
        return
    _initialised = False
    timeout = int(os.environ.get("PGX_COMM_EXIT_TIMEOUT_MS", "10000"))
    stuck = _lib.pgx_comm_quiesce(timeout) != 0
    if not stuck:
        try:
            _lib.pgx_stream_sync()
        finally:
            stuck = _lib.pgx_shutdown() != 0
    else:
        _lib.pgx_shutdown()                            # abandons the communicator, touches nothing else
    if stuck:
        msg = _lib.pgx_last_error().decode("utf-8", "replace")
        if at_exit:
            import sys
            try:
                sys.stderr.write(f"pygmu2_amd: {msg}; exiting with status 70\n")
                sys.stdout.flush()
                sys.stderr.flush()
            finally:
                os._exit(70)
        raise PgxError(msg)


def _shutdown_at_exit() -> None:
    shutdown(at_exit=True)


def device_available() -> bool:
    try:
        lib = load_library()
    except (RuntimeError, OSError, AttributeError):
        return False
    n = C.c_int(0)
    return lib.pgx_device_count(C.byref(n)) == 0 and n.value > 0


def synchronize() -> None:
    check(ensure_init().pgx_stream_sync(), "pgx_stream_sync")


def device_name() -> str:
    lib = ensure_init()
    buf = C.create_string_buffer(256)
    check(lib.pgx_device_name(buf, 256))
    return buf.value.decode()


_F32 = np.dtype(np.float32)


class DeviceBuffer:
    """
    A typed, shaped block of device memory owned by this Python object (returned to the
    library's pool when the object dies).  Exposes __cuda_array_interface__ so that
    torch.as_tensor(buf) wraps it without a copy (used for the RCCL reduction).
    """

    __slots__ = ("ptr", "shape", "dtype", "nbytes", "_owner", "__weakref__")

    def __init__(self, shape, dtype=np.float32, *, zero: bool = False):
        lib = _lib if _initialised else ensure_init()
        if type(shape) is tuple:
            self.shape = shape if all(type(v) is int for v in shape) else tuple(int(v) for v in shape)
        else:
            self.shape = tuple(int(v) for v in shape) if isinstance(shape, list) else (int(shape),)
        self.dtype = _F32 if dtype is np.float32 else np.dtype(dtype)
        n = 1
        for v in self.shape:
            n *= v
        self.nbytes = n * self.dtype.itemsize
        p = C.c_void_p(0)
        rc = lib.pgx_malloc(C.byref(p), self.nbytes or 1)
        if rc:
            check(rc, "pgx_malloc")
        self.ptr = p.value
        self._owner = True
        if zero and self.nbytes:
            check(lib.pgx_memset(self.ptr, 0, self.nbytes), "pgx_memset")

    @classmethod
    def from_host(cls, array) -> "DeviceBuffer":
        a = np.ascontiguousarray(array)
        buf = cls(a.shape, a.dtype)
        if a.nbytes:
            check(ensure_init().pgx_memcpy_h2d(buf.ptr, a.ctypes.data, a.nbytes), "pgx_memcpy_h2d")
        return buf

    def upload(self, array) -> None:
        a = np.ascontiguousarray(array, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError(f"upload size mismatch: {a.nbytes} != {self.nbytes}")
        if a.nbytes:
            check(ensure_init().pgx_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes), "pgx_memcpy_h2d")

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(ensure_init().pgx_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes), "pgx_memcpy_d2h")
        return out

    def begin_to_host(self):
        """Start an asynchronous copy into a pinned host block: (numpy view of the block, ticket).  The view
        holds valid data once `wait_to_host(ticket)` has returned; the block returns to the library's pinned
        pool when the last view of it dies."""
        lib = ensure_init()
        block = PinnedBlock(self.nbytes, self.shape, self.dtype)
        ticket = C.c_int64(0)
        check(lib.pgx_d2h_begin(block.ptr, self.ptr, self.nbytes, C.byref(ticket)), "pgx_d2h_begin")
        return np.asarray(block), ticket.value

    def zero_(self) -> None:
        if self.nbytes:
            check(ensure_init().pgx_memset(self.ptr, 0, self.nbytes), "pgx_memset")

    def offset_ptr(self, n_elements: int) -> int:
        return self.ptr + int(n_elements) * self.dtype.itemsize

    def rows(self, first: int, count: int) -> "DeviceBuffer":
        """Non-owning view of `count` leading-dimension rows starting at `first` (keeps self alive)."""
        view = object.__new__(DeviceBuffer)
        row_elems = 1
        for s in self.shape[1:]:
            row_elems *= s
        view.shape = (int(count),) + tuple(self.shape[1:])
        view.dtype = self.dtype
        view.nbytes = int(count) * row_elems * self.dtype.itemsize
        view.ptr = self.ptr + int(first) * row_elems * self.dtype.itemsize
        view._owner = self          # a DeviceBuffer here means "borrowed": never freed by the view
        return view

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False),
                "version": 2, "strides": None}

    def __del__(self):
        try:
            if getattr(self, "_owner", False) is True and self.ptr and _lib is not None:
                _lib.pgx_free(self.ptr)
        except Exception:
            pass
        self.ptr = 0

    def __repr__(self):
        return f"DeviceBuffer(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr or 0:x})"


class PinnedBlock:
    """A pinned (page-locked) host block from the library's pool, exposed to numpy through
    __array_interface__ (numpy keeps this object alive as the array's base)."""

    __slots__ = ("ptr", "nbytes", "__array_interface__")

    def __init__(self, nbytes: int, shape, dtype):
        p = C.c_void_p(0)
        check(ensure_init().pgx_host_malloc(C.byref(p), max(int(nbytes), 1)), "pgx_host_malloc")
        self.ptr = p.value
        self.nbytes = int(nbytes)
        self.__array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str,
                                    "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.pgx_host_free(self.ptr)
        except Exception:
            pass
        self.ptr = 0


def host_copy_done(ticket: int) -> bool:
    """Has the asynchronous copy of `ticket` (DeviceBuffer.begin_to_host) landed?  Never blocks."""
    done = C.c_int(0)
    check(_lib.pgx_d2h_query(ticket, C.byref(done)), "pgx_d2h_query")
    return bool(done.value)


def wait_to_host(ticket: int) -> None:
    check(_lib.pgx_d2h_wait(ticket), "pgx_d2h_wait")


def fence_to_host(ticket: int) -> None:
    check(_lib.pgx_d2h_fence(ticket), "pgx_d2h_fence")


def upload_struct(dtype: np.dtype, **fields) -> DeviceBuffer:
    """One parameter block (a C struct from pygmu_hip.h) -> device memory."""
    rec = np.zeros(1, dtype=dtype)
    for k, v in fields.items():
        rec[k] = v
    return DeviceBuffer.from_host(rec)


def upload_structs(records: np.ndarray) -> DeviceBuffer:
    return DeviceBuffer.from_host(np.ascontiguousarray(records))


class Event:
    """HIP event on the library stream (used by bench.py for kernel timing)."""

    def __init__(self):
        p = C.c_void_p(0)
        check(ensure_init().pgx_event_create(C.byref(p)), "pgx_event_create")
        self.ptr = p.value

    def record(self):
        check(_lib.pgx_event_record(self.ptr), "pgx_event_record")

    def elapsed_ms_since(self, start: "Event") -> float:
        ms = C.c_float(0)
        check(_lib.pgx_event_elapsed_ms(start.ptr, self.ptr, C.byref(ms)), "pgx_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.pgx_event_destroy(self.ptr)
        except Exception:
            pass
