"""ctypes binding of libsynthhip.so (include/synthhip.h).

There is no CPU fallback: every arithmetic entry point of this package goes through the HIP
library.  If the shared object is missing or no GPU is visible the calls raise -- loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SYNTHHIP_LIB", HERE / "libsynthhip.so"))

SH_OK = 0
SH_ERR_INVALID, SH_ERR_HIP, SH_ERR_NOMEM, SH_ERR_NOTINIT, SH_ERR_OVERFLOW, SH_ERR_RCCL, SH_ERR_LENGTH = -1, -2, -3, -4, -5, -6, -7
SH_SINE, SH_SAWTOOTH, SH_SQUARE, SH_PULSE, SH_HARMONICS, SH_TRIANGLE, SH_LINEAR, SH_NOISE, SH_BUFFER = range(9)
SH_EW_ADD, SH_EW_MUL, SH_EW_CLIP, SH_EW_ABS, SH_EW_COPY, SH_EW_FILL, SH_EW_AXPY, SH_EW_NEXTUP = range(8)
SH_OPT_QUANTISE_ROUND = 1
SH_INFO_LAST_MIXDOWN_FUSED = 2
SH_FM_NONE, SH_FM_SINE, SH_FM_BUFFER = range(3)
SH_DIST_ID_BYTES = 128


class SynthHipError(RuntimeError):
    def __init__(self, code: int, message: str) -> None:
        super().__init__("libsynthhip error %d: %s" % (code, message))
        self.code = code


class NativeLibraryMissing(ImportError):
    pass


# numpy structured dtypes that mirror the C structs byte for byte (align=True = C layout)
SEGMENT_DTYPE = np.dtype([("n0", "<u8"), ("t0", "<f8"), ("dt", "<f8")], align=True)
PARTIAL_DTYPE = np.dtype([("k", "<f8"), ("amp", "<f8")], align=True)
ENVELOPE_DTYPE = np.dtype([
    ("n_attack_end", "<u8"), ("n_decay_end", "<u8"), ("n_sustain_end", "<u8"), ("n_release_end", "<u8"),
    ("attack_slope", "<f8"), ("decay_slope", "<f8"), ("sustain_level", "<f8"), ("release_slope", "<f8"),
    ("tail_amp", "<f8"), ("enabled", "<i4"), ("has_tail", "<i4")], align=True)
VOICE_DTYPE = np.dtype([
    ("kind", "<i4"), ("fm_mode", "<i4"),
    ("amplitude", "<f8"), ("bias", "<f8"), ("pulsewidth", "<f8"),
    ("seg_offset", "<u4"), ("seg_count", "<u4"),
    ("harm_offset", "<u4"), ("harm_count", "<u4"), ("harm_dense", "<i4"), ("flip", "<i4"),
    ("frequency", "<f8"), ("fm_phase0", "<f8"), ("fm_inc", "<f8"),
    ("time_seg_offset", "<u4"), ("time_seg_count", "<u4"),
    ("lfo_a", "<f8"), ("lfo_d", "<f8"), ("lfo_amp", "<f8"), ("lfo_bias", "<f8"), ("lfo_K", "<f8"), ("lfo_C0", "<f8"),
    ("env", ENVELOPE_DTYPE),
    ("gain_l", "<f4"), ("gain_r", "<f4"),
    ("noise_seed", "<u8"), ("noise_hold", "<u4"), ("reserved0", "<u4"), ("start_frame", "<u8"),
    ("guard_t", "<f8"), ("guard_c", "<f8"), ("guard_offset", "<u4"), ("guard_count", "<u4")], align=True)

# sizes the C side must agree with (checked against the library's view in tests via sh_bank_create)
assert SEGMENT_DTYPE.itemsize == 24 and PARTIAL_DTYPE.itemsize == 16 and ENVELOPE_DTYPE.itemsize == 80
assert VOICE_DTYPE.itemsize == 272, VOICE_DTYPE.itemsize


class Counters(C.Structure):
    _fields_ = [("device_allocs", C.c_uint64), ("device_frees", C.c_uint64), ("stream_syncs", C.c_uint64), ("pool_hits", C.c_uint64),
                ("segmented_launches", C.c_uint64), ("tiled_launches", C.c_uint64), ("tiled_predicted", C.c_uint64)]


class DevInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("clock_mhz", C.c_int32), ("hbm_bytes", C.c_uint64), ("wavefront", C.c_int32), ("device", C.c_int32)]


# every symbol include/synthhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "sh_init": (C.c_int, [C.c_int]),
    "sh_shutdown": (C.c_int, []),
    "sh_is_initialized": (C.c_int, []),
    "sh_device_count": (C.c_int, []),
    "sh_device_info": (C.c_int, [C.POINTER(DevInfo)]),
    "sh_device_pci": (C.c_int, [C.c_char_p, C.c_int]),
    "sh_last_error": (C.c_char_p, []),
    "sh_version": (C.c_char_p, []),
    "sh_abi": (C.c_int, [C.POINTER(C.c_uint32), C.c_int]),
    "sh_sync": (C.c_int, []),
    "sh_set_option": (C.c_int, [C.c_int, C.c_int]),
    "sh_get_option": (C.c_int, [C.c_int]),
    "sh_debug_counters": (C.c_int, [C.POINTER(Counters)]),
    "sh_buf_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "sh_buf_free": (C.c_int, [_P]),
    "sh_buf_view": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    "sh_buf_size": (C.c_size_t, [_P]),
    "sh_buf_devptr": (_P, [_P]),
    "sh_buf_upload": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "sh_buf_download": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "sh_buf_fill_zero": (C.c_int, [_P, C.c_size_t, C.c_size_t]),
    "sh_buf_copy": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t]),
    "sh_timer_start": (C.c_int, []),
    "sh_timer_stop": (C.c_int, [C.POINTER(C.c_float)]),
    "sh_bank_create": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(_P)]),
    "sh_bank_destroy": (C.c_int, [_P]),
    "sh_bank_launch_stats": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sh_bank_nvoices": (C.c_uint32, [_P]),
    "sh_osc_render": (C.c_int, [_P, C.c_uint32, _P, _P, C.c_uint64, C.c_uint32, _P, _P, C.c_size_t, _P]),
    "sh_ew_f64": (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_double, C.c_double, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "sh_scan_f64": (C.c_int, [_P, C.c_uint32, C.c_double, _P, C.POINTER(C.c_double)]),
    "sh_bank_generate": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_size_t]),
    "sh_bank_render": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, _P]),
    "sh_bank_render_pcm": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_double, _P]),
    "sh_bank_render_run": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(_P), C.c_uint32, C.c_double]),
    "sh_bank_set_rows": (C.c_int, [_P, _P, _P]),
    "sh_bank_generate_f64": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_size_t, C.c_size_t]),
    "sh_scan_rows_f64": (C.c_int, [_P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t, _P]),
    "sh_bank_render_rows": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_size_t, _P, _P]),
    "sh_bank_generate_rows": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_size_t, _P, C.c_size_t]),
    "sh_bank_generate_i16": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_double, _P, C.c_size_t]),
    "sh_bank_generate_i16_async": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_double, _P, C.c_size_t]),
    "sh_overflow_check": (C.c_int, []),
    "sh_bank_mixdown_i16": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_double, _P]),
    "sh_bank_mixdown_i16_async": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_double, _P]),
    "sh_bank_generate_rows_i16": (C.c_int, [_P, C.c_uint64, C.c_uint32, _P, C.c_size_t, C.c_double, _P, C.c_size_t]),
    "sh_mix_bus_f32": (C.c_int, [_P, C.c_uint32, C.c_size_t, C.c_uint32, _P, _P]),
    "sh_mix_chain_i16": (C.c_int, [_P, C.c_uint32, C.c_size_t, C.c_uint32, _P]),
    "sh_mix_chain_pan_i16": (C.c_int, [_P, C.c_uint32, C.c_size_t, C.c_uint32, _P, _P]),
    "sh_mix_chain_gather_i16": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, _P, C.c_size_t]),
    "sh_mix_chain": (C.c_int, [_P, C.c_uint32, C.c_size_t, C.c_uint32, C.c_int, _P]),
    "sh_mix_chain_gather": (C.c_int, [C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, _P, C.c_size_t]),
    "sh_rt_create": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "sh_rt_destroy": (C.c_int, [_P]),
    "sh_rt_acquire": (C.c_int, [_P, _P]),
    "sh_rt_mix_turn": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "sh_quantize_f32": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_double, C.c_int, _P, C.c_size_t]),
    "sh_quantize_f64": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_double, C.c_int, _P, C.c_size_t]),
    "sh_quantize_clip_f32": (C.c_int, [_P, C.c_size_t, C.c_double, _P]),
    "sh_pcm_add": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_int, _P, C.c_size_t]),
    "sh_pcm_add_host": (C.c_int, [_P, _P, C.c_size_t, C.c_int, _P]),
    "sh_pcm_mul": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_int, C.c_double, _P, C.c_size_t]),
    "sh_pcm_fade": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, _P, C.c_size_t]),
    "sh_pcm_modulate": (C.c_int, [_P, C.c_size_t, C.c_int, _P, C.c_size_t, _P]),
    "sh_pcm_to_f64": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_double, _P]),
    "sh_pcm_pan_lfo": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, _P, _P]),
    "sh_pcm_bias": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, _P]),
    "sh_pcm_reverse": (C.c_int, [_P, C.c_size_t, C.c_int, _P]),
    "sh_pcm_tomono": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_double, C.c_double, _P]),
    "sh_pcm_tostereo": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_double, C.c_double, _P]),
    "sh_pcm_lin2lin": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, _P]),
    "sh_pcm_stats": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]),
    "sh_pcm_stats_stereo": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]),
    "sh_resample_out_frames": (C.c_size_t, [C.c_size_t, C.c_int, C.c_int]),
    "sh_resample": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_size_t)]),
    "sh_resample_span": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sh_resample_range": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, _P]),
    "sh_resample_host": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_size_t)]),
    "sh_dist_unique_id": (C.c_int, [_P]),
    "sh_dist_init": (C.c_int, [C.c_int, C.c_int, _P]),
    "sh_dist_shutdown": (C.c_int, []),
    "sh_dist_rank": (C.c_int, []),
    "sh_dist_world": (C.c_int, []),
    "sh_dist_comm_info": (C.c_int, [C.POINTER(C.c_int32), C.c_int]),
    "sh_dist_reduce_bus": (C.c_int, [_P, C.c_size_t, C.c_int]),
    "sh_dist_allreduce_bus": (C.c_int, [_P, C.c_size_t]),
    "sh_dist_barrier": (C.c_int, []),
    "sh_dist_slots": (C.c_int, []),
    "sh_dist_reduce_bus_async": (C.c_int, [_P, C.c_size_t, C.c_int, _P, C.c_int]),
    "sh_dist_wait_slot": (C.c_int, [C.c_int]),
    "sh_dist_mark_slot": (C.c_int, [C.c_int]),
    "sh_dist_reduce_bus_lagged": (C.c_int, [_P, C.c_size_t, C.c_int, _P, C.c_int]),
    "sh_dist_wait_slot_keep": (C.c_int, [C.c_int]),
    "sh_bus_finalize": (C.c_int, [_P, C.c_size_t, _P]),
}

_lib: Optional[C.CDLL] = None


SH_ABI_VERSION = 6           # include/synthhip.h; bumped with every change of a struct layout or of an entry point's meaning


class NativeLibraryStale(ImportError):
    pass


def _abi_expected() -> list:
    return [SH_ABI_VERSION, SEGMENT_DTYPE.itemsize, PARTIAL_DTYPE.itemsize, ENVELOPE_DTYPE.itemsize, VOICE_DTYPE.itemsize,
            C.sizeof(DevInfo), C.sizeof(Counters)]


def _check_abi(handle: C.CDLL, path) -> None:
    """Refuse a library whose view of the structs differs from this binding's: Python would pack records it misreads."""
    if not hasattr(handle, "sh_abi"):
        raise NativeLibraryStale("%s exports no sh_abi: it predates this binding (ABI %d); rebuild it with "
                                 "`python -m synthesizer_amd.build`" % (path, SH_ABI_VERSION))
    handle.sh_abi.restype = C.c_int
    handle.sh_abi.argtypes = [C.POINTER(C.c_uint32), C.c_int]
    got = (C.c_uint32 * 7)()
    n = handle.sh_abi(got, 7)
    want = _abi_expected()
    if n != 7 or list(got) != want:
        raise NativeLibraryStale("%s was built for another binary interface: (ABI, sizeof sh_segment, sh_partial, sh_envelope, sh_voice, "
                                 "sh_devinfo, sh_counters) = %s, this binding packs %s; rebuild it with `python -m synthesizer_amd.build`"
                                 % (path, list(got), want))


def lib() -> C.CDLL:
    """Load libsynthhip.so (CDLL: the GIL is released around every call).

    The in-tree library is rebuilt when its embedded source hash differs from the tree's.  A stale library is only ever loaded
    when the COMPILER is absent (a shipped .so on a box without hipcc) or SYNTHHIP_ALLOW_STALE=1 says so -- a tree that no longer
    compiles is an error, not a reason to run old code -- and in every case the library's binary interface (sh_abi: version and
    struct sizes) must be this binding's."""
    global _lib
    if _lib is None:
        if "SYNTHHIP_LIB" not in os.environ:
            from . import build as _build
            try:                                    # in-tree build when missing or stale (embedded source hash != the tree's)
                _build.build(verbose=False)
            except Exception as exc:
                if not LIB_PATH.exists():
                    raise NativeLibraryMissing(
                        "%s not found and building it failed (%s): run `python -m synthesizer_amd.build` "
                        "(hipcc, gfx950).  There is no CPU fallback." % (LIB_PATH, exc))
                no_compiler = isinstance(exc, FileNotFoundError) and not Path(_build.HIPCC).exists()
                if not (no_compiler or os.environ.get("SYNTHHIP_ALLOW_STALE") == "1"):
                    raise NativeLibraryStale(
                        "%s is stale (built from sources %s, the tree is %s) and rebuilding it FAILED (%s).  Fix the build, or set "
                        "SYNTHHIP_ALLOW_STALE=1 to load the old library knowingly." % (LIB_PATH, _build.built_hash() or "?", _build.source_hash(), exc))
                # a shipped library on a box without the compiler: use it, but say that it is not what the tree describes
                import warnings
                warnings.warn("%s is stale (built from sources %s, the tree is %s) and cannot be rebuilt here (%s): loading it as it is"
                              % (LIB_PATH, _build.built_hash() or "?", _build.source_hash(), exc), RuntimeWarning)
        if not LIB_PATH.exists():
            raise NativeLibraryMissing(
                "%s not found: build it with `python -m synthesizer_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(str(LIB_PATH))
        _check_abi(handle, LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if "SYNTHHIP_LIB" in os.environ and not hasattr(handle, name):
                continue                    # an older build named for an A/B timing (tools/): what it lacks cannot be called
            fn = getattr(handle, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def exported_symbols() -> list:
    return sorted(_SIGNATURES)


def check(rc: int) -> None:
    if rc != SH_OK:
        msg = lib().sh_last_error().decode("utf-8", "replace")
        if rc == SH_ERR_OVERFLOW:
            raise OverflowError(msg)
        if rc == SH_ERR_LENGTH:
            raise ValueError("Lengths should be the same (" + msg + ")")
        if rc == SH_ERR_INVALID:
            raise ValueError(msg)
        if rc == SH_ERR_NOMEM:
            raise MemoryError(msg)
        raise SynthHipError(rc, msg)


def set_quantise_round(on: bool) -> None:
    """params.variants["quantise"]: "round" -> the library's quantisers round half to even (sh_set_option); "trunc" -> int()'s rule."""
    check(lib().sh_set_option(SH_OPT_QUANTISE_ROUND, 1 if on else 0))


_initialized = False


def ensure_init(device: Optional[int] = None) -> None:
    """sh_init on first use.  Device: explicit argument, else SYNTHHIP_DEVICE, else LOCAL_RANK, else 0."""
    global _initialized
    if _initialized:
        return
    if device is None:
        device = int(os.environ.get("SYNTHHIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    L = lib()
    if L.sh_device_count() <= 0:
        raise SynthHipError(SH_ERR_NOTINIT, "no HIP device visible; this package has no CPU fallback")
    check(L.sh_init(device))
    _initialized = True


def shutdown() -> None:
    """sh_shutdown (which tears the RCCL state down first) and forget the initialisation, so that the next call
    re-initialises instead of failing with SH_ERR_NOTINIT."""
    global _initialized
    if _lib is not None:
        check(_lib.sh_shutdown())
    _initialized = False


def device_info() -> dict:
    ensure_init()
    info = DevInfo()
    check(lib().sh_device_info(C.byref(info)))
    return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
            "clock_mhz": info.clock_mhz, "hbm_bytes": info.hbm_bytes, "wavefront": info.wavefront,
            "device": info.device}


def device_pci() -> str:
    """PCI bus id of the GPU this process renders on (distinct per rank of a multi-GPU job: bench.py prints it per rank)."""
    ensure_init()
    buf = C.create_string_buffer(64)
    check(lib().sh_device_pci(buf, 64))
    return buf.value.decode()


class RtLane:
    """A real-time lane (sh_rt_*; include/synthhip.h): a stream, a lock and buffers of its own for a mixer that another thread drains
    while the library's one stream pair is busy.  ``mix_turn`` returns the chunk as bytes and takes the lane's lock only."""

    def __init__(self, max_chunk_bytes: int, max_sources: int = 1024) -> None:
        ensure_init()
        self._h = _P()
        self.max_chunk_bytes = int(max_chunk_bytes)
        check(lib().sh_rt_create(self.max_chunk_bytes, int(max_sources), C.byref(self._h)))
        self._out = (C.c_char * self.max_chunk_bytes)()

    def acquire(self, buf: "DeviceBuffer") -> None:
        """The lane's next turn waits (on the device) for everything the library has been given so far: what made ``buf``."""
        check(lib().sh_rt_acquire(self._h, buf.handle))

    def mix_turn(self, sources, nsamples: int, width: int) -> bytes:
        n = len(sources)
        bufs = (C.c_void_p * max(n, 1))(*[b.handle for b, _o, _n in sources])
        offs = (C.c_size_t * max(n, 1))(*[o for _b, o, _n in sources])
        lens = (C.c_uint32 * max(n, 1))(*[min(k, 0xFFFFFFFF) for _b, _o, k in sources])
        check(lib().sh_rt_mix_turn(self._h, bufs, offs, lens, n, nsamples, width, C.cast(self._out, C.c_void_p)))
        return self._out.raw[:nsamples * width]

    def close(self) -> None:
        if self._h:
            try:
                lib().sh_rt_destroy(self._h)
            finally:
                self._h = _P()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass


def debug_counters() -> dict:
    """Driver allocations / frees / self-inserted stream synchronisations / pool hits since sh_init (sh_debug_counters)."""
    c = Counters()
    check(lib().sh_debug_counters(C.byref(c)))
    return {name: getattr(c, name) for name, _ in Counters._fields_}


def sync() -> None:
    ensure_init()
    check(lib().sh_sync())


class DeviceBuffer:
    """RAII wrapper over sh_buf (HBM allocation owned by the library)."""

    def __init__(self, nbytes: int) -> None:
        ensure_init()
        self._h = _P()
        self.nbytes = int(nbytes)
        check(lib().sh_buf_alloc(self.nbytes, C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def view(self, offset: int, nbytes: int) -> "DeviceBuffer":
        """A non-owning window of this buffer (keeps the parent alive)."""
        v = DeviceBuffer.__new__(DeviceBuffer)
        v._h = _P()
        v.nbytes = int(nbytes)
        v._parent = self
        check(lib().sh_buf_view(self._h, offset, nbytes, C.byref(v._h)))
        return v

    @classmethod
    def from_array(cls, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        buf.upload(arr)
        return buf

    @classmethod
    def from_bytes(cls, data: bytes) -> "DeviceBuffer":
        buf = cls(len(data))
        if data:
            tmp = (C.c_char * len(data)).from_buffer_copy(data)
            check(lib().sh_buf_upload(buf._h, 0, C.addressof(tmp), len(data)))
        return buf

    def upload(self, arr: np.ndarray, offset: int = 0) -> None:
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            check(lib().sh_buf_upload(self._h, offset, arr.ctypes.data, arr.nbytes))

    def download(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            check(lib().sh_buf_download(self._h, offset, out.ctypes.data, out.nbytes))
        return out

    def download_bytes(self, nbytes: int, offset: int = 0) -> bytes:
        return self.download(np.uint8, nbytes, offset).tobytes()

    def zero(self, offset: int = 0, nbytes: Optional[int] = None) -> None:
        check(lib().sh_buf_fill_zero(self._h, offset, self.nbytes - offset if nbytes is None else nbytes))

    def free(self) -> None:
        if self._h:
            try:
                lib().sh_buf_free(self._h)
            finally:
                self._h = _P()

    def __del__(self) -> None:
        try:
            self.free()
        except Exception:
            pass


def _ptr(arr: Optional[np.ndarray]):
    return None if arr is None or arr.size == 0 else arr.ctypes.data


class Bank:
    """RAII wrapper over sh_bank."""

    def __init__(self, voices: np.ndarray, segs: np.ndarray, coefs: np.ndarray, partials: np.ndarray) -> None:
        ensure_init()
        assert voices.dtype == VOICE_DTYPE and segs.dtype == SEGMENT_DTYPE
        assert coefs.dtype == np.float64 and partials.dtype == PARTIAL_DTYPE
        self._keep = (np.ascontiguousarray(voices), np.ascontiguousarray(segs),
                      np.ascontiguousarray(coefs), np.ascontiguousarray(partials))
        v, s, c, p = self._keep
        self._h = _P()
        self.nvoices = len(v)
        check(lib().sh_bank_create(_ptr(v), len(v), _ptr(s), len(s), _ptr(c), len(c), _ptr(p), len(p), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def free(self) -> None:
        if self._h:
            try:
                lib().sh_bank_destroy(self._h)
            finally:
                self._h = _P()

    def __del__(self) -> None:
        try:
            self.free()
        except Exception:
            pass


def timer_start() -> None:
    check(lib().sh_timer_start())


def timer_stop() -> float:
    ms = C.c_float()
    check(lib().sh_timer_stop(C.byref(ms)))
    return float(ms.value)
