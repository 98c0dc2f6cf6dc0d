"""ctypes binding of libkapre_hip.so (C ABI declared in include/kapre_hip.h).

PyTorch is only the device-memory container: tensors are passed as raw device pointers together
with the raw hipStream_t of torch's current stream.  There is NO CPU fallback: if the shared
library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KAPRE_AMD_LIB: development override to A/B alternative builds of the same source
LIB_PATH = os.environ.get("KAPRE_AMD_LIB") or os.path.join(_HERE, "lib", "libkapre_hip.so")

CHANNELS_FIRST, CHANNELS_LAST = 0, 1
OUT_COMPLEX, OUT_MAGNITUDE, OUT_PHASE = 0, 1, 2


class StftGeom(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int64), ("channels", ctypes.c_int32), ("time", ctypes.c_int64),
                ("n_fft", ctypes.c_int32), ("win_length", ctypes.c_int32),
                ("hop_length", ctypes.c_int32), ("pad_begin", ctypes.c_int32),
                ("pad_end", ctypes.c_int32), ("in_layout", ctypes.c_int32),
                ("out_layout", ctypes.c_int32)]


class DbParams(ctypes.Structure):
    _fields_ = [("enabled", ctypes.c_int32), ("ref_value", ctypes.c_float),
                ("amin", ctypes.c_float), ("dynamic_range", ctypes.c_float)]


EXPORTS = {
    # name: (restype, argtypes)
    "kpr_version": (ctypes.c_int, []),
    "kpr_last_error": (ctypes.c_char_p, []),
    "kpr_last_launches": (ctypes.c_char_p, []),
    "kpr_fft_fast_path": (ctypes.c_int, [ctypes.c_int]),
    "kpr_fft_plan": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "kpr_num_frames": (ctypes.c_int64, [ctypes.POINTER(StftGeom)]),
    "kpr_stft_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int]),
    "kpr_stft_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_void_p]),
    "kpr_mel_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int,
                                                 ctypes.POINTER(DbParams)]),
    "kpr_mel_workspace_bytes_unpacked": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int]),
    "kpr_mel_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.POINTER(DbParams), ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int64, ctypes.c_void_p]),
    "kpr_filterbank_forget": (ctypes.c_int, [ctypes.c_void_p]),
    "kpr_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "kpr_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "kpr_debug_stamps": (ctypes.c_int, [ctypes.c_void_p]),
    "kpr_debug_sclk_mhz": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float)]),
    "kpr_debug_calib_read8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "kpr_filterbank_pack_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "kpr_filterbank_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_filterbank_kranges": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p]),
    "kpr_abs_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_void_p]),
    "kpr_angle_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                     ctypes.c_void_p]),
    "kpr_apply_filterbank_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_apply_filterbank_packed_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                       ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_db_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "kpr_mag_to_db_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.POINTER(DbParams), ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_void_p]),
    # float64 / complex128 variants (layers built with dtype='float64')
    "kpr_stft_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "kpr_istft_f64_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int64]),
    "kpr_istft_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p]),
    "kpr_abs_c128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_angle_c128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_apply_filterbank_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "kpr_mag_to_db_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    # backward passes (kapre_amd/autograd.py)
    "kpr_abs_c64_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "kpr_angle_c64_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "kpr_abs_c128_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                        ctypes.c_void_p]),
    "kpr_angle_c128_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "kpr_spec_edge_scale_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "kpr_spec_edge_scale_c128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "kpr_mag_to_db_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.POINTER(DbParams), ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_mag_to_db_bwd_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "kpr_frame_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_energy_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_delta_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_device_status": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint)]),
    "kpr_debug_spin_timeout": (ctypes.c_int, [ctypes.c_void_p]),
    "kpr_istft_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int64]),
    "kpr_istft_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_void_p]),
    "kpr_frame_count": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "kpr_frame_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_energy_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_delta_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p]),
}

PAD_MODES = {"constant": 0, "symmetric": 1, "reflect": 2}

_lib = None
_lock = threading.Lock()


def lib():
    """The loaded shared library; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "kapre_amd: %s is missing - build it with `python -m kapre_amd.build` "
                        "(or __graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
                # torch ships its own libamdhip64: import it FIRST so that libkapre_hip.so's dependency resolves to
                # that already-loaded copy -- loaded the other way round the process holds two HIP runtimes and
                # torch's device pointers mean nothing to ours ("no ROCm-capable device is detected")
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in EXPORTS.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().kpr_last_error().decode("utf-8", "replace")
        raise RuntimeError("kapre_amd: %s failed (code %d): %s" % (what, rc, msg))


def set_option(name: str, value: int) -> int:
    """kpr_set_option; returns the previous value (so tests can restore it)."""
    old = ctypes.c_int(0)
    check(lib().kpr_get_option(name.encode(), ctypes.byref(old)), "kpr_get_option")
    check(lib().kpr_set_option(name.encode(), int(value)), "kpr_set_option")
    return old.value


def last_launches() -> str:
    """Kernel names the calling thread's most recent hot-path call launched (kpr_last_launches)."""
    return lib().kpr_last_launches().decode("utf-8", "replace")


def device_status(synchronize: bool = True, raise_on_error: bool = True) -> int:
    """kpr_device_status: the bits raised by kernels that gave up a bounded wait since the last call (0 = healthy), cleared by
    reading.  ``synchronize`` waits for the device first (the word is only final for launches that have finished).  With
    ``raise_on_error`` a non-zero word raises RuntimeError (the outputs of the affected launches are wrong)."""
    if synchronize:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    flags = ctypes.c_uint(0)
    rc = lib().kpr_device_status(ctypes.byref(flags))
    if rc != 0 and raise_on_error:
        check(rc, "kpr_device_status")
    return int(flags.value)


def sclk_mhz() -> float:
    """Shader clock in MHz under a dense packed-f32 vector load (kpr_debug_sclk_mhz; blocking, ~0.3 ms of GPU time)."""
    out = ctypes.c_float(0.0)
    check(lib().kpr_debug_sclk_mhz(ctypes.byref(out)), "kpr_debug_sclk_mhz")
    return float(out.value)


PACK_HEADER_FLOATS = 64

# kpr_fft_plan codes (include/kapre_hip.h)
FFT_DFT_GEMM, FFT_POW2, FFT_MIXED_RADIX, FFT_TWO_PASS, FFT_BLUESTEIN, FFT_SUB_FFT, FFT_GENERIC = range(7)


def layout(data_format: str) -> int:
    return CHANNELS_LAST if data_format == "channels_last" else CHANNELS_FIRST


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("kapre_amd: no HIP device visible (torch.cuda.is_available() is "
                           "False); the MI355X kernels have no CPU fallback")


def as_device_f32(x, device=None):
    """numpy / torch (any device, any float dtype) -> contiguous float32 torch tensor on the GPU."""
    import torch
    if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32
            and x.is_contiguous()):
        return x                                   # steady-state fast path
    require_gpu()
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        x = x.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    return x.contiguous()


def as_device_dtype(x, dtype, device=None):
    """numpy / torch -> contiguous torch tensor of ``dtype`` on the GPU (float64 / complex128 path)."""
    import torch
    require_gpu()
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        x = x.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


def is_f64(x) -> bool:
    """True for float64 / complex128 numpy arrays and torch tensors."""
    import torch
    if isinstance(x, np.ndarray):
        return x.dtype in (np.float64, np.complex128)
    return isinstance(x, torch.Tensor) and x.dtype in (torch.float64, torch.complex128)


def as_device_c64(x, device=None):
    import torch
    require_gpu()
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    if x.dtype != torch.complex64:
        x = x.to(torch.complex64)
    return x.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def filterbank_pack(fb_host: np.ndarray, kranges: np.ndarray) -> np.ndarray:
    """Host copy of the filterbank in the MFMA-fragment order the fused kernel streams."""
    fb_host = np.ascontiguousarray(fb_host, dtype=np.float32)
    n_freq, n_filt = fb_host.shape
    kr = kranges.ctypes.data_as(ctypes.c_void_p) if kranges is not None else ctypes.c_void_p(0)
    n = int(lib().kpr_filterbank_pack_floats(n_freq, n_filt, kr))
    if n < 0:
        check(-1, "kpr_filterbank_pack_floats")
    out = np.zeros(n, dtype=np.float32)
    check(lib().kpr_filterbank_pack(fb_host.ctypes.data_as(ctypes.c_void_p), n_freq, n_filt, kr,
                                    out.ctypes.data_as(ctypes.c_void_p)), "kpr_filterbank_pack")
    return out


def filterbank_kranges(fb_host: np.ndarray) -> np.ndarray:
    """Per 16-filter tile [lo, hi) row range outside of which the (n_freq, n_filt) matrix is 0."""
    fb_host = np.ascontiguousarray(fb_host, dtype=np.float32)
    n_freq, n_filt = fb_host.shape
    out = np.zeros(2 * ((n_filt + 15) // 16), dtype=np.int32)
    check(lib().kpr_filterbank_kranges(fb_host.ctypes.data_as(ctypes.c_void_p), n_freq, n_filt,
                                       out.ctypes.data_as(ctypes.c_void_p)),
          "kpr_filterbank_kranges")
    return out
