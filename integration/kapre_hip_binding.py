"""The whole binding a Kapre maintainer would add for the STFT entry point and for the fused melspectrogram chain
(INTEGRATION.md sections 2 and 3) -- framework
neutral: tensors cross as DLPack capsules (`tf.experimental.dlpack.to_dlpack(t)`, `torch.utils.dlpack.to_dlpack(t)`,
`cupy.ndarray.toDlpack()`), the library sees raw device pointers.  Nothing here imports kapre_amd.

tests/test_integration_stub.py checks this file against include/kapre_hip.h (struct fields, argument counts and
kinds of every prototype it binds) and checks the DLPack pointer extraction with real capsules."""
import ctypes

KPR_OUT_COMPLEX, KPR_OUT_MAGNITUDE, KPR_OUT_PHASE = 0, 1, 2
KPR_CHANNELS_FIRST, KPR_CHANNELS_LAST = 0, 1


class StftGeom(ctypes.Structure):                       # include/kapre_hip.h: kpr_stft_geom
    _fields_ = [("batch", ctypes.c_int64), ("channels", ctypes.c_int32), ("time", ctypes.c_int64),
                ("n_fft", ctypes.c_int32), ("win_length", ctypes.c_int32),
                ("hop_length", ctypes.c_int32), ("pad_begin", ctypes.c_int32),
                ("pad_end", ctypes.c_int32), ("in_layout", ctypes.c_int32),
                ("out_layout", ctypes.c_int32)]


class DbParams(ctypes.Structure):                       # kpr_db_params
    _fields_ = [("enabled", ctypes.c_int32), ("ref_value", ctypes.c_float),
                ("amin", ctypes.c_float), ("dynamic_range", ctypes.c_float)]


# name: (restype, argtypes) -- exactly the prototypes of include/kapre_hip.h
PROTOTYPES = {
    "kpr_last_error": (ctypes.c_char_p, []),
    "kpr_device_status": (ctypes.c_int, [ctypes.c_void_p]),
    "kpr_num_frames": (ctypes.c_int64, [ctypes.POINTER(StftGeom)]),
    "kpr_stft_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int]),
    "kpr_stft_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "kpr_istft_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int64]),
    "kpr_istft_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_int64, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "kpr_mag_to_db_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(DbParams),
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    # fused STFT -> Magnitude -> ApplyFilterbank [-> MagnitudeToDecibel] (composed.get_melspectrogram_layer)
    "kpr_filterbank_kranges": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "kpr_filterbank_pack_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "kpr_filterbank_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p]),
    "kpr_mel_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int, ctypes.POINTER(DbParams)]),
    "kpr_mel_workspace_bytes_unpacked": (ctypes.c_int64, [ctypes.POINTER(StftGeom), ctypes.c_int]),
    "kpr_mel_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StftGeom), ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(DbParams),
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    # backward passes (tf.custom_gradient around the calls above; INTEGRATION.md, "Gradients"): the linear layers use the
    # forward entry points as adjoints (kpr_spec_edge_scale_c64 + kpr_istft_f32 / kpr_stft_f32), the others these
    "kpr_spec_edge_scale_c64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_abs_c64_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "kpr_angle_c64_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "kpr_mag_to_db_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.POINTER(DbParams), ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_frame_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_energy_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "kpr_delta_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
}

_lib = None


MIN_VERSION = 101        # kpr_version() of the first library with every entry point bound below (KPR_VERSION, kapre_hip.h)


def load(path="libkapre_hip.so"):
    global _lib
    lib = ctypes.CDLL(path)
    lib.kpr_version.restype, lib.kpr_version.argtypes = ctypes.c_int, []
    have = int(lib.kpr_version())
    if have < MIN_VERSION:                       # an older build lacks the backward / kpr_filterbank_forget / kpr_last_launches exports
        raise RuntimeError("libkapre_hip.so reports version %d, this binding needs >= %d" % (have, MIN_VERSION))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(rc):
    if rc:
        raise RuntimeError(_lib.kpr_last_error().decode())


def device_status() -> int:
    """Bits raised by kernels that gave up a bounded wait since the last call (0 = healthy; include/kapre_hip.h).  Call it after
    the stream has been waited for -- e.g. at the end of a tf.py_function / once per batch; a raised word also fails every later
    forward call with KPR_E_DEVICE."""
    flags = ctypes.c_uint(0)
    _lib.kpr_device_status(ctypes.byref(flags))
    return flags.value


class _DLTensor(ctypes.Structure):                      # dlpack.h: DLTensor (the head of DLManagedTensor)
    _fields_ = [("data", ctypes.c_void_p), ("device_type", ctypes.c_int32), ("device_id", ctypes.c_int32),
                ("ndim", ctypes.c_int32), ("dtype_code", ctypes.c_uint8), ("dtype_bits", ctypes.c_uint8),
                ("dtype_lanes", ctypes.c_uint16), ("shape", ctypes.POINTER(ctypes.c_int64)),
                ("strides", ctypes.POINTER(ctypes.c_int64)), ("byte_offset", ctypes.c_uint64)]


def dlpack_data_ptr(capsule) -> int:
    """DLPack capsule ("dltensor") -> address of element 0 (zero copy; the capsule stays owned by the caller)."""
    get = ctypes.pythonapi.PyCapsule_GetPointer
    get.restype, get.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
    t = _DLTensor.from_address(get(capsule, b"dltensor"))
    return (t.data or 0) + t.byte_offset


def stft(x_capsule, x_shape, layer, out_capsule, window_capsule, stream=0, mode=KPR_OUT_COMPLEX):
    """Replaces tf.transpose + tf.pad + tf.signal.stft + tf.transpose of STFT.call (kapre/time_frequency.py:164-185).
    `layer` carries Kapre's STFT attributes; x_shape is (batch, time, ch) or (batch, ch, time)."""
    last = layer.input_data_format == "channels_last"
    b, t, c = (x_shape[0], x_shape[1], x_shape[2]) if last else (x_shape[0], x_shape[2], x_shape[1])
    g = StftGeom(b, c, t, layer.n_fft, layer.win_length, layer.hop_length, int(bool(layer.pad_begin)),
                 int(bool(layer.pad_end)), KPR_CHANNELS_LAST if last else KPR_CHANNELS_FIRST,
                 KPR_CHANNELS_LAST if layer.output_data_format == "channels_last" else KPR_CHANNELS_FIRST)
    ws_bytes = _lib.kpr_stft_workspace_bytes(ctypes.byref(g), mode)
    if ws_bytes != 0:          # only the DFT-as-GEMM sizes with a real-valued epilogue; the caller allocates then
        raise NotImplementedError("this n_fft needs %d bytes of workspace" % ws_bytes)
    _check(_lib.kpr_stft_f32(dlpack_data_ptr(x_capsule), ctypes.byref(g), dlpack_data_ptr(window_capsule),
                             dlpack_data_ptr(out_capsule), mode, None, 0, stream))


def num_frames(x_shape, layer) -> int:
    last = layer.input_data_format == "channels_last"
    b, t, c = (x_shape[0], x_shape[1], x_shape[2]) if last else (x_shape[0], x_shape[2], x_shape[1])
    g = StftGeom(b, c, t, layer.n_fft, layer.win_length, layer.hop_length, int(bool(layer.pad_begin)),
                 int(bool(layer.pad_end)), int(last), int(layer.output_data_format == "channels_last"))
    return int(_lib.kpr_num_frames(ctypes.byref(g)))


def _geom(x_shape, layer):
    last = layer.input_data_format == "channels_last"
    b, t, c = (x_shape[0], x_shape[1], x_shape[2]) if last else (x_shape[0], x_shape[2], x_shape[1])
    return StftGeom(b, c, t, layer.n_fft, layer.win_length, layer.hop_length, int(bool(layer.pad_begin)),
                    int(bool(layer.pad_end)), int(last), int(layer.output_data_format == "channels_last"))


class MelFilterbank:
    """Host-side, once per filterbank matrix (the (n_freq, n_filt) float32 array backend.filterbank_mel /
    filterbank_log returns): the per-tile non-zero row ranges and the packed MFMA-fragment blob.  The caller uploads
    `packed` (and `fb` itself) to the device with its own framework and passes the capsules to melspectrogram()."""

    def __init__(self, fb):                                 # fb: C-contiguous float32 numpy array (n_freq, n_filt)
        import numpy as np
        self.fb = np.ascontiguousarray(fb, dtype=np.float32)
        self.n_freq, self.n_filt = self.fb.shape
        self.kranges = np.empty(2 * ((self.n_filt + 15) // 16), dtype=np.int32)
        _check(_lib.kpr_filterbank_kranges(self.fb.ctypes.data, self.n_freq, self.n_filt, self.kranges.ctypes.data))
        n = _lib.kpr_filterbank_pack_floats(self.n_freq, self.n_filt, self.kranges.ctypes.data)
        self.packed = None                                  # > 1024 filters: no packed form, dense product
        if n > 0:
            self.packed = np.empty(n, dtype=np.float32)
            _check(_lib.kpr_filterbank_pack(self.fb.ctypes.data, self.n_freq, self.n_filt, self.kranges.ctypes.data,
                                            self.packed.ctypes.data))


def mel_workspace_bytes(x_shape, layer, bank, db=None) -> int:
    g = _geom(x_shape, layer)
    if bank.packed is None:
        return int(_lib.kpr_mel_workspace_bytes_unpacked(ctypes.byref(g), bank.n_filt))
    return int(_lib.kpr_mel_workspace_bytes(ctypes.byref(g), bank.n_filt, ctypes.byref(db) if db else None))


def melspectrogram(x_capsule, x_shape, layer, bank, fb_capsule, packed_capsule, window_capsule, out_capsule,
                   workspace_capsule, workspace_bytes, db=None, stream=0):
    """Replaces the four layers get_melspectrogram_layer stacks (kapre/composed.py:138-261): STFT.call, Magnitude.call,
    ApplyFilterbank.call and -- with `db` (a DbParams with enabled = 1) -- MagnitudeToDecibel.call, in one launch.
    out: float32 (batch, frame, n_filt, ch) or (batch, ch, frame, n_filt) per layer.output_data_format."""
    g = _geom(x_shape, layer)
    _check(_lib.kpr_mel_f32(dlpack_data_ptr(x_capsule), ctypes.byref(g), dlpack_data_ptr(window_capsule),
                            dlpack_data_ptr(fb_capsule),
                            dlpack_data_ptr(packed_capsule) if packed_capsule is not None else None, bank.n_filt,
                            bank.kranges.ctypes.data, ctypes.byref(db) if db else None,
                            dlpack_data_ptr(out_capsule), dlpack_data_ptr(workspace_capsule), workspace_bytes, stream))
