"""The device status word (include/kapre_hip.h: kpr_device_status / KPR_E_DEVICE).

Kernels whose waves hand work to each other through LDS flags bound every wait; a wait that runs out raises a bit in mapped host
memory, the next forward call fails, kpr_device_status reads and clears.  The chain is driven end to end by a self-test kernel
whose wait cannot end (kpr_debug_spin_timeout)."""
import ctypes

import numpy as np
import pytest

from kapre_amd import _ffi

SELF_TEST = 1 << 31
KPR_E_DEVICE = -5


def test_status_api_is_exported_and_documented():
    import os
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "kapre_hip.h")).read()
    assert "KPR_E_DEVICE = -5" in header
    assert "int kpr_device_status(unsigned* flags_out);" in header
    lib = _ffi.lib()
    assert hasattr(lib, "kpr_device_status") and hasattr(lib, "kpr_debug_spin_timeout")


def test_status_is_zero_without_a_device_call():
    # (no kernel has been launched by this process in a CPU run; on the GPU box earlier tests leave it clean)
    flags = ctypes.c_uint(123)
    assert _ffi.lib().kpr_device_status(ctypes.byref(flags)) == 0
    assert flags.value == 0
    assert _ffi.lib().kpr_device_status(None) == 0


def test_entry_check_runs_without_a_device():
    # every forward entry point passes the status check first; an empty call returns before any HIP call, so this runs on CPU
    lib = _ffi.lib()
    assert lib.kpr_abs_c64(None, 0, None, None) == 0
    assert lib.kpr_angle_c64(None, 0, None, None) == 0
    assert lib.kpr_last_launches() == b""


@pytest.mark.gpu
def test_healthy_launches_leave_the_word_clean():
    import torch
    import kapre_amd as kapre
    from kapre_amd.composed import get_melspectrogram_layer, get_perfectly_reconstructing_stft_istft

    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.uniform(-1, 1, (64, 44100, 1)).astype(np.float32)).cuda()
    get_melspectrogram_layer(input_shape=(44100, 1), n_fft=2048, hop_length=512, sample_rate=44100)(x)
    st, ist = get_perfectly_reconstructing_stft_istft(n_fft=1024, hop_length=256, waveform_data_format='default',
                                                      stft_data_format='default')
    ist(st(x))                                               # k_istft_pw: the kernel with run-to-run hand-overs
    _ffi.set_option("istft_path", 3)                         # the ring kernel
    try:
        ist(st(x))
    finally:
        _ffi.set_option("istft_path", 0)
    assert _ffi.device_status() == 0


@pytest.mark.gpu
def test_a_wait_that_runs_out_fails_the_next_call_until_the_status_is_read():
    import torch
    from kapre_amd.time_frequency import Magnitude

    assert _ffi.device_status() == 0
    z = torch.complex(torch.ones(8, 4, 5, 1), torch.zeros(8, 4, 5, 1)).cuda()
    mag = Magnitude()
    assert float(mag(z).sum()) == 160.0
    _ffi.check(_ffi.lib().kpr_debug_spin_timeout(_ffi.current_stream_ptr()), "kpr_debug_spin_timeout")
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match=r"code -5.*bounded wait.*self-test"):
        mag(z)                                               # sticky: every forward entry point checks the word on entry
    with pytest.raises(RuntimeError, match=r"code -5"):
        mag(z)
    flags = ctypes.c_uint(0)
    assert _ffi.lib().kpr_device_status(ctypes.byref(flags)) == KPR_E_DEVICE
    assert flags.value == SELF_TEST
    assert b"self-test" in _ffi.lib().kpr_last_error()
    assert _ffi.device_status() == 0                         # read = cleared
    assert float(mag(z).sum()) == 160.0


@pytest.mark.gpu
def test_python_helper_raises():
    import torch

    _ffi.check(_ffi.lib().kpr_debug_spin_timeout(_ffi.current_stream_ptr()), "kpr_debug_spin_timeout")
    with pytest.raises(RuntimeError, match="bounded wait"):
        _ffi.device_status()                                 # synchronises, reads, raises
    assert _ffi.device_status() == 0
