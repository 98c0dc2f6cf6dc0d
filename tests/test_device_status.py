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
    with pytest.raises(RuntimeError, match=r"code -5.*device status word.*self-test"):
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
    with pytest.raises(RuntimeError, match="bounded wait ran out"):
        _ffi.device_status()                                 # synchronises, reads, raises
    assert _ffi.device_status() == 0


@pytest.mark.gpu
def test_a_blob_replaced_under_a_cached_band_plan_is_reported_not_trusted():
    """the band plan of k_mel_pw is cached per device address.  Another packed filterbank written to the SAME address
    (same shape, same k-ranges, no kpr_filterbank_forget) used to be read through the old plan's table offsets, silently.  Now the
    kernel compares the header on the device with the plan it was launched for: it computes nothing, the next call fails, and
    after kpr_device_status the call goes through with the blob that is actually there."""
    import torch
    import kapre_oracle as o
    from kapre_amd import composed

    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (32, 44100, 1)).astype(np.float32)
    kw = dict(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128)
    model = composed.get_melspectrogram_layer(**kw)
    xd = torch.from_numpy(x).cuda()
    first = model(xd).cpu().numpy()
    assert "k_mel_pw" in _ffi.last_launches()
    want_a = o.kapre_melspectrogram(x, **kw)
    assert np.abs(first - want_a).max() <= 1e-4 * np.abs(want_a).max()

    fb_layer = [l for l in model.layers if type(l).__name__ == "ApplyFilterbank"][0]
    packed = fb_layer._fb_packed_device(xd.device)
    kr = fb_layer._fb_kranges()
    fb_a = np.asarray(fb_layer.filterbank, np.float32)
    # same shape, same k-ranges, but three non-zeros per bin: no band plan (k_mel_pw cannot run it)
    fb_b = fb_a.copy()
    nz = fb_a > 0
    fb_b[:, 2:] += 0.25 * fb_a[:, :-2] * nz[:, 1:-1]        # filter m + 2 also sees the bins filters m, m + 1 share ...
    fb_b *= (np.arange(fb_b.shape[1]) // 16 == (np.argmax(nz, axis=1) // 16)[:, None]) | nz   # ... inside the rows its tile covers
    assert np.array_equal(_ffi.filterbank_kranges(fb_b), kr)
    blob_b = _ffi.filterbank_pack(fb_b, kr)
    assert blob_b.size == packed.numel()
    hdr = blob_b[:12].view(np.uint32)
    assert hdr[6] == 0, "the replacement must not carry a band plan"
    packed.copy_(torch.from_numpy(blob_b).to(packed.dtype))
    torch.cuda.synchronize()

    model(xd)                                                # launched with the cached plan; the kernel refuses
    assert "k_mel_pw" in _ffi.last_launches()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match=r"code -5.*kpr_filterbank_forget"):
        model(xd)
    assert _ffi.device_status(raise_on_error=False) == 1 << 4
    got = model(xd).cpu().numpy()                            # re-verified: the MFMA kernel on the blob that IS there
    assert "k_mel_pw" not in _ffi.last_launches()
    mag = np.abs(o.kapre_stft(x, 2048, None, 512))
    want_b = o.apply_filterbank(mag, fb_b, "channels_last")
    assert np.abs(got - want_b).max() <= 1e-4 * np.abs(want_b).max()
    assert _ffi.device_status() == 0


@pytest.mark.gpu
def test_check_device_is_the_post_call_check_of_torch_tensor_callers():
    """a launch that gave up a bounded wait (or refused a stale plan) has long returned 0; a caller that hands
    torch tensors to the layers and never makes a second call learns of it through kapre_amd.check_device() -- what
    Sequential.predict does after its copy."""
    import torch
    import kapre_amd

    x = torch.rand((4, 8000, 1), device="cuda")
    kapre_amd.get_melspectrogram_layer(n_fft=512, hop_length=128, sample_rate=16000, n_mels=40)(x)
    assert kapre_amd.check_device() is None                  # healthy: nothing raised
    _ffi.check(_ffi.lib().kpr_debug_spin_timeout(_ffi.current_stream_ptr()), "kpr_debug_spin_timeout")
    with pytest.raises(RuntimeError, match="bounded wait ran out"):
        kapre_amd.check_device(x.device)
    assert kapre_amd.check_device() is None                  # read and cleared
