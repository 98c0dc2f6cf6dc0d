"""-m gpu: the layers are differentiable, like the TensorFlow graphs they replace (a Kapre front end sits inside
model.fit; /root/reference/kapre/time_frequency.py:146-187, :289-319, :351-359, :402, :535-548, backend.py:186-192).

Checker: the same layer arithmetic written with plain torch float64 ops on the CPU and differentiated by torch's own
autograd (`ref_*` below, test infrastructure only); the HIP backward passes (kapre_amd/autograd.py,
csrc/kpr_grad_kernels.h, and the forward kernels they reuse as adjoints) must give the same input gradient for the same
scalar loss <y, R> with a fixed random R.  Tolerance: 2e-4 of the largest gradient entry (float32), 1e-9 (float64).
"""
import numpy as np
import pytest
import torch

import kapre_amd as kapre
from kapre_amd import STFT, InverseSTFT, Magnitude, Phase, MagnitudeToDecibel, ApplyFilterbank, backend
from kapre_amd.composed import get_melspectrogram_layer, get_perfectly_reconstructing_stft_istft
from kapre_amd.keras_shim import Sequential
from kapre_amd.signal import Frame, Energy, LogmelToMFCC
from kapre_amd import Delta

pytestmark = pytest.mark.gpu

CL, CF = 'channels_last', 'channels_first'


# ---------------------------------------------------------------------------------------------
# the checker: torch float64 on the CPU
# ---------------------------------------------------------------------------------------------
def ref_stft(x_bct, n_fft, win, hop, window, pad_begin, pad_end):
    """tf.signal.stft as STFT.call drives it (time_frequency.py:164-182): (B, C, T) -> (B, C, F, K) complex128."""
    x = x_bct
    if pad_begin:
        x = torch.nn.functional.pad(x, (n_fft - hop, 0))
    t = x.shape[-1]
    if pad_end:
        n_frames = -(-t // hop)
        x = torch.nn.functional.pad(x, (0, max(0, (n_frames - 1) * hop + win - t)))
    frames = x.unfold(-1, win, hop) * torch.as_tensor(window, dtype=torch.float64)
    return torch.fft.rfft(frames, n=n_fft)


def ref_istft(spec_bcfk, n_fft, win, hop, synth):
    """tf.signal.inverse_stft (time_frequency.py:307-314): (B, C, F, K) -> (B, C, (F - 1) hop + win)."""
    y = torch.fft.irfft(spec_bcfk, n=n_fft)[..., :win] * torch.as_tensor(synth, dtype=torch.float64)
    b, c, f, _ = y.shape
    out = torch.zeros(b, c, (f - 1) * hop + win, dtype=torch.float64)
    for i in range(f):
        out[..., i * hop:i * hop + win] = out[..., i * hop:i * hop + win] + y[..., i, :]
    return out


def ref_db(x, ref_value, amin, dyn):
    """backend.magnitude_to_decibel (backend.py:186-192), items = batch entries."""
    log10 = lambda v: torch.log(v) / np.log(10.0)
    amin_t = torch.tensor(amin, dtype=x.dtype)
    l = 10.0 * log10(torch.maximum(x, amin_t)) - 10.0 * np.log10(max(amin, ref_value))
    m = l.reshape(l.shape[0], -1).amax(dim=1).reshape([-1] + [1] * (l.dim() - 1))
    return torch.maximum(l, m - dyn)


def to_bct(x, fmt):
    return x.permute(0, 2, 1) if fmt == CL else x


def spec_from_bcfk(s, fmt):
    return s.permute(0, 2, 3, 1) if fmt == CL else s


def spec_to_bcfk(s, fmt):
    return s.permute(0, 3, 1, 2) if fmt == CL else s


def loss_of(y, r):
    """<y, R> with R real; a complex y is viewed as (re, im) pairs."""
    if y.is_complex():
        y = torch.view_as_real(y)
    return (y * r.to(y.device, y.dtype)).sum()


def cotangent(shape, complex_, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(tuple(shape) + ((2,) if complex_ else ()), generator=g, dtype=torch.float64)


def check(got, want, tol, what):
    got = got.detach().cpu()
    if got.is_complex():
        got, want = torch.view_as_real(got.to(torch.complex128)), torch.view_as_real(want)
    got, want = got.to(torch.float64), want.to(torch.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(want.abs().max())
    assert scale > 0, what
    err = float((got - want).abs().max()) / scale
    assert err <= tol, '%s: max error %.3g of the largest gradient entry (limit %.1g)' % (what, err, tol)


def wave(batch, ch, t, fmt, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((batch, ch, t), generator=g, dtype=torch.float64) * 2 - 1
    x = x * torch.linspace(0.2, 1.0, batch, dtype=torch.float64).reshape(-1, 1, 1)      # items of different loudness
    x = (x.permute(0, 2, 1) if fmt == CL else x).contiguous()
    return x.to(dtype)


# ---------------------------------------------------------------------------------------------
# STFT
# ---------------------------------------------------------------------------------------------
STFT_CASES = [
    # n_fft, win, hop, pad_begin, pad_end, in_fmt, out_fmt, ch, T
    (512, None, 128, False, False, CL, CL, 1, 4000),
    (512, None, 128, True, True, CF, CL, 2, 4001),
    (1024, None, 256, True, False, CL, CF, 2, 6000),
    (2048, None, 512, False, True, CL, CL, 1, 9000),
    (2048, 1500, 512, True, True, CF, CF, 1, 7000),          # win_length < n_fft
    (400, 400, 160, False, False, CL, CL, 1, 5000),          # mixed-radix plan
    (256, 200, 64, True, True, CL, CF, 3, 3000),
    (1000, None, 250, False, False, CF, CF, 1, 6000),        # the reference's own test size
    (1200, None, 300, False, True, CL, CL, 2, 5000),         # size-generic engine
]


@pytest.mark.parametrize('n_fft,win,hop,pad_begin,pad_end,in_fmt,out_fmt,ch,t', STFT_CASES)
def test_stft_backward(n_fft, win, hop, pad_begin, pad_end, in_fmt, out_fmt, ch, t):
    layer = STFT(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=pad_begin, pad_end=pad_end,
                 input_data_format=in_fmt, output_data_format=out_fmt)
    win = win or n_fft
    x0 = wave(3, ch, t, in_fmt, seed=n_fft + hop)
    xg = x0.cuda().requires_grad_(True)
    y = layer(xg)
    assert y.grad_fn is not None and y.dtype == torch.complex64
    plain = layer(x0.cuda())
    assert plain.grad_fn is None and torch.equal(plain, y.detach())        # same launch either way
    r = cotangent(y.shape, True, seed=7)
    loss_of(y, r).backward()

    xr = x0.to(torch.float64).requires_grad_(True)
    window = backend.get_window_fn(None)(win).astype(np.float64)
    yr = spec_from_bcfk(ref_stft(to_bct(xr, in_fmt), n_fft, win, hop, window, pad_begin, pad_end), out_fmt)
    assert tuple(yr.shape) == tuple(y.shape)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 2e-4, 'dL/dx through STFT')


def test_stft_backward_float64():
    layer = STFT(n_fft=512, hop_length=128, pad_begin=True, dtype='float64')
    x0 = wave(2, 2, 3000, CL, seed=3, dtype=torch.float64)
    xg = x0.cuda().requires_grad_(True)
    y = Magnitude(dtype='float64')(layer(xg))
    assert y.dtype == torch.float64
    r = cotangent(y.shape, False, seed=8)
    loss_of(y, r).backward()
    xr = x0.clone().requires_grad_(True)
    window = backend.window_values(backend.get_window_fn(None), 512, np.float64)
    yr = spec_from_bcfk(ref_stft(to_bct(xr, CL), 512, 512, 128, window, True, False), CL).abs()
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 1e-9, 'float64 dL/dx through STFT + Magnitude')


def test_input_dtype_and_device_are_followed():
    """A float64 CPU leaf through float32 layers gets a float64 CPU gradient (the casts are on the tape)."""
    layer = STFT(n_fft=256, hop_length=64)
    x = wave(2, 1, 2000, CL, seed=5, dtype=torch.float64).requires_grad_(True)
    y = Magnitude()(layer(x))
    assert y.is_cuda and y.dtype == torch.float32
    y.sum().backward()
    assert x.grad is not None and x.grad.dtype == torch.float64 and not x.grad.is_cuda
    assert float(x.grad.abs().max()) > 0


# ---------------------------------------------------------------------------------------------
# InverseSTFT
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n_fft,win,hop,in_fmt,out_fmt,ch,frames', [
    (512, None, 128, CL, CL, 1, 30),
    (1024, None, 256, CF, CL, 2, 21),
    (2048, None, 512, CL, CF, 1, 12),
    (400, 400, 100, CF, CF, 2, 25),
    (256, 200, 50, CL, CL, 3, 33),
    (1000, None, 250, CL, CL, 1, 14),
])
def test_istft_backward(n_fft, win, hop, in_fmt, out_fmt, ch, frames):
    layer = InverseSTFT(n_fft=n_fft, win_length=win, hop_length=hop, input_data_format=in_fmt,
                        output_data_format=out_fmt)
    win = win or n_fft
    k = n_fft // 2 + 1
    g = torch.Generator().manual_seed(n_fft)
    s0 = torch.randn((2, ch, frames, k, 2), generator=g, dtype=torch.float64)
    s0 = spec_from_bcfk(torch.view_as_complex(s0), in_fmt).contiguous()
    sg = s0.to(torch.complex64).cuda().requires_grad_(True)
    y = layer(sg)
    assert y.grad_fn is not None
    r = cotangent(y.shape, False, seed=9)
    loss_of(y, r).backward()

    sr = s0.clone().requires_grad_(True)
    synth = backend.window_values(layer.window_fn, win, np.float64)
    yr = ref_istft(spec_to_bcfk(sr, in_fmt), n_fft, win, hop, synth)
    yr = yr.permute(0, 2, 1) if out_fmt == CL else yr
    assert tuple(yr.shape) == tuple(y.shape)
    loss_of(yr, r).backward()
    check(sg.grad, sr.grad, 2e-4, 'dL/dX through InverseSTFT')


def test_round_trip_backward():
    """get_perfectly_reconstructing_stft_istft (composed.py:388-417): d<istft(stft(x)), R>/dx."""
    n_fft, hop = 512, 128
    stft, istft = get_perfectly_reconstructing_stft_istft(n_fft=n_fft, hop_length=hop, waveform_data_format=CL,
                                                          stft_data_format=CL)
    x0 = wave(2, 2, 4000, CL, seed=11)
    xg = x0.cuda().requires_grad_(True)
    y = istft(stft(xg))
    r = cotangent(y.shape, False, seed=12)
    loss_of(y, r).backward()
    xr = x0.to(torch.float64).requires_grad_(True)
    window = backend.get_window_fn('hann_window')(n_fft).astype(np.float64)
    synth = backend.window_values(istft.window_fn, n_fft, np.float64)
    sr = ref_stft(to_bct(xr, CL), n_fft, n_fft, hop, window, True, True)
    yr = ref_istft(sr, n_fft, n_fft, hop, synth).permute(0, 2, 1)
    assert tuple(yr.shape) == tuple(y.shape)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 2e-4, 'dL/dx through STFT -> InverseSTFT')


# ---------------------------------------------------------------------------------------------
# Magnitude / Phase / ApplyFilterbank / LogmelToMFCC / MagnitudeToDecibel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('layer_cls,ref_fn', [(Magnitude, torch.abs), (Phase, torch.angle)])
def test_complex_to_real_backward(layer_cls, ref_fn):
    g = torch.Generator().manual_seed(21)
    s0 = torch.view_as_complex(torch.randn((2, 17, 129, 2, 2), generator=g, dtype=torch.float64))
    sg = s0.to(torch.complex64).cuda().requires_grad_(True)
    y = layer_cls()(sg)
    assert y.grad_fn is not None and y.dtype == torch.float32
    r = cotangent(y.shape, False, seed=22)
    loss_of(y, r).backward()
    sr = s0.clone().requires_grad_(True)
    loss_of(ref_fn(sr), r).backward()
    check(sg.grad, sr.grad, 2e-5, 'dL/dX through %s' % layer_cls.__name__)


@pytest.mark.parametrize('fmt', [CL, CF])
@pytest.mark.parametrize('n_freq,n_mels', [(257, 40), (1025, 128)])
def test_filterbank_backward(fmt, n_freq, n_mels):
    layer = ApplyFilterbank(type='mel', filterbank_kwargs=dict(sample_rate=22050, n_freq=n_freq, n_mels=n_mels),
                            data_format=fmt)
    g = torch.Generator().manual_seed(31)
    shape = (2, 19, n_freq, 3) if fmt == CL else (2, 3, 19, n_freq)
    x0 = torch.rand(shape, generator=g, dtype=torch.float64)
    xg = x0.to(torch.float32).cuda().requires_grad_(True)
    y = layer(xg)
    assert y.grad_fn is not None
    r = cotangent(y.shape, False, seed=32)
    loss_of(y, r).backward()
    xr = x0.clone().requires_grad_(True)
    fb = torch.as_tensor(np.asarray(layer.filterbank, np.float64))
    yr = torch.einsum('bfkc,km->bfmc', xr, fb) if fmt == CL else xr @ fb
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 2e-5, 'dL/dx through ApplyFilterbank')


def test_mfcc_backward():
    layer = LogmelToMFCC(n_mfccs=20, data_format=CL)
    g = torch.Generator().manual_seed(41)
    x0 = torch.randn((2, 23, 64, 2), generator=g, dtype=torch.float64)
    xg = x0.to(torch.float32).cuda().requires_grad_(True)
    y = layer(xg)
    r = cotangent(y.shape, False, seed=42)
    loss_of(y, r).backward()
    # the layer is linear: its Jacobian is whatever matrix the forward applies -- read it off the forward pass
    eye = torch.eye(64, dtype=torch.float32).reshape(1, 64, 64, 1).cuda()
    mat = layer(eye)[0, :, :, 0].cpu().to(torch.float64)                     # (n_mels, n_mfccs)
    xr = x0.clone().requires_grad_(True)
    loss_of(torch.einsum('bfkc,km->bfmc', xr, mat), r).backward()
    check(xg.grad, xr.grad, 2e-5, 'dL/dx through LogmelToMFCC')


@pytest.mark.parametrize('dyn,dtype', [(80.0, 'float32'), (15.0, 'float32'), (15.0, 'float64')])
def test_decibel_backward(dyn, dtype):
    """With dynamic_range 15 a large share of every item sits on the floor: its cotangent reaches the item's maximum."""
    tdt = torch.float64 if dtype == 'float64' else torch.float32
    layer = MagnitudeToDecibel(ref_value=0.7, amin=1e-3, dynamic_range=dyn, dtype=dtype)
    g = torch.Generator().manual_seed(51)
    x0 = torch.exp(3.0 * torch.randn((4, 31, 40, 2), generator=g, dtype=torch.float64)) * 1e-2   # some below amin
    x0 = x0.to(tdt).to(torch.float64)                       # the checker sees the values the layer sees
    xg = x0.to(tdt).cuda().requires_grad_(True)
    y = layer(xg)
    assert y.grad_fn is not None and y.dtype == tdt
    r = cotangent(y.shape, False, seed=52)
    loss_of(y, r).backward()
    xr = x0.clone().requires_grad_(True)
    yr = ref_db(xr, 0.7, 1e-3, dyn)
    if dyn < 80:
        assert float((yr == yr.reshape(4, -1).amin(dim=1).reshape(4, 1, 1, 1)).double().mean()) > 0.05
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), atol=2e-4 if dtype == 'float32' else 1e-9)
    loss_of(yr, r).backward()
    if dtype == 'float32':
        # elements within rounding of the floor may fall on either side of it in float32: compare where the checker's
        # distance to the floor is not marginal, and the item maxima (which collect the floor's cotangent) always
        l = yr.detach()
        thr = l.reshape(4, -1).amax(dim=1).reshape(4, 1, 1, 1) - dyn
        lf = 10.0 * torch.log10(torch.clamp(x0, min=1e-3)) - 10.0 * np.log10(0.7)
        safe = (lf - thr).abs() > 1e-3
        got = xg.grad.detach().cpu().to(torch.float64)
        scale = float(xr.grad.abs().max())
        assert float(((got - xr.grad).abs() * safe).max()) / scale <= 2e-4
        assert float(safe.double().mean()) > 0.99
    else:
        check(xg.grad, xr.grad, 1e-9, 'dL/dx through MagnitudeToDecibel (float64)')


def test_decibel_backward_ties_follow_tensorflow():
    """Exact ties: two elements share the item maximum, one element sits EXACTLY on the floor.  TensorFlow's
    conventions (/root/reference/kapre/backend.py:186-192 differentiated by tf): reduce_max splits its cotangent evenly over
    the tied maxima; tf.maximum(l, m - dyn) passes the cotangent to l when l == m - dyn (greater_equal), so the element on
    the floor keeps its own cotangent and does not feed the maximum.  Powers of two make every quantity exact in float32:
    v_log_f32(2^k) = k, l = fl(c k), and dyn := l(16) - l(8) is a Sterbenz-exact difference."""
    c = np.float32(3.01029995663981195)
    l16, l8 = np.float32(c * np.float32(4.0)), np.float32(c * np.float32(3.0))
    dyn = float(np.float32(l16 - l8))
    assert np.float32(l16 - np.float32(dyn)) == l8
    layer = MagnitudeToDecibel(ref_value=1.0, amin=1e-10, dynamic_range=dyn)
    vals = [16.0, 4.0, 8.0, 2.0, 16.0, 1.0, 0.5, 8.0]                   # max twice, 8 = on the floor (twice), four below it
    x = torch.tensor(vals, dtype=torch.float32).reshape(1, 2, 4, 1).cuda().requires_grad_(True)
    y = layer(x)
    yv = y.detach().cpu().numpy().reshape(-1)
    assert yv[0] == l16 and yv[4] == l16 and yv[2] == l8 and yv[7] == l8 and (yv[[1, 3, 5, 6]] == l8).all()
    r = torch.tensor([0.3, -1.1, 0.7, 2.0, -0.4, 0.9, 1.6, -0.2], dtype=torch.float64)
    loss_of(y, r.reshape(1, 2, 4, 1)).backward()
    rn = r.numpy()
    below = rn[[1, 3, 5, 6]].sum()
    gl = np.array([rn[0] + below / 2, 0.0, rn[2], 0.0, rn[4] + below / 2, 0.0, 0.0, rn[7]])
    want = gl * 10.0 / (np.log(10.0) * np.array(vals))
    got = x.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=0)


def test_decibel_backward_mask_agrees_with_the_forward_clamp_near_the_floor():
    """Elements within a few ulp of the floor: the backward's [l >= max - dyn] must be the mask the forward applied (both
    run to_db: v_log_f32 + one fma; round 3's backward used libm logf and could disagree by an ulp).  Checked without
    knowing l: wherever the forward output is above the floor value the cotangent must come through, the mask must be
    monotone in x, and every masked cotangent must arrive at the maximum."""
    dyn = 30.0
    layer = MagnitudeToDecibel(ref_value=1.0, amin=1e-10, dynamic_range=dyn)
    xmax = np.float32(37.25)
    x_thr = np.float32(xmax * np.float32(10.0 ** (-dyn / 10.0)))
    n = 400                                                             # consecutive floats across the threshold
    around = (x_thr.view(np.int32) + np.arange(-n // 2, n // 2, dtype=np.int32)).view(np.float32)
    vals = np.concatenate([[xmax], around]).astype(np.float32)
    x = torch.from_numpy(vals).reshape(1, 1, -1, 1).cuda().requires_grad_(True)
    y = layer(x)
    yv = y.detach().cpu().numpy().reshape(-1)
    floor_v = yv.min()
    g = torch.Generator().manual_seed(77)
    r = torch.rand(vals.shape, generator=g, dtype=torch.float64) + 0.5  # positive cotangents
    loss_of(y, r.reshape(1, 1, -1, 1)).backward()
    got = x.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
    passed = got[1:] != 0.0
    assert passed[yv[1:] > floor_v].all(), 'an element the forward left above the floor lost its cotangent'
    assert (np.diff(passed.astype(np.int32)) >= 0).all(), 'the mask is not monotone in x'
    assert 0 < passed.sum() < n, 'the run of consecutive floats does not straddle the floor'
    rn = r.numpy()
    scale = 10.0 / np.log(10.0)
    np.testing.assert_allclose(got[1:][passed], rn[1:][passed] * scale / vals[1:][passed].astype(np.float64), rtol=2e-6)
    np.testing.assert_allclose(got[0], (rn[0] + rn[1:][~passed].sum()) * scale / float(xmax), rtol=2e-6)


# ---------------------------------------------------------------------------------------------
# the fused chains
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n_fft,hop,n_mels,sr,ch,fmt,decibel', [
    (2048, 512, 128, 44100, 1, CL, False),           # the north-star configuration
    (2048, 512, 128, 44100, 1, CL, True),
    (512, 128, 40, 22050, 2, CL, True),              # the reference's own test shape
    (1024, 160, 80, 16000, 1, CF, False),
    (400, 160, 80, 16000, 1, CL, True),
])
def test_melspectrogram_backward(n_fft, hop, n_mels, sr, ch, fmt, decibel):
    t = 6 * n_fft
    model = get_melspectrogram_layer(input_shape=(t, ch) if fmt == CL else (ch, t), n_fft=n_fft, hop_length=hop,
                                     sample_rate=sr, n_mels=n_mels, return_decibel=decibel, db_dynamic_range=60.0,
                                     input_data_format=fmt, output_data_format=fmt, pad_end=True)
    x0 = wave(3, ch, t, fmt, seed=61)
    xg = x0.cuda().requires_grad_(True)
    y = model(xg)
    assert y.grad_fn is not None
    plain = model(x0.cuda())
    assert plain.grad_fn is None and torch.equal(plain, y.detach())        # forward = the fused launch either way
    r = cotangent(y.shape, False, seed=62)
    loss_of(y, r).backward()

    xr = x0.to(torch.float64).requires_grad_(True)
    window = backend.get_window_fn(None)(n_fft).astype(np.float64)
    fb = torch.as_tensor(np.asarray(model.layers[2].filterbank, np.float64))
    mag = ref_stft(to_bct(xr, fmt), n_fft, n_fft, hop, window, False, True).abs()        # (B, C, F, K)
    mel = mag @ fb
    mel = mel.permute(0, 2, 3, 1) if fmt == CL else mel
    yr = ref_db(mel, 1.0, 1e-5, 60.0) if decibel else mel
    assert tuple(yr.shape) == tuple(y.shape)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 3e-4, 'dL/dx through the fused mel chain')


def test_stft_magnitude_chain_backward():
    model = Sequential([STFT(n_fft=1024, hop_length=256, pad_begin=True), Magnitude()])
    x0 = wave(2, 2, 5000, CL, seed=71)
    xg = x0.cuda().requires_grad_(True)
    y = model(xg)
    r = cotangent(y.shape, False, seed=72)
    loss_of(y, r).backward()
    xr = x0.to(torch.float64).requires_grad_(True)
    window = backend.get_window_fn(None)(1024).astype(np.float64)
    yr = spec_from_bcfk(ref_stft(to_bct(xr, CL), 1024, 1024, 256, window, True, False), CL).abs()
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 2e-4, 'dL/dx through the fused STFT -> Magnitude')


# ---------------------------------------------------------------------------------------------
# behaviour around the tape
# ---------------------------------------------------------------------------------------------
def test_no_grad_and_detached_inputs_take_the_plain_path():
    layer = STFT(n_fft=256, hop_length=64)
    x = wave(1, 1, 1000, CL, seed=81).cuda().requires_grad_(True)
    with torch.no_grad():
        assert layer(x).grad_fn is None
    assert layer(x.detach()).grad_fn is None
    assert layer(x.detach().cpu().numpy()).grad_fn is None


# ---------------------------------------------------------------------------------------------
# Frame / Energy / Delta (signal.py:22-240, time_frequency.py:563-644)
# ---------------------------------------------------------------------------------------------
def ref_frame(x_bct, length, hop, pad_end, pad_value):
    """tf.signal.frame on the last axis: (B, C, T) -> (B, C, F, L)."""
    t = x_bct.shape[-1]
    if pad_end:
        n_frames = -(-t // hop)
        x_bct = torch.nn.functional.pad(x_bct, (0, max(0, (n_frames - 1) * hop + length - t)), value=pad_value)
    return x_bct.unfold(-1, length, hop)


@pytest.mark.parametrize('fmt', [CL, CF])
@pytest.mark.parametrize('length,hop,pad_end,t', [(256, 64, False, 3000), (200, 77, True, 2999), (64, 64, True, 1000)])
def test_frame_backward(fmt, length, hop, pad_end, t):
    layer = Frame(frame_length=length, hop_length=hop, pad_end=pad_end, pad_value=0.25, data_format=fmt)
    x0 = wave(2, 3, t, fmt, seed=91)
    xg = x0.cuda().requires_grad_(True)
    y = layer(xg)
    assert y.grad_fn is not None
    r = cotangent(y.shape, False, seed=92)
    loss_of(y, r).backward()
    xr = x0.to(torch.float64).requires_grad_(True)
    yr = ref_frame(to_bct(xr, fmt), length, hop, pad_end, 0.25)                  # (B, C, F, L)
    yr = yr.permute(0, 2, 3, 1) if fmt == CL else yr
    assert tuple(yr.shape) == tuple(y.shape)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), atol=1e-6)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 1e-6, 'dL/dx through Frame')


@pytest.mark.parametrize('fmt', [CL, CF])
@pytest.mark.parametrize('length,hop,pad_end,t', [(2205, 1102, False, 22050), (400, 160, True, 7000)])
def test_energy_backward(fmt, length, hop, pad_end, t):
    layer = Energy(sample_rate=22050, ref_duration=0.1, frame_length=length, hop_length=hop, pad_end=pad_end,
                   pad_value=0, data_format=fmt)
    x0 = wave(2, 2, t, fmt, seed=93)
    xg = x0.cuda().requires_grad_(True)
    y = layer(xg)
    assert y.grad_fn is not None
    r = cotangent(y.shape, False, seed=94)
    loss_of(y, r).backward()
    xr = x0.to(torch.float64).requires_grad_(True)
    fr = ref_frame(to_bct(xr, fmt), length, hop, pad_end, 0.0)
    yr = (fr * fr).sum(-1) * (0.1 / (length / 22050))                             # (B, C, F)
    yr = yr.permute(0, 2, 1) if fmt == CL else yr
    assert tuple(yr.shape) == tuple(y.shape)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=1e-6)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 2e-6, 'dL/dx through Energy')


@pytest.mark.parametrize('fmt', [CL, CF])
@pytest.mark.parametrize('win,mode,t', [(5, 'symmetric', 40), (9, 'reflect', 23), (3, 'constant', 17), (9, 'symmetric', 3)])
def test_delta_backward(fmt, win, mode, t):
    layer = Delta(win_length=win, mode=mode, data_format=fmt)
    n = (win - 1) // 2
    g = torch.Generator().manual_seed(95)
    shape = (2, t, 13, 3) if fmt == CL else (2, 3, t, 13)
    t_axis = 1 if fmt == CL else 2
    x0 = torch.randn(shape, generator=g, dtype=torch.float64)
    xg = x0.to(torch.float32).cuda().requires_grad_(True)
    y = layer(xg)
    assert y.grad_fn is not None
    r = cotangent(y.shape, False, seed=96)
    loss_of(y, r).backward()
    xr = x0.clone().requires_grad_(True)
    # tf.pad along time, then the correlation with [-n .. n] / (2 sum i^2) (time_frequency.py:614-635)
    if mode == 'constant':
        idx = np.arange(-n, t + n)
        valid = torch.as_tensor(((idx >= 0) & (idx < t)).astype(np.float64))
        idx = np.clip(idx, 0, t - 1)
    else:
        if mode == 'reflect' and n >= t:
            pytest.skip('tf.pad REFLECT needs pad < size')
        idx = np.pad(np.arange(t), n, mode=mode)
        valid = torch.ones(len(idx), dtype=torch.float64)
    xp = xr.index_select(t_axis, torch.as_tensor(idx))
    vshape = [1, 1, 1, 1]
    vshape[t_axis] = len(idx)
    xp = xp * valid.reshape(vshape)
    denom = 2.0 * sum(i * i for i in range(1, n + 1))
    yr = sum(j * xp.narrow(t_axis, n + j, t) for j in range(-n, n + 1)) / denom
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), atol=2e-6)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 2e-6, 'dL/dx through Delta')


def test_mfcc_front_end_backward_end_to_end():
    """waveform -> log-mel (fused) -> MFCC -> Delta: every node of a typical speech front end carries gradient."""
    model = get_melspectrogram_layer(input_shape=(8000, 1), n_fft=512, hop_length=160, sample_rate=16000, n_mels=40,
                                     return_decibel=True, input_data_format=CL, output_data_format=CL)
    x = wave(2, 1, 8000, CL, seed=97).cuda().requires_grad_(True)
    y = Delta(win_length=5, data_format=CL)(LogmelToMFCC(n_mfccs=13, data_format=CL)(model(x)))
    assert y.shape[2] == 13 and y.grad_fn is not None
    y.square().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0


# ---------------------------------------------------------------------------------------------
# corners
# ---------------------------------------------------------------------------------------------
def test_mag_phase_model_backward():
    """get_stft_mag_phase (composed.py:420-511): the spectrum feeds two branches, their cotangents add up."""
    from kapre_amd.composed import get_stft_mag_phase
    model = get_stft_mag_phase(input_shape=(4000, 2), n_fft=512, hop_length=128, return_decibel=True,
                               input_data_format=CL, output_data_format=CL)
    x0 = wave(2, 2, 4000, CL, seed=101)
    xg = x0.cuda().requires_grad_(True)
    y = model(xg)
    assert y.shape[3] == 4 and y.grad_fn is not None
    r = cotangent(y.shape, False, seed=102)
    # the phase of bins with a tiny magnitude is ill conditioned: weight its cotangent down (the same in both worlds)
    loss_of(y, r).backward()
    xr = x0.to(torch.float64).requires_grad_(True)
    window = backend.get_window_fn(None)(512).astype(np.float64)
    s = spec_from_bcfk(ref_stft(to_bct(xr, CL), 512, 512, 128, window, False, False), CL)
    yr = torch.cat([ref_db(s.abs(), 1.0, 1e-5, 80.0), torch.angle(s)], dim=3)
    loss_of(yr, r).backward()
    check(xg.grad, xr.grad, 5e-4, 'dL/dx through get_stft_mag_phase')


def test_istft_backward_follows_a_cropped_or_padded_frequency_axis():
    layer = InverseSTFT(n_fft=256, hop_length=64)
    for k in (140, 100):                                   # 129 bins expected: cropped / zero-padded by the forward
        g = torch.Generator().manual_seed(k)
        s0 = torch.view_as_complex(torch.randn((2, 9, k, 1, 2), generator=g, dtype=torch.float64))
        sg = s0.to(torch.complex64).cuda().requires_grad_(True)
        y = layer(sg)
        r = cotangent(y.shape, False, seed=k + 1)
        loss_of(y, r).backward()
        sr = s0.clone().requires_grad_(True)
        full = sr[:, :, :129] if k > 129 else torch.nn.functional.pad(torch.view_as_real(sr), (0, 0, 0, 0, 0, 129 - k))
        full = full if k > 129 else torch.view_as_complex(full.contiguous())
        synth = backend.window_values(layer.window_fn, 256, np.float64)
        yr = ref_istft(full.permute(0, 3, 1, 2), 256, 256, 64, synth).permute(0, 2, 1)
        loss_of(yr, r).backward()
        assert tuple(sg.grad.shape) == (2, 9, k, 1)
        check(sg.grad, sr.grad, 2e-4, 'dL/dX with %d bins' % k)


def test_non_contiguous_cotangents_and_retained_graphs():
    layer = STFT(n_fft=256, hop_length=64, output_data_format=CF)
    x = wave(2, 1, 2000, CL, seed=103).cuda().requires_grad_(True)
    y = Magnitude()(layer(x))                               # (B, C, F, K)
    z = y.permute(0, 1, 3, 2)                               # the cotangent reaching Magnitude is a permuted view
    w = torch.rand(z.shape, device='cuda')
    (g1,) = torch.autograd.grad((z * w).sum(), x, retain_graph=True)
    (g2,) = torch.autograd.grad((z * w).sum(), x)
    assert torch.equal(g1, g2) and float(g1.abs().max()) > 0
    xr = x.detach().cpu().to(torch.float64).requires_grad_(True)
    window = backend.get_window_fn(None)(256).astype(np.float64)
    yr = ref_stft(to_bct(xr, CL), 256, 256, 64, window, False, False).abs().permute(0, 1, 3, 2)
    (yr * w.cpu().to(torch.float64)).sum().backward()
    check(g1, xr.grad, 2e-4, 'dL/dx with a permuted cotangent')


def test_signal_shorter_than_a_frame_has_zero_gradient_and_empty_output():
    layer = STFT(n_fft=512, hop_length=128)
    x = wave(2, 1, 300, CL, seed=104).cuda().requires_grad_(True)
    y = layer(x)
    assert y.shape[1] == 0 and y.grad_fn is not None
    Magnitude()(y).sum().backward()
    assert x.grad is not None and tuple(x.grad.shape) == (2, 300, 1) and float(x.grad.abs().max()) == 0.0
