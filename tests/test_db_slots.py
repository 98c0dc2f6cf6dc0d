"""-m gpu: statistics slots of the fused decibel kernels (DbDev::slot_mask; kapre_hip.hip: db_slots).  With few batch items
every workgroup's closing max / min atomics used to land on the same few words (8 six-channel items: 1024 waves on 16
addresses, +13 us on a 12 us kernel); small batches now spread them over up to 32 slots per item, which k_db_clamp
reduces.  The result must not change: one slot (`db_slots` 1, the layout of rounds 1-2) and the automatic number give the
same bits, for every kernel family, with and without an active floor (backend.magnitude_to_decibel,
/root/reference/kapre/backend.py:186-192)."""
import numpy as np
import pytest
import torch

from kapre_amd import _ffi
from kapre_amd.composed import get_melspectrogram_layer

pytestmark = pytest.mark.gpu


def _audio(batch, ch, t, fmt, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (batch, ch, t)).astype(np.float32)
    for b in range(0, batch, 2):                          # every other item: a silent half (its floor becomes active)
        x[b, :, t // 2:] *= 1e-7
    if batch > 2:
        x[batch // 3] *= 1e-3                              # a quiet item: its own, lower, maximum
    return np.ascontiguousarray(x.transpose(0, 2, 1)) if fmt == 'channels_last' else x


CASES = [
    # n_fft, hop, mels, sr, ch, T, fmt                       kernel
    (2048, 512, 128, 44100, 1, 44100, 'channels_last'),      # k_mel_ws<1024>
    (2048, 1024, 128, 44100, 6, 22050, 'channels_first'),
    (2048, 1024, 128, 44100, 6, 22050, 'channels_last'),
    (1024, 160, 80, 16000, 1, 16000, 'channels_last'),       # k_mel_ws<512>
    (1024, 160, 80, 16000, 1, 160000, 'channels_last'),      # k_mel_ts<512> from 12 k frames
    (512, 128, 40, 22050, 2, 22050, 'channels_last'),        # ring kernel / k_mel_ts<256>
    (256, 64, 40, 22050, 1, 22050, 'channels_last'),         # k_mel_ts<128>
    (400, 160, 80, 16000, 1, 48000, 'channels_last'),        # k_mel_mr
    (400, 160, 77, 16000, 2, 16000, 'channels_first'),       # M not a multiple of 4
    (1200, 300, 64, 22050, 1, 12000, 'channels_last'),       # two-kernel path (its own atomics use slot 0)
]


@pytest.mark.parametrize('batch', [1, 3, 8, 33, 100])
@pytest.mark.parametrize('n_fft,hop,n_mels,sr,ch,t,fmt', CASES)
def test_slots_do_not_change_the_result(n_fft, hop, n_mels, sr, ch, t, fmt, batch):
    if batch * ch * t > 40e6:
        pytest.skip('kept small')
    shape = (t, ch) if fmt == 'channels_last' else (ch, t)
    x = torch.from_numpy(_audio(batch, ch, t, fmt, seed=n_fft + batch)).cuda()
    outs = []
    try:
        for slots in (1, 0, 32):
            _ffi.set_option('db_slots', slots)
            model = get_melspectrogram_layer(input_shape=shape, n_fft=n_fft, hop_length=hop, sample_rate=sr, n_mels=n_mels,
                                             return_decibel=True, db_dynamic_range=40.0, input_data_format=fmt,
                                             output_data_format=fmt, pad_end=True)      # (a new model: a new workspace plan)
            y = model(x)
            assert torch.equal(model(x), y)                 # and again: the slots are initialised by every call
            outs.append(y.clone())
    finally:
        _ffi.set_option('db_slots', 0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    item = outs[0].reshape(batch, -1)
    floors = (item == item.amax(dim=1, keepdim=True) - 40.0).any(dim=1)
    assert floors.any()                                     # the silent halves sit on their items' floors
    if batch > 2:
        assert not floors.all()


def test_a_cached_plan_survives_a_change_of_the_option():
    """the mel plan caches its workspace; kpr_mel_workspace_bytes must not depend on `db_slots` (the statistics
    region always holds the 32-slot maximum), so the SAME model keeps working -- and keeps its result -- when the option
    changes between calls."""
    x = torch.from_numpy(_audio(8, 1, 22050, 'channels_last', seed=5)).cuda()
    model = get_melspectrogram_layer(input_shape=(22050, 1), n_fft=512, hop_length=128, sample_rate=22050, n_mels=40,
                                     return_decibel=True, db_dynamic_range=40.0, pad_end=True)
    try:
        _ffi.set_option('db_slots', 1)
        y1 = model(x).clone()
        _ffi.set_option('db_slots', 32)
        y32 = model(x).clone()                              # same cached plan, more slots than it was sized under
        _ffi.set_option('db_slots', 0)
        y0 = model(x).clone()
    finally:
        _ffi.set_option('db_slots', 0)
    assert torch.equal(y1, y32) and torch.equal(y1, y0)
