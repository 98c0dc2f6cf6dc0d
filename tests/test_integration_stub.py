"""integration/kapre_hip_binding.py (the stub INTEGRATION.md shows) against include/kapre_hip.h: struct layouts, the
argument list of every prototype it binds, the DLPack pointer extraction, and -- on a GPU box -- one STFT through it."""
import ctypes
import os
import re
import sys
import types

import numpy as np
import pytest

from conftest import REPO, rel_err

sys.path.insert(0, os.path.join(REPO, "integration"))
import kapre_hip_binding as kb  # noqa: E402

HEADER = open(os.path.join(REPO, "include", "kapre_hip.h")).read()
_C2CT = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
         "double": ctypes.c_double, "kpr_stream_t": ctypes.c_void_p}


def _struct_fields(name):
    body = re.search(r"typedef struct \{([^}]*)\}\s*%s;" % name, HEADER).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [(m.group(2), _C2CT[m.group(1)]) for m in re.finditer(r"(\w+)\s+(\w+);", body)]


def _prototype(name):
    m = re.search(r"\n([\w \*]+?)\s*\b%s\(([^;]*?)\);" % name, HEADER, re.S)
    ret, args = m.group(1).strip(), [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
    if args == ["void"]:
        args = []
    return ret, args


def _kind(carg):
    if "*" in carg:
        for s, t in (("kpr_stft_geom", ctypes.POINTER(kb.StftGeom)), ("kpr_db_params", ctypes.POINTER(kb.DbParams))):
            if s in carg:
                return t
        return ctypes.c_void_p if "char" not in carg else ctypes.c_char_p
    return _C2CT[carg.replace("const ", "").split()[0]]


def test_stub_structs_match_the_header():
    assert [(n, t) for n, t in kb.StftGeom._fields_] == _struct_fields("kpr_stft_geom")
    assert [(n, t) for n, t in kb.DbParams._fields_] == _struct_fields("kpr_db_params")


@pytest.mark.parametrize("name", sorted(kb.PROTOTYPES))
def test_stub_prototypes_match_the_header(name):
    ret, args = _prototype(name)
    res, argtypes = kb.PROTOTYPES[name]
    assert len(args) == len(argtypes), (args, argtypes)
    for carg, ct in zip(args, argtypes):
        assert _kind(carg) is ct, (name, carg, ct)
    assert (ctypes.c_char_p if "char" in ret else _C2CT[ret.replace("const ", "").split()[0]]) is res


def test_dlpack_pointer_extraction_on_real_capsules():
    import torch
    from torch.utils.dlpack import to_dlpack
    t = torch.arange(24, dtype=torch.float32).reshape(4, 6)
    assert kb.dlpack_data_ptr(to_dlpack(t)) == t.data_ptr()
    v = t[1:, 2:]                                       # a view: storage offset 8 elements
    assert kb.dlpack_data_ptr(to_dlpack(v)) == v.data_ptr() == t.data_ptr() + 8 * 4
    a = np.arange(10, dtype=np.float64)
    assert kb.dlpack_data_ptr(a.__dlpack__()) == a.ctypes.data


@pytest.mark.gpu
def test_stft_through_the_stub_matches_the_oracle():
    import torch
    from torch.utils.dlpack import to_dlpack
    import kapre_oracle as o
    from kapre_amd import backend
    kb.load(os.path.join(REPO, "kapre_amd", "lib", "libkapre_hip.so"))
    layer = types.SimpleNamespace(n_fft=512, win_length=400, hop_length=160, pad_begin=True, pad_end=False,
                                  input_data_format="channels_last", output_data_format="channels_first")
    x = np.random.default_rng(3).uniform(-1, 1, (3, 8000, 2)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    wd = torch.from_numpy(backend.hann_window(400)).cuda()
    f = kb.num_frames(x.shape, layer)
    out = torch.empty((3, 2, f, 257), dtype=torch.complex64, device="cuda")
    kb.stft(to_dlpack(xd), x.shape, layer, to_dlpack(torch.view_as_real(out)), to_dlpack(wd),
            stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = o.kapre_stft(x, 512, 400, 160, None, True, False, "channels_last", "channels_first")
    assert rel_err(out.cpu().numpy(), want) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("with_db", [False, True])
def test_fused_melspectrogram_through_the_stub_matches_the_oracle(with_db):
    """kpr_filterbank_kranges / _pack on the host, upload, kpr_mel_f32 -- the path the fused Keras layer would take."""
    import torch
    from torch.utils.dlpack import to_dlpack
    import kapre_oracle as o
    from kapre_amd import backend
    kb.load(os.path.join(REPO, "kapre_amd", "lib", "libkapre_hip.so"))
    layer = types.SimpleNamespace(n_fft=1024, win_length=1024, hop_length=256, pad_begin=False, pad_end=True,
                                  input_data_format="channels_last", output_data_format="channels_last")
    x = np.random.default_rng(5).uniform(-1, 1, (3, 16000, 2)).astype(np.float32)
    fb = np.asarray(backend.filterbank_mel(sample_rate=16000, n_freq=513, n_mels=64, f_min=0.0, f_max=8000.0),
                    dtype=np.float32)
    bank = kb.MelFilterbank(fb)
    assert bank.packed is not None and bank.kranges.shape == (8,)
    db = kb.DbParams(1, 1.0, 1e-5, 80.0) if with_db else None
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(backend.hann_window(1024)).cuda()
    fbd, pkd = torch.from_numpy(bank.fb).cuda(), torch.from_numpy(bank.packed).cuda()
    f = kb.num_frames(x.shape, layer)
    out = torch.empty((3, f, 64, 2), dtype=torch.float32, device="cuda")
    nws = kb.mel_workspace_bytes(x.shape, layer, bank, db)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device="cuda")
    kb.melspectrogram(to_dlpack(xd), x.shape, layer, bank, to_dlpack(fbd), to_dlpack(pkd), to_dlpack(wd),
                      to_dlpack(out), to_dlpack(ws), nws, db=db, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    mag = np.abs(o.kapre_stft(x, 1024, 1024, 256, None, False, True, "channels_last", "channels_last"))
    want = o.apply_filterbank(mag, fb, "channels_last")
    if with_db:
        want = o.magnitude_to_decibel(want, 1.0, 1e-5, 80.0)
        assert np.max(np.abs(out.cpu().numpy() - want)) <= 2e-3          # dB, absolute (as tests/test_gpu_parity.py)
    else:
        assert rel_err(out.cpu().numpy(), want) <= 1e-4
