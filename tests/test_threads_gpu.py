"""-m gpu: the C ABI under concurrent callers ("re-entrant and thread-safe for distinct streams; cache initialisation
guarded").  Eight Python threads, each on its own HIP stream, each with a different layer chain of the path, start TOGETHER in a
fresh process -- so the first use of every cache (twiddle tables, windows, packed filterbanks and their verification cache, LDS
opt-ins, the status word, the CU count) happens under contention -- and then run 25 rounds each.  ctypes drops the GIL around
every kpr_* call, so the calls really overlap.  Every output of every round must equal (bit for bit) what the same chain
produced single-threaded in a process of its own (chains the parity tests pin to the oracle).  Child processes: cold caches
cannot be had in a process that has already run other tests."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, json, threading, hashlib
sys.path.insert(0, os.environ["KPR_REPO"])
import numpy as np
import torch
import kapre_amd as kapre
from kapre_amd import _ffi, composed

mode = sys.argv[1]                     # "serial": one thread, chain after chain; "threads": all chains at once
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def wave(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).uniform(-1, 1, shape).astype(np.float32)).to(dev)


def chains():
    st, ist = composed.get_perfectly_reconstructing_stft_istft(1024, 256, "channels_last", "channels_last")
    st2, ist2 = composed.get_perfectly_reconstructing_stft_istft(400, 100, "channels_first", "channels_first")
    return [
        ("mel2048", composed.get_melspectrogram_layer(n_fft=2048, hop_length=512, sample_rate=44100, n_mels=128), wave((24, 44100, 1), 1)),
        ("logmel512_db_stereo", composed.get_melspectrogram_layer(n_fft=512, hop_length=128, sample_rate=22050, n_mels=40, return_decibel=True,
                                                                  input_data_format="channels_first", output_data_format="channels_first"), wave((16, 2, 22050), 2)),
        ("mel400_speech", composed.get_melspectrogram_layer(n_fft=400, hop_length=160, sample_rate=16000, n_mels=80), wave((8, 32000, 1), 3)),
        ("logfreq1024", composed.get_log_frequency_spectrogram_layer(n_fft=1024, hop_length=256, sample_rate=22050, return_decibel=True), wave((8, 22050, 2), 4)),
        ("stft_istft_1024", kapre.Sequential([st, ist]), wave((12, 30000, 1), 5)),
        ("stft_istft_400_cf", kapre.Sequential([st2, ist2]), wave((6, 3, 9000), 6)),
        ("stftmag_phase_2048", composed.get_stft_mag_phase((30000, 2), n_fft=2048, hop_length=512, return_decibel=True), wave((4, 30000, 2), 7)),
        ("mfcc", kapre.Sequential([composed.get_melspectrogram_layer(n_fft=1024, hop_length=160, sample_rate=16000, n_mels=40, return_decibel=True),
                                   kapre.LogmelToMFCC(n_mfccs=13)]), wave((8, 16000, 1), 8)),
    ]


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


work = chains()
out = {}
errors = []
if mode == "serial":
    for name, model, x in work:
        ds = {digest(model(x)) for _ in range(3)}
        assert len(ds) == 1, (name, ds)
        out[name] = ds.pop()
else:
    gate = threading.Barrier(len(work))

    def run(name, model, x):
        try:
            stream = torch.cuda.Stream()
            gate.wait()                                              # every first call of the process happens now, together
            seen = set()
            with torch.cuda.stream(stream):
                for _ in range(25):
                    y = model(x)
                    stream.synchronize()
                    seen.add(digest(y))
            out[name] = sorted(seen)
        except Exception as ex:                                      # noqa: BLE001
            errors.append("%s: %r" % (name, ex))

    threads = [threading.Thread(target=run, args=w) for w in work]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
status = _ffi.device_status(raise_on_error=False)
print(json.dumps({"out": out, "errors": errors, "status": status}))
'''


def _run(tmp_path, mode):
    env = dict(os.environ)
    env["KPR_REPO"] = REPO
    script = os.path.join(str(tmp_path), "child_%s.py" % mode)
    with open(script, "w") as f:
        f.write(CHILD)
    p = subprocess.run([sys.executable, script, mode], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_eight_threads_on_eight_streams_from_cold_caches(tmp_path):
    serial = _run(tmp_path, "serial")
    assert serial["status"] == 0 and not serial["errors"]
    for attempt in range(2):                                         # (two cold starts: the first-use races are the point)
        par = _run(tmp_path, "threads")
        assert not par["errors"], par["errors"]
        assert par["status"] == 0
        assert set(par["out"]) == set(serial["out"])
        for name, want in serial["out"].items():
            assert par["out"][name] == [want], (attempt, name, par["out"][name], want)
