#!/usr/bin/env python3
"""Generate tests/golden/tflite_stft_cases.npz by running the reference's OWN primitive-op
restatement of tf.signal.stft -- /root/reference/kapre/tflite_compatible_stft.py (`stft_tflite`:
gather framing, right zero padding, matmul DFT) -- which upstream asserts equal to tf.signal.stft
(/root/reference/tests/test_time_frequency.py:270-337).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    python oracle/make_golden_l0.py

Unlike make_golden.py, NO oracle arithmetic is involved: the `tf.*` symbols that file touches
(matmul, pad, reshape, gather, slice, stack, concat ...) resolve to one-line numpy primitives in
oracle/ref_stubs/tensorflow, and the analysis windows come from scipy.signal (periodic Hann /
Hamming for even lengths, i.e. what tf.signal.*_window(periodic=True) documents).  The fixture
therefore pins the oracle's framing (pad_end / frame count), right-padding when
frame_length < fft_length, and the unnormalised rDFT to reference code, independently of
oracle/kapre_oracle.py.  The reference's matrices are complex64, so agreement is ~1e-6 relative.
"""
import importlib
import os
import sys
import types

import numpy as np
import scipy.signal

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("KAPRE_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

#        name            T     frame fft   step pad_end window
CASES = [("pow2",        3000, 512,  512,  256, False, "hann"),
         ("pow2_padend", 3000, 512,  512,  256, True,  "hann"),
         ("short_win",   2500, 200,  256,  80,  False, "hamming"),
         ("short_win_pe", 2500, 200, 256,  80,  True,  "hann"),
         ("n1000",       4000, 1000, 1000, 250, False, "hann"),
         ("hop_gt_win",  2000, 64,   128,  96,  False, "hann"),
         ("tiny",        40,   8,    8,    4,   False, "hann")]


def main():
    sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
    sys.path.insert(0, HERE)
    pkg = types.ModuleType("kapre")
    pkg.__path__ = [os.path.join(REF, "kapre")]
    sys.modules["kapre"] = pkg
    tfl = importlib.import_module("kapre.tflite_compatible_stft")
    src = np.load(os.path.join(OUT, "speech_test_file.npz"))["audio_data"].astype(np.float32)
    out = {}
    for name, t, frame, fft, step, pad_end, wname in CASES:
        x = np.stack([src[100:100 + t], 0.5 * src[900:900 + t]])[None]          # (1, 2, T)
        w = scipy.signal.get_window(wname, frame, fftbins=True).astype(np.float64)
        y = tfl.stft_tflite(x, frame, step, fft, lambda n, _w=w: _w, pad_end)     # (1, 2, F, K, 2)
        out[name + "/x"] = x.astype(np.float32)
        out[name + "/y"] = np.asarray(y, np.float64)
        out[name + "/window"] = w
        out[name + "/params"] = np.array([frame, fft, step, int(pad_end)], np.int64)
        print(name, x.shape, "->", np.asarray(y).shape)
    np.savez_compressed(os.path.join(OUT, "tflite_stft_cases.npz"), **out)


if __name__ == "__main__":
    main()
