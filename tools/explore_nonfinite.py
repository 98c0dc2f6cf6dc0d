#!/usr/bin/env python3
"""What every filterbank kernel returns for a magnitude row with one NaN / one Inf bin (development aid for tests/test_nonfinite.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kapre_amd import ApplyFilterbank, _ffi, backend

def classes(a):
    return "".join("N" if np.isnan(v) else "+" if v == np.inf else "-" if v == -np.inf else "." for v in a)

def run(k, n_mels, fbtype, variant, rows=40, fmt="channels_first"):
    rng = np.random.default_rng(0)
    x = np.abs(rng.standard_normal((2, 1, rows, k))).astype(np.float32)
    x[0, 0, 3, k // 3] = np.nan
    x[0, 0, 5, k // 2] = np.inf
    x[1, 0, 7, 0] = np.inf
    x[1, 0, 9, k - 1] = np.nan
    kw = dict(sample_rate=22050, n_freq=k, n_mels=n_mels) if fbtype == "mel" else dict(sample_rate=22050, n_freq=k)
    layer = ApplyFilterbank(type=fbtype, filterbank_kwargs=kw, data_format=fmt)
    prev = _ffi.set_option("fb_variant", variant)
    try:
        y = layer(x).cpu().numpy()
    finally:
        _ffi.set_option("fb_variant", prev)
    fb = np.asarray(layer.filterbank, np.float64)
    with np.errstate(all="ignore"):
        want = x.astype(np.float64) @ fb
    print("== k %d %s n_filt %d variant %d: %s" % (k, fbtype, fb.shape[1], variant, _ffi.last_launches()))
    for (b, r) in ((0, 3), (0, 5), (1, 7), (1, 9), (0, 4)):
        print("  row (%d,%2d) got  %s\n             want %s" % (b, r, classes(y[b, 0, r])[:100], classes(want[b, 0, r])[:100]))

if __name__ == "__main__":
    run(1025, 128, "mel", 0); run(1025, 128, "mel", 1); run(513, 80, "mel", 0)
    run(1025, 0, "log", 0); run(257, 40, "mel", 0); run(257, 40, "mel", 1)
    run(256, 40, "mel", 0)   # thin gemm (n_freq % 4 == 0)
    run(1300, 64, "mel", 0)  # generic GEMM
