#!/usr/bin/env python3
"""A/B of the stand-alone ApplyFilterbank kernels ("fb_variant": 1 = the MFMA kernels, k_mel_ws<1024, FROM_MAG> for wide banks (rounds
2-5), 0 = the library's choice: k_fb_pw for banks with a band plan on contiguous rows), same process, settled clocks, same buffers and
rotating buffers (bench.py's helpers).      python tools/kbench_fb.py [variant ...] [shape=k,rows,batch,ch,fmt,n_mels]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    import torch

    from kapre_amd import ApplyFilterbank, _ffi
    shape = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("shape=")]
    k, rows, batch, ch, fmt, n_mels = (shape[0].split(",") if shape else "1025,83,256,1,channels_first,128".split(","))
    k, rows, batch, ch, n_mels = int(k), int(rows), int(batch), int(ch), int(n_mels)
    variants = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 0]
    layer = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=44100, n_freq=k, n_mels=n_mels), data_format=fmt)
    rng = np.random.default_rng(0)
    shp = (batch, rows, k, ch) if fmt == "channels_last" else (batch, ch, rows, k)
    nbytes = 4 * batch * ch * rows * (k + n_mels)
    nrot = max(5, -(-(640 << 20) // nbytes))
    xs = [torch.from_numpy(np.abs(rng.standard_normal(shp, dtype=np.float32))).cuda() for _ in range(nrot)]
    print("shape %s -> %d mels, %.1f MB per launch, %d rotating buffers" % (shp, n_mels, nbytes / 1e6, nrot))
    for rep in range(2):
        for v in variants:
            _ffi.set_option("fb_variant", v)
            us, _ = bench.kernel_time_us(layer, xs[0], launches=100, settle_s=1.0)
            usr, _ = bench.kernel_time_us(layer, xs[0], launches=100, settle_s=1.0, rotate=xs)
            print("fb_variant %d  %-22s same buffers %7.2f us (%.3f of 8 TB/s)   rotating %7.2f us (%.3f)" % (
                v, _ffi.last_launches(), us, nbytes / us / 8e6, usr, nbytes / usr / 8e6))
    _ffi.set_option("fb_variant", 0)


if __name__ == "__main__":
    main()
