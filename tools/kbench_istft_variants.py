#!/usr/bin/env python3
"""InverseSTFT kernel paths (istft_path 0 = automatic: ring kernel where it applies, 1 = barrier kernel, 2 = irFFT +
overlap-add gather): kernel time (hipGraph of 200 launches, HIP events; min / median of four repetitions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi


def time_graph(fn, launches=200):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(launches): fn()
    torch.cuda.synchronize(); graph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


cases = [(128, 434, 1024, 256), (16, 434, 1024, 256), (4, 434, 1024, 256), (256, 83, 2048, 512), (64, 83, 2048, 512),
         (32, 860, 2048, 512), (256, 61, 512, 256), (256, 173, 512, 128), (1, 1000, 1024, 256)]
if "small" in sys.argv:      # where does the barrier kernel take over? (launches of a few thousand frames)
    cases = [(8, 434, 1024, 256), (2, 434, 1024, 256), (8, 83, 2048, 512), (16, 83, 2048, 512), (32, 83, 2048, 512),
             (2, 860, 2048, 512), (16, 61, 512, 256), (64, 61, 512, 256), (8, 173, 512, 128), (32, 173, 512, 128)]
    sys.argv.remove("small")
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
for b, f, n_fft, hop in cases:
    k = n_fft // 2 + 1
    rng = np.random.default_rng(1)
    s = torch.from_numpy((rng.standard_normal((b, f, k, 1)) + 1j * rng.standard_normal((b, f, k, 1))).astype(np.complex64)).cuda()
    layer = kapre.InverseSTFT(n_fft=n_fft, hop_length=hop)
    res = {v: [] for v in variants}
    for rep in range(5):
        for v in variants:
            _ffi.set_option("istft_path", v)
            res[v].append(time_graph(lambda: layer(s)))
    print("%4d x %4d frames n_fft %4d hop %4d  " % (b, f, n_fft, hop) +
          "   ".join("path%d %7.2f / %7.2f" % (v, min(res[v][1:]), sorted(res[v][1:])[2]) for v in variants), flush=True)
_ffi.set_option("istft_path", 0)
