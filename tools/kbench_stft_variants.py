#!/usr/bin/env python3
"""STFT kernel variants (stft_variant 0 = automatic: k_stft3 from 8 frame groups per CU up, k_stft below; 1 = k_stft; 3 = k_stft3 wherever it applies; k_stft2, variant 2, was removed in round 5):
kernel time (hipGraph of 200 launches, HIP events; five repetitions, min / median) for complex and magnitude output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kapre_amd as kapre
from kapre_amd import _ffi
from kapre_amd.keras_shim import Sequential


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


cases = [(128, 110250, 1024, 256, True), (16, 110250, 1024, 256, True), (256, 44100, 2048, 512, False),
         (64, 44100, 2048, 512, False), (32, 441000, 2048, 512, False), (256, 44100, 2048, 1024, False)]
variants = [int(a) for a in sys.argv[1:]] or [0, 1]
for b, t, n_fft, hop, pads in cases:
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, t, 1)).astype(np.float32)).cuda()
    st = kapre.STFT(n_fft=n_fft, hop_length=hop, pad_begin=pads, pad_end=pads, window_name="hann_window")
    for mode, model in (("complex", st), ("magnitude", Sequential([st, kapre.Magnitude()]))):
        res = {v: [] for v in variants}
        for rep in range(5):
            for v in variants:
                _ffi.set_option("stft_variant", v)
                res[v].append(time_graph(lambda: model(x)))
        print("%4d x %6d n_fft %4d hop %4d %-9s " % (b, t, n_fft, hop, mode) +
              "   ".join("v%d %6.2f / %6.2f" % (v, min(res[v][1:]), sorted(res[v][1:])[2]) for v in variants), flush=True)
_ffi.set_option("stft_variant", 0)
