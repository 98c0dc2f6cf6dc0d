#!/usr/bin/env python3
"""One pass over every dispatch decision the library makes by itself: for a grid of shapes and launch sizes, the kernel
time of the automatic choice against every forced alternative (mel_variant, stft_variant, istft_path, db_slots), and a
flag on each row where the automatic choice is more than 3 % behind the best one.  Meant as the FIRST GPU call of a round:
the thresholds in kapre_hip.hip were set on a handful of shapes each, and in round 3 three of them turned out wrong
elsewhere.  ~2-3 minutes on the GPU box.
    python tools/sweep_dispatch.py [mel] [logf] [stft] [istft] [db] [mr]          default: all but mr
mr (round 6, before the prune of the mixed-radix ISTFT ring instances): every mixed-radix / two-pass size, the inverse transform
through the ring kernel k_istft_ws_mr (istft_path 0 / 3) against irFFT + overlap-add as two kernels (istft_path 2), and the fused
mel chain k_mel_mr (mel_variant 0) against the two-launch path (3), on a speech-sized batch"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import kapre_amd as kapre  # noqa: E402
from kapre_amd import _ffi  # noqa: E402
from kapre_amd.keras_shim import Sequential  # noqa: E402

TOL = 1.03


def time_graph(fn, launches=100, reps=3):
    """kernel time per call: hipGraph of `launches` calls, HIP events, best of `reps`"""
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(launches):
                fn()
    torch.cuda.synchronize(); graph.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / launches)
    return best


def report(label, option, values, make_fn):
    """values[0] is the automatic setting; make_fn() builds the callable AFTER the option is set (plans are cached)"""
    row = []
    for v in values:
        _ffi.set_option(option, v)
        try:
            row.append(time_graph(make_fn()))
        except RuntimeError as e:            # a forced variant that does not apply to this shape
            row.append(float("nan"))
    _ffi.set_option(option, values[0])
    best = np.nanmin(row)
    flag = "" if row[0] <= TOL * best else "   <-- automatic is %.0f %% behind %s=%d" % (
        100 * (row[0] / best - 1), option, values[int(np.nanargmin(row))])
    print("%-66s %s%s" % (label, " ".join("%s=%d %7.2f" % (option[:4], v, t) for v, t in zip(values, row)), flag), flush=True)


def mel_rows():
    shapes = ["target_mel_b256x1x44100_nfft2048_hop512_mel128", "cfg5_mel_b256x1x160000_nfft1024_hop160_mel80",
              "reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40", "cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf",
              "cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cl", "speech_mel_b256x1x160000_nfft400_hop160_mel80"]
    for name in shapes:
        for batch in (1, 4, 16, 64, 256):
            w = dict(bench.WORKLOADS[name]); w["batch"] = batch
            x = bench.make_input(w, 0, torch.device("cuda", 0), batch)
            variants = [0, 3, 4, 5, 6, 7] if w["n_fft"] in (512, 1024, 2048) else [0, 3]     # 5 / 6 / 7: k_mel_pw with 8 / 4 / 16 waves (1, k_mel_fused, was removed in round 5)

            def make(w=w, x=x):
                model = bench.build_model(w)
                return lambda: model(x)
            report("mel  %-48s batch %4d" % (name[:48], batch), "mel_variant", variants, make)


def logfreq_rows():
    """banks WITHOUT a band plan (log-frequency spectrograms: wide overlapping bumps): k_mel_pw does not apply, the choice is
    between k_mel_ws (3), k_mel_ts (4) and the two-launch path (the 4-wave ring kernel k_mel_fused, variant 1, was removed in round 5)"""
    for b, t, sr, n_fft, hop, ch in [(256, 44100, 44100, 2048, 512, 1), (16, 44100, 44100, 2048, 512, 1), (256, 22050, 22050, 1024, 256, 1),
                                     (16, 22050, 22050, 1024, 256, 1), (256, 22050, 22050, 512, 128, 1), (16, 22050, 22050, 512, 128, 1),
                                     (64, 22050, 22050, 512, 128, 2), (256, 16000, 16000, 256, 64, 1), (64, 160000, 16000, 1024, 160, 2)]:
        x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, t, ch)).astype(np.float32)).cuda()

        def make(sr=sr, n_fft=n_fft, hop=hop, x=x):
            model = kapre.get_log_frequency_spectrogram_layer(n_fft=n_fft, hop_length=hop, sample_rate=sr, return_decibel=True)
            return lambda: model(x)
        report("logf %4d x %6d x %d n_fft %4d hop %4d" % (b, t, ch, n_fft, hop), "mel_variant", [0, 3, 4], make)


def stft_rows():
    for b, t, n_fft, hop in [(128, 110250, 1024, 256), (16, 110250, 1024, 256), (2, 110250, 1024, 256), (256, 44100, 2048, 512),
                             (64, 44100, 2048, 512), (8, 44100, 2048, 512), (32, 441000, 2048, 512), (256, 44100, 2048, 1024),
                             (256, 16000, 512, 256), (256, 22050, 512, 128), (64, 160000, 1024, 160)]:
        x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (b, t, 1)).astype(np.float32)).cuda()
        st = kapre.STFT(n_fft=n_fft, hop_length=hop)
        for mode, model in (("complex", st), ("magnitude", Sequential([st, kapre.Magnitude()]))):
            report("stft %4d x %6d n_fft %4d hop %4d %-9s" % (b, t, n_fft, hop, mode), "stft_variant", [0, 1, 3],
                   lambda model=model, x=x: (lambda: model(x)))


def istft_rows():
    for b, f, n_fft, hop in [(128, 434, 1024, 256), (16, 434, 1024, 256), (8, 434, 1024, 256), (2, 434, 1024, 256),
                             (256, 83, 2048, 512), (64, 83, 2048, 512), (32, 83, 2048, 512), (8, 83, 2048, 512),
                             (256, 61, 512, 256), (64, 61, 512, 256), (16, 61, 512, 256), (256, 173, 512, 128),
                             (32, 173, 512, 128), (64, 173, 512, 128), (1, 1000, 1024, 256)]:
        k = n_fft // 2 + 1
        rng = np.random.default_rng(1)
        s = torch.from_numpy((rng.standard_normal((b, f, k, 1)) + 1j * rng.standard_normal((b, f, k, 1))).astype(np.complex64)).cuda()
        layer = kapre.InverseSTFT(n_fft=n_fft, hop_length=hop)
        report("istft %4d x %4d frames n_fft %4d hop %4d" % (b, f, n_fft, hop), "istft_path", [0, 1, 3],
               lambda layer=layer, s=s: (lambda: layer(s)))


def mr_rows():
    sizes = [160, 200, 320, 400, 640, 800, 1000, 96, 120, 192, 240, 360, 384, 480, 600, 720, 768, 960]
    for n_fft in sizes:
        hop = n_fft // 4
        k = n_fft // 2 + 1
        for b, secs in ((64, 10), (4, 10)):
            f = (secs * 16000 - n_fft) // hop + 1
            rng = np.random.default_rng(1)
            s = torch.from_numpy((rng.standard_normal((b, f, k, 1)) + 1j * rng.standard_normal((b, f, k, 1))).astype(np.complex64)).cuda()
            layer = kapre.InverseSTFT(n_fft=n_fft, hop_length=hop)
            report("mr istft %3d x %5d frames n_fft %4d hop %4d" % (b, f, n_fft, hop), "istft_path", [0, 2, 3],
                   lambda layer=layer, s=s: (lambda: layer(s)))
            x = torch.from_numpy(rng.uniform(-1, 1, (b, secs * 16000, 1)).astype(np.float32)).cuda()

            def make(n_fft=n_fft, hop=hop, x=x):
                model = kapre.get_melspectrogram_layer(n_fft=n_fft, hop_length=hop, sample_rate=16000, n_mels=40)
                return lambda: model(x)
            report("mr mel   %3d x %5d frames n_fft %4d hop %4d" % (b, f, n_fft, hop), "mel_variant", [0, 3], make)


def db_rows():
    for name, over in (("speech_mel_b256x1x160000_nfft400_hop160_mel80", {"db": True}),
                       ("reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40", {}),
                       ("cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf", {}),
                       ("target_mel_b256x1x44100_nfft2048_hop512_mel128", {"db": True})):
        for batch in (1, 8, 32, 128, 256, 512):
            w = dict(bench.WORKLOADS[name]); w.update(over); w["batch"] = batch
            x = bench.make_input(w, 0, torch.device("cuda", 0), batch)

            def make(w=w, x=x):
                model = bench.build_model(w)
                return lambda: model(x)
            report("dB   %-48s batch %4d" % (name[:48], batch), "db_slots", [0, 1, 4, 32], make)


if __name__ == "__main__":
    groups = [a for a in sys.argv[1:]] or ["mel", "logf", "stft", "istft", "db"]
    for g_ in groups:
        {"mel": mel_rows, "logf": logfreq_rows, "stft": stft_rows, "istft": istft_rows, "db": db_rows, "mr": mr_rows}[g_]()
