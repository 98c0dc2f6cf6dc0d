"""The fixed-seed fuzz gate (`-m gpu`): ~430 configurations of the path against the float64 oracle, drawn
so that EVERY dispatch threshold of the library is crossed in both directions and every instance of the large-launch
kernels is reached -- the class of bug that shipped twice in round 4 (a new large-launch instance x an untested layout pair x
a launch size past a dispatch threshold) was only found by tools/fuzz_parity.py, which no gate runs.

* thresholds (kapre_hip.hip; 256 CUs): k_stft3 from 8 frame groups per CU and its CL instances; k_istft_pw from 3/4 item per
  CU (all nine <n_fft, hop> instances + the interleaved ones), more items than workgroups, > 256 items, several segments per
  signal, k_istft_fused up to 3072 frames, the ring kernel in between; k_mel_pw with 4 / 8 / 16 waves per workgroup (4 and 16
  tickets per CU), its PAIR form from 24 pair tickets per CU; k_mel_mr; k_mel_ts / k_mel_ws for banks without a band plan.
* every (fmt_in, fmt_out) pair, C in {1, 2, 3, 4, 6}, windows shorter than n_fft, padded edges, decibels on / off.
* `kpr_last_launches()` of every configuration is recorded: a configuration that names a kernel asserts that this kernel
  ran, and the last test asserts that every instance in REQUIRED was reached by some configuration.
* tolerance: north_star's contract 1e-4 of the ITEM's scale, and next to it a regression bound of 4e-6 (measured: 1e-7 ...
  8e-7; a kernel that loses three digits must not pass because the contract is loose).

Reference ops: kapre/time_frequency.py:146-187 (STFT), :289-319 (InverseSTFT), composed.py:138-261 (mel chain), :264-385
(log-frequency chain).  ~60 s on the box.
"""
import numpy as np
import pytest

import kapre_oracle as o

pytestmark = pytest.mark.gpu

CL, CF = "channels_last", "channels_first"
PAIRS = [(CL, CL), (CL, CF), (CF, CL), (CF, CF)]
CONTRACT, REGRESSION = 1e-4, 4e-6
REACHED = {}          # kernel label -> number of configurations that launched it


def _nframes(t, n_fft, win, hop, pad_b, pad_e):
    """frames of tf.signal.stft as Kapre calls it (kapre/time_frequency.py:164-185): left pad n_fft - hop, frame length win"""
    return o.num_frames(t + (n_fft - hop if pad_b else 0), win, hop, pad_e)


def _stft_kernel(n_fft, ch, fi, fo, total):
    """kapre_hip.hip launch_stft_inst on 256 CUs, complex / magnitude output"""
    g = 64 // (n_fft // 32)
    cfast = ch > 1 and (fi == CL or fo == CL)
    cl_out = fo == CL and ch > 1
    clinst = cfast and ch % g == 0 and (n_fft == 1024 or cl_out)
    if -(-total // g) >= 8 * 256 and (not cl_out or (cfast and ch % g == 0)):
        return ("k_stft3_cl<%d" if clinst else "k_stft3<%d") % (n_fft // 2)
    return "k_stft<%d" % (n_fft // 2)


def _mel_kernel(n_fft, ch, fi, total):
    """kapre_hip.hip kpr_mel_f32, banks with a band plan, on 256 CUs"""
    g = 64 // (n_fft // 32)
    tickets = -(-total // g)
    if fi == CL and ch > 1 and ch % 2 == 0 and n_fft in (1024, 2048) and (n_fft == 2048 or ch >= 4) and tickets >= 24 * 256:
        return "k_mel_pw_pair<%d>" % (n_fft // 2)
    return "k_mel_pw<%d,w%d>" % (n_fft // 2, 16 if tickets >= 16 * 256 else 8 if tickets >= 4 * 256 else 4)


def _cfgs():
    """the configuration list: deterministic, independent of the device (built at collection time)"""
    rng = np.random.default_rng(20260925)
    out = []

    def add(kind, expect=None, **kw):
        kw["kind"], kw["expect"], kw["seed"] = kind, expect, len(out)
        if kind in ("stft", "mel", "logf"):                           # the waveform length is part of the configuration
            pad_e = kw.get("pad", (False, kw.get("pad_end", False)))[1]
            m = max(kw["win"], kw["n_fft"])
            t = m + (kw["frames"] - 1) * kw["hop"] - int(rng.integers(0, kw["hop"]))
            kw["t"] = t if pad_e else max(t, m)
        out.append(kw)
        return kw

    def around(threshold_frames, per_item):
        """batch sizes whose launch is the smallest at / the largest below `threshold_frames` frames"""
        hi = -(-threshold_frames // per_item)
        return [hi, hi - 1] if hi > 1 else [hi]

    # ---- STFT: k_stft3 / k_stft3_cl / k_stft on both sides of 8 groups per CU (2048 groups), every layout pair -------------
    for n_fft, hop in ((1024, 256), (2048, 512), (1024, 160), (2048, 1024)):
        g = 64 // (n_fft // 32)                                       # frames per wave
        for ch in (1, 2, 3, 4, 6):
            for fi, fo in (PAIRS if ch > 1 else [(CL, CL)]):
                frames = int(rng.choice([37, 83, 129]))
                pad = (bool(rng.integers(2)), bool(rng.integers(2)))
                c = add("stft", None, n_fft=n_fft, hop=hop, win=n_fft, ch=ch, fi=fi, fo=fo, frames=frames, batch=1, pad=pad)
                f = _nframes(c["t"], n_fft, n_fft, hop, *pad)
                bs = around(2047 * g + 1, f * ch)                     # groups = ceil(frames / g) >= 2048
                c["batch"], c["expect"] = bs[0], _stft_kernel(n_fft, ch, fi, fo, bs[0] * ch * f)
                if len(bs) > 1:
                    c2 = add("stft", None, n_fft=n_fft, hop=hop, win=n_fft, ch=ch, fi=fi, fo=fo, frames=frames, batch=bs[1], pad=pad)
                    c2["t"] = c["t"]
                    c2["expect"] = _stft_kernel(n_fft, ch, fi, fo, bs[1] * ch * f)
    # ---- STFT: every transform family at small and medium sizes ---------------------------------------------------------
    for n_fft in (256, 512, 1024, 2048, 400, 320, 1000, 480, 4096, 384, 250, 96, 8192):
        for rep in range(3):
            hop = int(rng.choice([n_fft // 4, n_fft // 2, max(1, n_fft // 4 - 3), int(rng.integers(1, n_fft + 1))]))
            win = int(rng.choice([n_fft, n_fft, max(2, n_fft - int(rng.integers(1, n_fft // 2)))]))
            ch = int(rng.choice([1, 2, 3, 4, 6]))
            fi, fo = PAIRS[int(rng.integers(4))]
            frames = int(rng.choice([1, 3, 17, 60, 200]))
            batch = int(rng.choice([1, 2, 5, 16]))
            if frames * batch * ch * n_fft > 3e7:                     # keeps the float64 oracle in fractions of a second
                batch = max(1, int(3e7 // (frames * ch * n_fft)))
            add("stft", None, n_fft=n_fft, hop=hop, win=win, ch=ch, fi=fi, fo=fo, frames=frames, batch=batch,
                pad=(bool(rng.integers(2)), bool(rng.integers(2))), phase=(rep == 2))
    # ---- InverseSTFT: the nine k_istft_pw instances (>= 3/4 item per CU), > 256 items, several segments, IL instances ---------
    for n_fft in (512, 1024, 2048):
        nstr = 16 * (64 // (n_fft // 32))
        for s_ in (2, 4, 8):
            hop, r = n_fft * s_ // 16, 16 // s_
            need = nstr * (r - 1)
            lab = "k_istft_pw<%d,s%d>" % (n_fft // 2, s_)
            add("istft", lab, n_fft=n_fft, hop=hop, win=n_fft, ch=1, fi=CF, fo=CF, frames=need + 5, batch=max(197, 3300 // (need + 5) + 1))
            add("istft", lab, n_fft=n_fft, hop=hop, win=n_fft, ch=2, fi=CF, fo=CF, frames=need + 1, batch=max(101, 1700 // (need + 1) + 1))
            if s_ == 8:                                               # more items than workgroups, > 256 items
                add("istft", lab, n_fft=n_fft, hop=hop, win=n_fft, ch=1, fi=CL, fo=CL, frames=need + 3, batch=301)
            if s_ == 4:                                               # several segments per signal (halo frames), win < n_fft
                add("istft", lab, n_fft=n_fft, hop=hop, win=n_fft, ch=1, fi=CF, fo=CF, frames=7 * need + 11, batch=40)
                add("istft", lab, n_fft=n_fft, hop=hop, win=n_fft - 2 * hop + 2 if n_fft > 512 else n_fft, ch=3, fi=CF, fo=CF,
                    frames=need + 2, batch=67)
            # just short of 3/4 item per CU: the ring kernel (or the barrier kernel below 3072 frames)
            add("istft", "k_istft_ws<%d" % (n_fft // 2) if 190 * (need + 1) > 3072 else None, n_fft=n_fft, hop=hop, win=n_fft, ch=1,
                fi=CF, fo=CF, frames=need + 1, batch=190)
            if s_ != 2:                                               # interleaved instances: power-of-two channel counts
                for ch in (2, 4):
                    for fi, fo in ((CL, CL), (CL, CF), (CF, CL)):
                        runs = nstr // ch
                        add("istft", "k_istft_pw_il<%d,s%d>" % (n_fft // 2, s_), n_fft=n_fft, hop=hop, win=n_fft, ch=ch, fi=fi, fo=fo,
                            frames=runs * (r - 1) + 4, batch=max(195, 3200 // (runs * (r - 1) + 4) + 1))
    # ---- InverseSTFT: barrier kernel, ring kernel (RJ 2 / 4 / 8), two-kernel path, mixed radix, Bluestein -------------------
    for n_fft, hop, win, frames, batch, ch in ((1024, 256, 1024, 40, 4, 1), (2048, 512, 2048, 83, 16, 2), (512, 128, 512, 173, 64, 1),
                                               (1024, 512, 1024, 300, 24, 1), (1024, 128, 1024, 260, 30, 1), (1024, 200, 1000, 200, 40, 2),
                                               (512, 100, 400, 150, 60, 3), (400, 100, 400, 200, 64, 1), (1000, 250, 1000, 120, 32, 2),
                                               (480, 120, 480, 90, 40, 1), (4096, 1024, 4096, 20, 6, 1), (250, 60, 250, 50, 8, 2),
                                               (320, 80, 300, 260, 48, 1), (2048, 300, 2018, 60, 9, 2), (256, 64, 256, 400, 70, 1)):
        for fi, fo in (PAIRS if ch > 1 else [(CF, CF)]):
            add("istft", None, n_fft=n_fft, hop=hop, win=win, ch=ch, fi=fi, fo=fo, frames=frames, batch=batch)
    # ---- fused mel: k_mel_pw with 4 / 8 / 16 waves per workgroup = tickets on both sides of 4 and 16 per CU, and the PAIR form on
    # both sides of 24 tickets per CU (interleaved input, even channel count; n_fft 2048, or four channels and more at n_fft 1024)
    for n_fft, hop, sr, n_mels in ((2048, 512, 44100, 128), (1024, 160, 16000, 80), (512, 128, 22050, 40), (256, 64, 16000, 64),
                                   (2048, 1024, 44100, 128), (1024, 256, 22050, 96)):
        g = 64 // (n_fft // 32)
        for thr in (4 * 256, 16 * 256, 24 * 256):
            for ch, fi, fo in ((1, CL, CL), (2, CF, CF), (3, CL, CF), (2, CL, CL), (4, CL, CL), (6, CL, CF), (4, CF, CL)):
                if thr == 24 * 256 and not (fi == CL and ch % 2 == 0 and n_fft >= 1024):
                    continue                                          # (only the PAIR form has a threshold there)
                if thr != 24 * 256 and ch > 3 and n_fft < 1024:
                    continue
                frames = int(rng.choice([23, 61, 97]))
                pad_e = bool(rng.integers(2))
                c = add("mel", None, n_fft=n_fft, hop=hop, win=n_fft, ch=ch, fi=fi, fo=fo, frames=frames, batch=1, sr=sr, n_mels=n_mels,
                        db=bool(rng.integers(2)), pad_end=pad_e)
                f = _nframes(c["t"], n_fft, n_fft, hop, False, pad_e)
                bs = around((thr - 1) * g + 1, f * ch)
                c["batch"], c["expect"] = bs[0], _mel_kernel(n_fft, ch, fi, bs[0] * ch * f)
                if len(bs) > 1:
                    c2 = add("mel", None, n_fft=n_fft, hop=hop, win=n_fft, ch=ch, fi=fi, fo=fo, frames=frames, batch=bs[1], sr=sr,
                             n_mels=n_mels, db=c["db"], pad_end=pad_e)
                    c2["t"] = c["t"]
                    c2["expect"] = _mel_kernel(n_fft, ch, fi, bs[1] * ch * f)
    # ---- fused mel: small and medium launches over every transform family, mixed radix, short windows, odd hops ---------------
    for n_fft in (256, 512, 1024, 2048, 400, 320, 1000, 480, 640, 96, 4096, 250):
        for rep in range(5):
            hop = int(rng.choice([n_fft // 4, n_fft // 2, max(1, n_fft // 4 - 3), int(rng.integers(1, n_fft + 1))]))
            win = int(rng.choice([n_fft, n_fft, max(2, n_fft - int(rng.integers(1, n_fft // 2)))]))
            ch = int(rng.choice([1, 1, 2, 3, 4, 6]))
            fi, fo = PAIRS[int(rng.integers(4))]
            frames = int(rng.choice([1, 3, 17, 60, 200, 700]))
            batch = int(rng.choice([1, 2, 5, 16, 48]))
            if frames * batch * ch * n_fft > 3e7:
                batch = max(1, int(3e7 // (frames * ch * n_fft)))
            add("mel", None, n_fft=n_fft, hop=hop, win=win, ch=ch, fi=fi, fo=fo, frames=frames, batch=batch,
                sr=int(rng.choice([16000, 22050, 44100])), n_mels=int(rng.choice([40, 64, 80, 128])), db=bool(rng.integers(2)),
                pad_end=bool(rng.integers(2)))
    # ---- log-frequency spectrograms (banks without a band plan: the MFMA kernels k_mel_ts / k_mel_ws) --------------------------
    for n_fft, hop, sr, frames, batch, ch, exp in ((2048, 512, 44100, 83, 64, 1, "k_mel_ws<1024>"), (1024, 256, 22050, 80, 40, 1, "k_mel_ws<512>"),
                                                   (1024, 256, 22050, 83, 160, 1, "k_mel_ts<512>"), (512, 128, 22050, 170, 48, 1, "k_mel_ts<256>"),
                                                   (512, 128, 22050, 100, 700, 1, "k_mel_ts<256>"), (256, 64, 16000, 250, 32, 1, "k_mel_ts<128>"),
                                                   (1024, 160, 16000, 99, 24, 2, None), (2048, 700, 44100, 31, 9, 3, "k_mel_ws<1024>"),
                                                   (400, 160, 16000, 98, 16, 1, None)):
        for fi, fo in (PAIRS if ch > 1 else [(CL, CL), (CF, CF)]):
            add("logf", exp if (ch == 1 or fi == fo) else None, n_fft=n_fft, hop=hop, win=n_fft, ch=ch, fi=fi, fo=fo, frames=frames,
                batch=batch, sr=sr, db=bool(rng.integers(2)))
    # ---- the stand-alone layers: ApplyFilterbank (MFMA consumers / thin GEMM / generic GEMM), Magnitude, decibels ---------------
    for k, rows, batch, ch, nm, fmt in ((1025, 83, 256, 1, 128, CF), (1025, 83, 64, 2, 128, CL), (513, 400, 32, 1, 80, CL), (257, 3000, 4, 3, 40, CF),
                                        (1025, 7, 3, 6, 13, CL), (129, 60, 16, 2, 20, CF), (1025, 173, 24, 4, 128, CL), (513, 1, 1, 1, 64, CL)):
        add("layers", None, k=k, rows=rows, batch=batch, ch=ch, n_mels=nm, fmt=fmt, sr=int(rng.choice([16000, 44100])))
    # ---- round 6: forward transforms with win_length > n_fft (frames cut at win_length, cropped to n_fft: time_frequency.py:174-182),
    # every FFT family; the stand-alone ApplyFilterbank on contiguous rows (k_fb_pw) at every n_freq ----------------------------------
    for n_fft in (256, 512, 1024, 2048, 400, 1000, 480, 300, 4096, 1200):
        for rep in range(2):
            win = n_fft + int(rng.integers(1, n_fft))
            hop = int(rng.choice([n_fft // 4, n_fft // 2, int(rng.integers(1, n_fft + 1))]))
            ch = int(rng.choice([1, 2, 3]))
            fi, fo = PAIRS[int(rng.integers(4))]
            frames = int(rng.choice([1, 5, 40, 120]))
            batch = int(rng.choice([1, 3, 8]))
            if frames * batch * ch * n_fft > 1e7:
                batch = max(1, int(1e7 // (frames * ch * n_fft)))
            add("stft", None, n_fft=n_fft, hop=hop, win=win, ch=ch, fi=fi, fo=fo, frames=frames, batch=batch,
                pad=(bool(rng.integers(2)), bool(rng.integers(2))), phase=(rep == 1))
    for k, rows, batch, ch, nm in ((129, 300, 7, 1, 20), (257, 50, 9, 2, 40), (513, 1000, 40, 1, 80), (1025, 1, 1, 1, 128), (1025, 300, 30, 3, 96),
                                   (257, 2000, 33, 1, 64), (129, 5, 2, 6, 13)):
        add("layers", "k_fb_pw<%d>" % (k - 1), k=k, rows=rows, batch=batch, ch=ch, n_mels=nm, fmt=CF, sr=int(rng.choice([16000, 44100])))
    for k, rows, batch, nm in ((513, 200, 8, 80), (1025, 40, 5, 128), (257, 1, 1, 40)):      # two interleaved channels: the ST instances
        add("layers", "k_fb_pw<%d,st>" % (k - 1), k=k, rows=rows, batch=batch, ch=2, n_mels=nm, fmt=CL, sr=int(rng.choice([16000, 44100])))
    return out


CFGS = _cfgs()


def _id(c):
    keys = ("n_fft", "hop", "win", "ch", "frames", "batch")
    lay = "%s-%s" % (c.get("fi", c.get("fmt", ""))[9:10], c.get("fo", "")[9:10])
    return "%03d-%s-%s-%s" % (c["seed"], c["kind"], "x".join(str(c[k]) for k in keys if k in c), lay)


def _item_err(got, want):
    """(max error of any batch item relative to that item's own scale, the same relative to the scale of the batch)"""
    got, want = np.asarray(got, np.float64 if not np.iscomplexobj(want) else np.complex128), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0
    b = want.shape[0]
    d = np.abs(got - want).reshape(b, -1).max(axis=1)
    sc = np.abs(want).reshape(b, -1).max(axis=1)
    return float((d / np.maximum(sc, 1e-30 + 1e-12 * sc.max())).max())


def _run(c):
    from kapre_amd import (STFT, InverseSTFT, Magnitude, Phase, Sequential, ApplyFilterbank, MagnitudeToDecibel, LogmelToMFCC, Delta, Frame,
                           Energy, composed, _ffi)
    rng = np.random.default_rng(1000 + c["seed"])
    kind = c["kind"]
    errs, label = [], ""
    if kind == "layers":
        k, rows, b, ch, fmt = c["k"], c["rows"], c["batch"], c["ch"], c["fmt"]
        xs = np.abs(rng.standard_normal((b, rows, k, ch) if fmt == CL else (b, ch, rows, k))).astype(np.float32) ** 3
        fbl = ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=c["sr"], n_freq=k, n_mels=c["n_mels"]), data_format=fmt)
        errs.append(_item_err(fbl(xs).cpu().numpy(), o.apply_filterbank(xs, o.filterbank_mel(c["sr"], k, c["n_mels"]), fmt)))
        label = _ffi.last_launches()
        got = MagnitudeToDecibel()(xs).cpu().numpy()
        errs.append(_item_err(10.0 ** (got / 10.0), 10.0 ** (o.magnitude_to_decibel(xs) / 10.0)))
        label += " + " + _ffi.last_launches()
        z = (rng.standard_normal(xs.shape) + 1j * rng.standard_normal(xs.shape)).astype(np.complex64)
        errs.append(_item_err(Magnitude()(z).cpu().numpy(), np.abs(z.astype(np.complex128))))
        label += " + " + _ffi.last_launches()
        # a bank without k-ranges of its own kind (log-frequency bumps): the generic MFMA GEMM
        lfb = ApplyFilterbank(type="log", filterbank_kwargs=dict(sample_rate=c["sr"], n_freq=k), data_format=fmt)
        errs.append(_item_err(lfb(xs).cpu().numpy(), o.apply_filterbank(xs, o.filterbank_log(c["sr"], k), fmt)))
        label += " + " + _ffi.last_launches()
        # the consumers of the path: MFCC = DCT-II as a thin GEMM, Delta, Frame, Energy
        lm = rng.standard_normal((b, rows, c["n_mels"], ch) if fmt == CL else (b, ch, rows, c["n_mels"])).astype(np.float32)
        nmf = min(13, c["n_mels"])
        errs.append(_item_err(LogmelToMFCC(n_mfccs=nmf, data_format=fmt)(lm).cpu().numpy(), o.kapre_logmel_to_mfcc(lm, nmf, fmt)))
        label += " + " + _ffi.last_launches()
        if rows >= 9:
            errs.append(_item_err(Delta(win_length=9, data_format=fmt)(lm).cpu().numpy(), o.kapre_delta(lm, 9, "symmetric", fmt)))
        w = rng.standard_normal((b, rows * 37 + 400, ch) if fmt == CL else (b, ch, rows * 37 + 400)).astype(np.float32)
        errs.append(_item_err(Frame(400, 160, data_format=fmt)(w).cpu().numpy(), o.kapre_frame(w, 400, 160, data_format=fmt)))
        errs.append(_item_err(Energy(frame_length=400, hop_length=160, data_format=fmt)(w).cpu().numpy(),
                              o.kapre_energy(w, frame_length=400, hop_length=160, data_format=fmt)))
        return errs, label
    n_fft, hop, win, ch, fi, fo, frames, batch = (c[k] for k in ("n_fft", "hop", "win", "ch", "fi", "fo", "frames", "batch"))
    if kind == "istft":
        kk = n_fft // 2 + 1
        shape = (batch, frames, kk, ch) if fi == CL else (batch, ch, frames, kk)
        s = (rng.standard_normal(shape, dtype=np.float32) + 1j * rng.standard_normal(shape, dtype=np.float32)).astype(np.complex64)
        kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, input_data_format=fi, output_data_format=fo)
        want = o.kapre_istft(s, **kw)
        got = InverseSTFT(**kw)(s).cpu().numpy()
        return [_item_err(got, want)], _ffi.last_launches()
    t = c["t"]
    pad_b, pad_e = c.get("pad", (False, c.get("pad_end", False)))
    x = rng.standard_normal((batch, t, ch) if fi == CL else (batch, ch, t), dtype=np.float32)
    x *= np.logspace(-2, 0, batch, dtype=np.float32).reshape((batch, 1, 1))            # items of very different scale
    if kind == "stft":
        kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_begin=pad_b, pad_end=pad_e, input_data_format=fi, output_data_format=fo)
        want = o.kapre_stft(x, **kw)
        errs.append(_item_err(STFT(**kw)(x).cpu().numpy(), want))
        label = _ffi.last_launches()
        errs.append(_item_err(Sequential([STFT(**kw), Magnitude()])(x).cpu().numpy(), np.abs(want)))
        label += " + " + _ffi.last_launches()
        if c.get("phase"):
            ph = Sequential([STFT(**kw), Phase()])(x).cpu().numpy()
            strong = np.abs(want) > 1e-3 * np.abs(want).max()                            # the angle of a bin near zero is noise
            dphi = np.angle(np.exp(1j * (ph - np.angle(want))))
            assert float(np.abs(dphi[strong]).max()) < 2e-3
            label += " + " + _ffi.last_launches()
        return errs, label
    if kind == "mel":
        kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, pad_end=pad_e, sample_rate=c["sr"], n_mels=c["n_mels"],
                  return_decibel=c["db"], input_data_format=fi, output_data_format=fo)
        want = o.kapre_melspectrogram(x, **kw)
        got = composed.get_melspectrogram_layer(**kw)(x).cpu().numpy()
    else:
        kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, sample_rate=c["sr"], return_decibel=c["db"], input_data_format=fi,
                  output_data_format=fo)
        fb = o.filterbank_log(c["sr"], n_fft // 2 + 1)
        mag = np.abs(o.kapre_stft(x, n_fft=n_fft, win_length=win, hop_length=hop, input_data_format=fi, output_data_format=fo))
        want = o.apply_filterbank(mag, fb, fo)
        if c["db"]:
            want = o.magnitude_to_decibel(want)
        got = composed.get_log_frequency_spectrogram_layer(**kw)(x).cpu().numpy()
    label = _ffi.last_launches()
    if c["db"]:                                       # back to linear: a dB error far below the item maximum is not an error
        got, want = 10.0 ** (np.asarray(got, np.float64) / 10.0), 10.0 ** (want / 10.0)
    return [_item_err(got, want)], label


@pytest.mark.parametrize("c", CFGS, ids=[_id(c) for c in CFGS])
def test_fuzz_configuration(c):
    import torch
    errs, label = _run(c)
    for part in label.replace(" + ", "+").split("+"):
        if part:
            REACHED[part] = REACHED.get(part, 0) + 1
    e = max(errs)
    assert e <= CONTRACT, "north_star contract: relative error %.3g [%s]" % (e, label)
    assert e <= REGRESSION, "regression bound: relative error %.3g (measured 1e-7 ... 8e-7) [%s]" % (e, label)
    if c["expect"] and torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert c["expect"] in label, "expected %s, the library launched [%s]" % (c["expect"], label)


REQUIRED = (["k_istft_pw<%d,s%d>" % (nc, s) for nc in (256, 512, 1024) for s in (2, 4, 8)] +
            ["k_istft_pw_il<%d,s%d>" % (nc, s) for nc in (256, 512, 1024) for s in (4, 8)] +
            ["k_mel_pw<%d,w%d>" % (nc, w) for nc in (128, 256, 512, 1024) for w in (4, 8, 16)] +
            ["k_mel_pw_pair<512>", "k_mel_pw_pair<1024>", "k_stft3<512,complex>", "k_stft3<512,magnitude>", "k_stft3<1024,complex>",
             "k_stft3<1024,magnitude>", "k_stft3_cl<512,complex>", "k_stft3_cl<512,magnitude>", "k_stft3_cl<1024,complex>",
             "k_stft3_cl<1024,magnitude>", "k_stft<128,complex", "k_stft<256,magnitude", "k_stft<512,complex,cl>", "k_stft<1024,magnitude,cl>",
             "k_stft<512,phase", "k_stft_mr", "k_stft_bs", "k_stft_big", "k_istft_fused", "k_istft_ws<", "k_istft_ws_mr", "k_irfft", "k_ola",
             "k_mel_mr<200>", "k_mel_ts<128>", "k_mel_ts<256>", "k_mel_ts<512>", "k_mel_ws<512>", "k_mel_ws<1024>", "k_thin_gemm", "k_gemm",
             "k_db_log", "k_db_clamp", "k_stats_init", "k_cplx_to_real", "k_fb_pw<128>", "k_fb_pw<256>", "k_fb_pw<512>", "k_fb_pw<1024>", "k_fb_pw<1024,st>",
             "k_fb_pw<512,st>"])


def test_every_instance_was_reached():
    """runs last: the union of kpr_last_launches() over the configurations above"""
    import torch
    if not REACHED:
        pytest.skip("the configurations did not run in this session")
    assert len(CFGS) >= 400
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the launch sizes are drawn for 256 CUs")
    missing = [r for r in REQUIRED if not any(r in k for k in REACHED)]
    assert not missing, (missing, sorted(REACHED))
