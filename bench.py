#!/usr/bin/env python3
"""bench.py -- throughput of the Kapre time-frequency hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the hot path over one batch of synthetic waveforms already resident in HBM.

Headline (every N): the north-star workload -- get_melspectrogram_layer (STFT -> Magnitude ->
ApplyFilterbank[mel]) on batch=256 x 1ch x 44100 samples @44.1 kHz, n_fft=2048, hop=512, n_mels=128
PER GPU (weak scaling: the batch axis shards with no data-path collective; RCCL is used once to
broadcast the filterbank).  `metric`/`unit` are BASELINE.json's; `value` = frames of all ranks / wall
time of the K timed steps (barrier + synchronize on both sides, max over ranks).

`also` (same JSON line) carries the other BASELINE.json configs, each timed the same way with its own
algorithmic-bytes roofline: cfg2 (batch 64), cfg3 channels_last / channels_first (LogMel + dB, 256 x 6ch),
cfg4 forward (kpr_stft_f32) and inverse (kpr_istft_f32), cfg5 one GPU's share (256 items) -- and cfg5
STRONG-scaled: the 2048-item batch split over the N ranks (2048/N items each), at every N including 1, so
that the per-N lines hold both a weak (headline) and a strong (configs[4]) scaling curve.

For N > 1 the driver launches this file with torch.distributed.run (one rank per GPU).

Output.  The LAST stdout line is the result: ONE compact JSON object (< 4 KB; tests/test_bench_line.py holds it to
that) with the contract's keys + `roofline`, `roofline_compute`, `cpu_baseline`, `gpu_over_cpu`, `sustained` and a
one-row-per-workload `also` summary.  Everything longer -- per-workload detail, the CPU sweep, the notes on how to read
the fractions -- goes to gpurun_out/bench_full.json and to earlier stdout lines (one small JSON object per `also`
workload).  (Round 3 printed all of it on one 22 KB line; the driver keeps 8 KB of stdout and could not parse it.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_16x16x4_f32 dense peak = the fp32 VALU FMA peak of 256 CUs

# kind: mel = fused kpr_mel_f32 ; stft = kpr_stft_f32 (complex out) ; istft = kpr_istft_f32
WORKLOADS = {
    "target_mel_b256x1x44100_nfft2048_hop512_mel128": dict(
        kind="mel", batch=256, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=128, db=False,
        fmt="channels_last", seed=1239),
    "cfg2_mel_b64x1x44100_nfft2048_hop512_mel128": dict(
        kind="mel", batch=64, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=128, db=False,
        fmt="channels_last", seed=1235),
    "cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cl": dict(
        kind="mel", batch=256, ch=6, t=44100, sr=44100, n_fft=2048, hop=1024, n_mels=128, db=True,
        fmt="channels_last", seed=1236),
    "cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf": dict(
        kind="mel", batch=256, ch=6, t=44100, sr=44100, n_fft=2048, hop=1024, n_mels=128, db=True,
        fmt="channels_first", seed=1236),
    "cfg4_stft_b128x1x110250_nfft1024_hop256_pad": dict(
        kind="stft", batch=128, ch=1, t=110250, sr=22050, n_fft=1024, hop=256, pad=True,
        fmt="channels_last", seed=1237),
    # composed.get_stft_magnitude_layer on the cfg4 waveforms: STFT + Magnitude as one launch, float32 rows of K values
    "cfg4_stftmag_b128x1x110250_nfft1024_hop256_pad": dict(
        kind="stftmag", batch=128, ch=1, t=110250, sr=22050, n_fft=1024, hop=256, pad=True,
        fmt="channels_last", seed=1237),
    "cfg4_istft_b128x1x434f_nfft1024_hop256": dict(
        kind="istft", batch=128, ch=1, t=110250, sr=22050, n_fft=1024, hop=256, pad=True,
        fmt="channels_last", seed=1237),
    "cfg5_mel_b256x1x160000_nfft1024_hop160_mel80": dict(
        kind="mel", batch=256, ch=1, t=160000, sr=16000, n_fft=1024, hop=160, n_mels=80, db=False,
        fmt="channels_last", seed=1238),
    "cfg5_mel_b2048x1x160000_nfft1024_hop160_mel80_strong": dict(
        kind="mel", batch=2048, ch=1, t=160000, sr=16000, n_fft=1024, hop=160, n_mels=80, db=False,
        fmt="channels_last", seed=1238, strong=True),
    # the ubiquitous speech front-end (25 ms window / 10 ms hop at 16 kHz, 80 mels) and the shape the reference's own
    # mel tests use (/root/reference/tests/test_time_frequency.py:188-267: n_fft 512, sr 22050, 40 mels up to 8 kHz, dB)
    "speech_mel_b256x1x160000_nfft400_hop160_mel80": dict(
        kind="mel", batch=256, ch=1, t=160000, sr=16000, n_fft=400, hop=160, n_mels=80, db=False,
        fmt="channels_last", seed=1240),
    "reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40": dict(
        kind="mel", batch=256, ch=2, t=22050, sr=22050, n_fft=512, hop=128, n_mels=40, db=True, mel_f_max=8000.0,
        fmt="channels_last", seed=1241),
}
# stand-alone layers on the north-star shape ("HBM for K1-only runs, MFMA for K2-only runs"): K2-only = ApplyFilterbank
# on the 21 248 x 1025 magnitude rows (round 6: k_fb_pw, banded row sums on the vector ALU -- HBM bound; rounds 2-5: the fp32 MFMA
# consumers of k_mel_ws), Magnitude on the complex spectrogram, MagnitudeToDecibel on the magnitude spectrogram
WORKLOADS.update({
    "k2_filterbank_b256x83x1025_mel128": dict(
        kind="fb", batch=256, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=128, fmt="channels_last", seed=1243),
    # the same number of rows as channels_last STEREO: (item, frame) blocks of 1025 x 2 floats (round 6: the ST instances of k_fb_pw)
    "k2_filterbank_cl2_b128x83x1025x2_mel128": dict(
        kind="fb", batch=128, ch=2, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=128, fmt="channels_last", seed=1247),
    # the same rows through a bank WITHOUT a band plan (log-frequency bumps, 84 bins): the case that still runs on fp32 MFMA
    # (k_mel_ws<1024, FROM_MAG>) -- north_star's "MFMA utilisation for the filterbank stage"
    "k2_logfb_b256x83x1025_bins84": dict(
        kind="fb", bank="log", batch=256, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=84, fmt="channels_last", seed=1246),
    "magnitude_b256x83x1025": dict(
        kind="mag", batch=256, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, fmt="channels_last", seed=1244),
    "decibel_b256x83x1025": dict(
        kind="db", batch=256, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, fmt="channels_last", seed=1245),
})
DEFAULT = "target_mel_b256x1x44100_nfft2048_hop512_mel128"
STRONG = "cfg5_mel_b2048x1x160000_nfft1024_hop160_mel80_strong"
ALSO_N1 = [k for k in WORKLOADS if k not in (DEFAULT, STRONG)]


def frames_of(w):
    """Frames per signal (tf.signal.frame; cfg4 pads n_fft - hop on the left and to whole hops on the right)."""
    if w.get("pad"):
        return -(-(w["t"] + w["n_fft"] - w["hop"]) // w["hop"])
    return 1 + (w["t"] - w["n_fft"]) // w["hop"]


def algorithmic(w):
    """algorithmic HBM bytes and dense-equivalent flops PER FRAME."""
    f = frames_of(w)
    k = w["n_fft"] // 2 + 1
    fft = 2.5 * w["n_fft"] * np.log2(w["n_fft"])
    if w["kind"] == "stft":
        return 4.0 * w["t"] / f + 8.0 * k, fft
    if w["kind"] == "istft":
        return 8.0 * k + 4.0 * w["hop"], fft
    if w["kind"] == "stftmag":                              # samples in, |X| row out
        return 4.0 * w["t"] / f + 4.0 * k, fft + 4.0 * k
    if w["kind"] == "fb":                                   # |X| row in, mel row out; the (K x M) product
        return 4.0 * k + 4.0 * w["n_mels"], 2.0 * k * w["n_mels"]
    if w["kind"] == "mag":                                  # complex row in, |X| row out
        return 8.0 * k + 4.0 * k, 4.0 * k
    if w["kind"] == "db":                                   # one read + one write per value (the clamp pass returns at once: no item
        return 4.0 * k + 4.0 * k, 2.0 * k                   # of uniform noise spans 80 dB)
    # dB: the log is the fused kernel's epilogue; the clamp pass (k_db_clamp) skips every item whose minimum is already
    # above max - dynamic_range, so it is data dependent and NOT priced here (pricing it would flatter roofline.frac)
    bytes_per_frame = 4.0 * w["t"] / f + 4.0 * w["n_mels"]
    return bytes_per_frame, fft + 4 * k + 2.0 * k * w["n_mels"]


def build_model(w):
    import kapre_amd as kapre

    if w["kind"] == "mel":
        extra = {"mel_f_max": w["mel_f_max"]} if "mel_f_max" in w else {}
        return kapre.get_melspectrogram_layer(
            n_fft=w["n_fft"], hop_length=w["hop"], sample_rate=w["sr"], n_mels=w["n_mels"],
            return_decibel=w["db"], input_data_format=w["fmt"], output_data_format=w["fmt"], **extra)
    if w["kind"] == "fb" and w.get("bank") == "log":
        return kapre.ApplyFilterbank(type="log", filterbank_kwargs=dict(sample_rate=w["sr"], n_freq=w["n_fft"] // 2 + 1, n_bins=w["n_mels"]),
                                     data_format=w["fmt"])
    if w["kind"] == "fb":
        return kapre.ApplyFilterbank(type="mel", filterbank_kwargs=dict(sample_rate=w["sr"], n_freq=w["n_fft"] // 2 + 1, n_mels=w["n_mels"]),
                                     data_format=w["fmt"])
    if w["kind"] == "mag":
        return kapre.Magnitude()
    if w["kind"] == "stftmag":
        return kapre.get_stft_magnitude_layer(n_fft=w["n_fft"], hop_length=w["hop"], pad_begin=bool(w.get("pad")), pad_end=bool(w.get("pad")),
                                              input_data_format=w["fmt"], output_data_format=w["fmt"])
    if w["kind"] == "db":
        return kapre.MagnitudeToDecibel()
    stft, istft = kapre.get_perfectly_reconstructing_stft_istft(w["n_fft"], w["hop"], w["fmt"], w["fmt"])
    return stft if w["kind"] == "stft" else istft


def make_input(w, rank, device, batch):
    """uniform(-1, 1) float32 waveforms, generated on the device (SURVEY 8d input law; the parity tests use
    the numpy generator with the same seeds).  istft: the spectrum of such a batch."""
    import torch

    shape = (batch, w["t"], w["ch"]) if w["fmt"] == "channels_last" else (batch, w["ch"], w["t"])
    gen = torch.Generator(device=device)
    gen.manual_seed(w["seed"] + 1000 * rank)
    x = torch.rand(shape, generator=gen, device=device, dtype=torch.float32) * 2.0 - 1.0
    if w["kind"] == "istft":
        import kapre_amd as kapre
        stft, _ = kapre.get_perfectly_reconstructing_stft_istft(w["n_fft"], w["hop"], w["fmt"], w["fmt"])
        x = stft(x)
    if w["kind"] in ("fb", "mag", "db"):                    # the stand-alone layers take the spectrogram of such a batch
        import kapre_amd as kapre
        x = kapre.STFT(n_fft=w["n_fft"], hop_length=w["hop"], input_data_format=w["fmt"], output_data_format=w["fmt"])(x)
        if w["kind"] != "mag":
            x = kapre.Magnitude()(x)
    return x


def timed_steps(model, x, steps, warmup, world):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize; max over ranks."""
    import torch
    import torch.distributed as dist

    y = None
    for _ in range(warmup):
        y = model(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        y = model(x)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([dt, dev_ms], dtype=torch.float64, device=x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dev_ms = float(t[0]), float(t[1])
    del y
    return dt, dev_ms


def kernel_time_us(model, x, launches=100, settle_s=0.0, rotate=None):
    """Average GPU time of ONE step's kernels, measured with HIP events on the stream the kernels are
    launched on, with `launches` steps captured into one hipGraph so that host launch overhead is not in
    the measurement (inter-kernel gaps of ~1-2 us remain and are part of the reported figure).
    settle_s > 0: the graph is first replayed back to back for that long -- the part reaches its settled clocks only
    after tens of milliseconds of continuous work (round 4: 45.0 us per step from a cold start, 39.6 us over a 10 s
    run of the same graph) -- and the figure is the median of three timed replays after that.
    rotate = a list of input batches: step i reads rotate[i % n] and its output stays alive for n steps, so that consecutive
    steps touch n distinct input AND output buffers (the caller sizes n for > 256 MiB in total: what a step reads was not left
    in the Infinity Cache by the step before -- DRAM bandwidth, not MALL bandwidth)."""
    import torch

    model(x)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            model(x)                                   # plan + first-use table uploads for this stream
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                ring = [None] * (len(rotate) if rotate else 1)
                for i in range(launches):
                    ring[i % len(ring)] = model(rotate[i % len(rotate)] if rotate else x)
                del ring
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while settle_s > 0 and time.perf_counter() - t0 < settle_s:
            for _ in range(10):
                graph.replay()
            torch.cuda.synchronize()
        times = []
        for _ in range(3 if settle_s > 0 else 1):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            graph.replay()
            ev1.record()
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1) * 1e3 / launches)
        us = sorted(times)[len(times) // 2]
        how = "hipGraph of %d steps, HIP events" % launches + (", median of 3 after %.1f s of continuous replay" % settle_s if settle_s > 0 else "")
        del graph
        return us, how
    except Exception as e:  # noqa: BLE001  (graph capture unavailable: plain back-to-back launches)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(launches):
            model(x)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / launches, "back-to-back launches, HIP events (%s)" % type(e).__name__


def cpu_baseline(w, screen_s=1.0, final_s=3.0, top=2, rounds=3):
    """Kapre's op graph restated on the host CPU in float32 (oracle/cpu_graph.py), timed on the FULL batch of the
    headline workload.  Forked single-threaded worker processes PINNED to distinct physical cores (one per core first,
    SMT siblings after) are screened at a few worker counts next to the two thread-based variants (~screen_s each);
    the best `top` are then re-timed `rounds` times for >= final_s each and the MEDIAN of the best variant is the
    figure, with its min / max beside it (an unpinned 0.9 s sweep moved 2x between rounds).  The oracle
    package is used here only as the measured baseline."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import cpu_graph
    import kapre_oracle as oracle

    order, n_phys = cpu_graph.physical_cpus()
    logical = len(order)
    b = w["batch"]
    x = np.random.default_rng(w["seed"]).uniform(-1, 1, (b, w["t"], w["ch"])).astype(np.float32)
    window = oracle.hann_window(w["n_fft"]).astype(np.float32)
    fb = oracle.filterbank_mel(w["sr"], w["n_fft"] // 2 + 1, w["n_mels"])
    db = (1.0, 1e-5, 80.0) if w["db"] else None
    frames = b * w["ch"] * frames_of(w)

    def procs_variant(n):
        def run(seconds, rate_hint):
            reps = 1 if not rate_hint else int(max(1, min(400, round(seconds * rate_hint / frames))))
            return cpu_graph.throughput_procs(x, window, fb, w["n_fft"], w["hop"], db, procs=n, repeats=reps, sub=4,
                                              cpus=order)
        return run

    def loop_variant(fn):
        def run(seconds, rate_hint):
            fn()                                               # warm-up (thread pool start, plan caches)
            cnt, t0 = 0, time.perf_counter()
            while True:
                fn()
                cnt += 1
                el = time.perf_counter() - t0
                if el > seconds or cnt >= 200:
                    return frames * cnt / el
        return run

    variants = {}
    for n in sorted({min(v, b, logical) for v in (n_phys // 2, n_phys, logical) if v >= 1}):
        variants["forked x%d pinned: scipy.fft.rfft + |.| + sgemm, 4-item pieces" % n] = (n, procs_variant(n))
    nt = min(n_phys, 64)
    variants["thread pool x%d: scipy.fft.rfft + |.| + sgemm per chunk" % nt] = (
        nt, loop_variant(lambda: cpu_graph.melspectrogram_pooled(x, window, fb, w["n_fft"], w["hop"], db, workers=nt)))
    variants["torch.stft + abs + matmul, %d threads" % nt] = (
        nt, loop_variant(lambda: cpu_graph.melspectrogram_torch(x, window, fb, w["n_fft"], w["hop"], db, threads=nt)))
    screen = {}
    for name, (n, run) in variants.items():
        try:
            r1 = run(0.0, None)                                # one pass: a rate hint for sizing the timed run
            screen[name] = run(screen_s, r1)
        except Exception as e:  # noqa: BLE001
            screen[name + " [failed: %s]" % type(e).__name__] = 0.0
    best = sorted((k for k in screen if screen[k] > 0 and k in variants), key=lambda k: -screen[k])[:top]
    finals = {}
    for name in best:
        finals[name] = sorted(variants[name][1](final_s, screen[name]) for _ in range(rounds))
    name = max(finals, key=lambda k: finals[k][len(finals[k]) // 2])
    runs = finals[name]
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:  # noqa: BLE001
        model = "unknown"
    med = runs[len(runs) // 2]
    return {"value": med, "unit": "mel-frames/s", "cores": variants[name][0], "kind": "port", "variant": name,
            "min": runs[0], "max": runs[-1], "spread": (runs[-1] - runs[0]) / med, "rounds": rounds,
            "physical_cores": n_phys, "host_logical_cpus": logical, "cpu_model": model,
            "sample": "full batch (%d items = %d frames) per pass, >= %.0f s per timed run, median of %d; CPU restatement "
                      "of Kapre's TF graph in float32 (TensorFlow is not installable in this image)"
                      % (b, frames, final_s, rounds),
            "screen": {k: round(v, 1) for k, v in screen.items()},
            "finals": {k: [round(v, 1) for v in vs] for k, vs in finals.items()}}


def lib_sha16():
    """first 16 hex digits of the sha256 of the library this process loaded (the PMC passes record theirs)"""
    import hashlib
    from kapre_amd import _ffi
    try:
        return hashlib.sha256(open(_ffi.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(workload):
    """(HBM bytes per launch, provenance) from the committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json: FETCH_SIZE +
    WRITE_SIZE with the guide's gfx950 corrections); (None, None) when no pass exists for this workload.  The provenance
    names the file, the commit it was collected into and whether the pass ran on THIS binary (the
    figure comes from another run, possibly of another build)."""
    best, src = None, None
    pdir = os.path.join(REPO, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_hbm_traffic.json"):
            with open(os.path.join(pdir, name)) as f:
                d = json.load(f)
            if workload in d:
                best = d[workload]["hbm_bytes_per_launch"]
                sha = d.get("_lib_sha16")
                src = {"file": "profiles/" + name, "commit": d.get("_commit"),
                       "same_binary": (sha == lib_sha16()) if sha else None}
    return best, src


def sq_counters(workload):
    """Committed rocprofv3 SQ-counter pass of this workload's dominant kernel (profiles/*_sq_counters_<workload>.json,
    written by tools/profile_collect.py), newest round last; None when there is none."""
    best = None
    pdir = os.path.join(REPO, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_sq_counters_%s.json" % workload):
            with open(os.path.join(pdir, name)) as f:
                best = (name, json.load(f))
    return best


def issue_util(workload, cus=256, xcds=8):
    """(VALU issue cycles + MFMA busy cycles) / kernel cycles per SIMD, from the committed counter pass:
    SQ_INSTS_VALU x 4 (a packed-f32 / 3-source op issues for 4 cycles, plain VOP1/2 for 2: the kernels' mix is ~85 %
    packed, so 4 is an upper bound of ~8 %) and SQ_VALU_MFMA_BUSY_CYCLES, both summed over the chip by rocprofv3, over
    GRBM_GUI_ACTIVE / 8 (the counter has one instance per XCD and rocprofv3 reports their sum; it brackets the dispatch,
    a few us more than the kernel, so the ratios are slight under-estimates) x 4 SIMDs x CUs."""
    got = sq_counters(workload)
    if not got:
        return None
    name, c = got
    try:
        cycles = c["GRBM_GUI_ACTIVE"] / xcds
        simd_cycles = cycles * 4.0 * cus
        valu = c["SQ_INSTS_VALU"] * 4.0 / simd_cycles
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
    except (KeyError, ZeroDivisionError):
        return None
    return {"valu_issue": valu, "mfma_busy": mfma, "sum": valu + mfma, "source": "profiles/" + name,
            "lds_pipe_busy": (c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cycles * cus)) if c.get("SQ_LDS_IDX_ACTIVE") else None}


def issued_flops_per_frame(w, kernel=""):
    """Flops the fused mel kernel actually ISSUES per frame: the FFT + magnitude on the vector ALU, and the filterbank
    either as its non-zeros on the vector ALU (k_mel_pw: 2 per non-zero) or as the chunks that are not exactly zero on
    the MFMA pipe (k_mel_ws / k_mel_ts: 16 frames x 16 filters x 32 rows x 2 per chunk and tile, i.e. 1024 per frame
    and chunk; the chunk count is in the packed filterbank's header)."""
    from kapre_amd import _ffi, backend

    k = w["n_fft"] // 2 + 1
    if w.get("bank") == "log":
        fb = np.asarray(backend.filterbank_log(w["sr"], k, n_bins=w["n_mels"]), np.float32)
    else:
        fb = np.asarray(backend.filterbank_mel(w["sr"], k, w["n_mels"], **({"f_max": w["mel_f_max"]} if "mel_f_max" in w else {})),
                        np.float32)
    valu = 2.5 * w["n_fft"] * np.log2(w["n_fft"]) + 4 * k
    if "k_mel_pw" in kernel or "k_fb_pw" in kernel:
        return valu + 2.0 * float(np.count_nonzero(fb)), 0.0
    chunks = int(_ffi.filterbank_pack(fb, _ffi.filterbank_kranges(fb))[:8].view(np.uint32)[4])
    return valu, chunks * 1024.0


NOTES = {
    "roofline": "achieved = SURVEY 8(d) algorithmic bytes per launch / kernel_us; the bench re-reads the same input every "
                "step: working sets under 256 MB (target 56 MB, cfg2 14 MB) stay in the Infinity Cache, so for those "
                "`achieved` is fabric, not DRAM, bandwidth; `frac` is against the 8 TB/s HBM spec either way (measured "
                "ceilings on this part: fill 6.9, copy 5.5 TB/s).  kernel_us = hipGraph wall time of 100 steps / 100 on the "
                "launch stream: it INCLUDES every launch of a step (k_stats_init and k_db_clamp on the dB workloads) and the "
                "~1-2 us gaps between them; `kernel` = kpr_last_launches() of the library for that step.  traffic = "
                "rocprofv3 --pmc passes committed under profiles/ (another run of the same command).",
    "roofline_compute": "the fused kernel is compute-bound (SURVEY 8d): FFT + |.| + the banded mel sums on the vector ALU "
                        "(k_mel_pw) or the non-zero filterbank chunks on fp32 MFMA (k_mel_ws / k_mel_ts) share one issue "
                        "port per SIMD; peak = 256 CU x 2.4 GHz x 256 flop/clk needs an all-FMA stream -- the FFT is ~75 % "
                        "adds, so `frac` understates how busy the ALU is: issue_util (issue + matrix-pipe cycles per SIMD "
                        "cycle, committed counter pass) is the binding figure",
}


def rooflines(name, w, batch, step_us, kernel):
    bpf, fpf = algorithmic(w)
    frames = batch * w["ch"] * frames_of(w)
    gbs = bpf * frames / (step_us * 1e-6) / 1e9
    out = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           # the PMC pass profiled the workload's full batch; a strong-scaled shard launches batch / N of it
           "traffic": (lambda t: None if t is None else t * batch / w["batch"])(pmc_traffic(name)[0]),
           "traffic_from": pmc_traffic(name)[1],
           "kernel": kernel, "kernel_us": step_us, "kernel_us_covers": "all launches of one step (hipGraph)",
           "algorithmic_bytes_per_frame": bpf, "algorithmic_bytes_per_launch": bpf * frames,
           "traffic_source": "profiles/*_hbm_traffic.json (rocprofv3 --pmc, separate passes)"}
    comp = None
    if w["kind"] in ("mel", "fb"):
        valu, mfma = issued_flops_per_frame(w, kernel)
        if w["kind"] == "fb":                           # K2-only: the magnitude rows exist; nothing but the product
            valu = valu - (2.5 * w["n_fft"] * np.log2(w["n_fft"]) + 4 * (w["n_fft"] // 2 + 1))
        tfs = (valu + mfma) * frames / (step_us * 1e-6) / 1e12
        comp = {"bound": ("mfma" if mfma > 0 else "valu") if w["kind"] == "fb" else "valu+mfma", "achieved": tfs, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": tfs / MFMA_F32_PEAK_TF, "issued_valu_flops_per_frame": valu,
                "issued_mfma_flops_per_frame": mfma, "dense_equivalent_flops_per_frame": fpf,
                "issue_util": issue_util(name)}
    return out, comp


def measure(name, w, rank, world, device, steps, warmup, with_kernel=True, settle_s=1.0, rotating=True):
    """One workload on this rank's shard; returns the result dict on every rank (values are whole-job)."""
    import torch
    from kapre_amd import _ffi
    from kapre_amd import dist as kdist

    batch = w["batch"] // world if w.get("strong") else w["batch"]
    model = build_model(w)
    bcast = kdist.broadcast_constants(model, src=0, device=device)          # RCCL, once (no-op at N = 1)
    x = make_input(w, rank, device, batch)
    model(x)
    kernel = _ffi.last_launches()                      # what the library dispatched for this shape (not a table here)
    # the kernel-time measurement (a hipGraph of 100 steps replayed for settle_s, then timed) runs FIRST and on every rank:
    # it also brings the GPU to its settled clocks, so that a short timed run (the driver uses K = 20, W = 5: ~1 ms of GPU
    # work) measures the steady state and not the power-management ramp (45 vs 39.6 us per step on the headline)
    k_us, how = kernel_time_us(model, x, settle_s=settle_s) if with_kernel else (None, None)
    k_rot, n_rot = None, 0
    if with_kernel and rotating:
        # as many distinct (input, output) buffer pairs as it takes to pass 2 x 256 MiB (at least 5): DRAM, not MALL, figures
        y0 = model(x)
        pair_bytes = x.numel() * x.element_size() + y0.numel() * y0.element_size()
        del y0
        n_rot = int(min(64, max(5, -(-(2 * 256 * 2 ** 20) // pair_bytes))))
        xs = [x] + [make_input(w, rank + 17 * (i + 1), device, batch) for i in range(n_rot - 1)]
        k_rot, _ = kernel_time_us(model, x, settle_s=min(settle_s, 0.5), rotate=xs)
        del xs
    dt, dev_ms = timed_steps(model, x, steps, warmup, world)
    frames_rank = batch * w["ch"] * frames_of(w)
    res = {"workload": name, "value": frames_rank * world * steps / dt, "unit": "mel-frames/s" if w["kind"] == "mel" else "frames/s",
           "audio_sec_per_sec": batch * w["ch"] * w["t"] / w["sr"] * world * steps / dt,
           "ms_per_step": dt / steps * 1e3, "device_ms_per_step": dev_ms / steps, "steps": steps,
           "per_gpu_batch": batch, "frames_per_step_per_gpu": frames_rank,
           "scaling": "strong" if w.get("strong") else "weak", "constants_broadcast_bytes": bcast}
    if with_kernel and world > 1:                          # every rank's own kernel time (the line prints them all)
        import torch.distributed as dist
        per_rank = [None] * world
        dist.all_gather_object(per_rank, round(float(k_us), 3))
        res["kernel_us_per_rank"] = per_rank
    if with_kernel and rank == 0:
        hbm, comp = rooflines(name, w, batch, k_us, kernel)
        hbm["measured"] = how
        res["kernel_us"] = k_us
        if k_rot is not None:
            # the HBM figures of the line are the ROTATING ones (consecutive steps share no buffer: DRAM); the replay of one
            # buffer pair (what the K timed steps and rocprofv3's per-kernel average see) stays beside them
            hbm["kernel_us_same_buffers"] = k_us
            hbm["achieved_same_buffers"] = hbm["achieved"]
            hbm["frac_same_buffers"] = hbm["frac"]
            hbm["kernel_us_rotating"] = k_rot
            hbm["rotating_buffer_pairs"] = n_rot
            hbm["achieved_rotating"] = hbm["achieved"] * k_us / k_rot
            hbm["frac_rotating"] = hbm["frac"] * k_us / k_rot
            res["kernel_us_rotating"] = k_rot
        res["roofline"] = hbm
        if comp:
            res["roofline_compute"] = comp
    del x, model
    torch.cuda.empty_cache()
    return res


def sustained(name, w, rank, world, device, seconds):
    """>= `seconds` of back-to-back hipGraph replays of the headline step (100 steps per replay) on every rank: the
    thermally settled rate, and GPU activity long enough for an external SMI sampler to see (the K timed steps of the
    contract are ~1 ms of GPU work).  Returns (whole-job frames/s over the run, seconds, steps)."""
    import torch

    batch = w["batch"] // world if w.get("strong") else w["batch"]
    model = build_model(w)
    x = make_input(w, rank, device, batch)
    model(x)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        model(x)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(100):
                model(x)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(20):                                   # ~0.1 s of queued work per host round trip
            graph.replay()
        n += 20
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    del graph
    frames = batch * w["ch"] * frames_of(w)
    return frames * world * 100.0 * n / el, el, 100 * n


def _r(v, nd=4):
    """Round floats for the compact line (4 significant digits by default)."""
    if isinstance(v, float):
        return float("%.*g" % (nd, v))
    return v


def compact_line(result, also, cap=4000):
    """The final stdout line: the contract's keys + roofline / roofline_compute / cpu_baseline / gpu_over_cpu /
    sustained + one short row per `also` workload, without the explanatory strings; shrinks the `also` rows if the
    line would pass `cap` bytes."""
    keep = ["metric", "value", "unit", "audio_sec_per_sec", "n_gpus", "steps", "warmup", "ms_per_step", "settle_s",
            "device_ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "primary_time", "value_device_time", "rccl_ranks", "dist_backend", "sclk_mhz", "kernel_frames_per_s"]
    line = {k: _r(result[k], 6) if k in ("value", "ms_per_step") else _r(result[k]) for k in keep if k in result}
    rf = result.get("roofline")
    if rf:
        line["roofline"] = {k: _r(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                   "kernel_us", "kernel_us_covers", "algorithmic_bytes_per_launch")}
        for k in ("kernel_us_rotating", "frac_rotating", "rotating_buffer_pairs"):        # consecutive steps on distinct buffers (DRAM)
            if k in rf:
                line["roofline"][k] = _r(rf[k])
        tf = rf.get("traffic_from")
        if tf:
            line["roofline"]["traffic_from"] = {"commit": tf.get("commit"), "same_binary": tf.get("same_binary")}
        if result.get("kernel_us_per_rank"):
            line["roofline"]["kernel_us_per_rank"] = result["kernel_us_per_rank"]
    rc = result.get("roofline_compute")
    if rc:
        iu = rc.get("issue_util") or {}
        line["roofline_compute"] = {k: _r(rc[k]) for k in ("bound", "achieved", "peak", "unit", "frac")}
        line["roofline_compute"]["issue_util"] = {k: _r(iu.get(k), 3) for k in ("valu_issue", "mfma_busy", "sum", "lds_pipe_busy")} if iu else None
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "variant", "min", "max", "spread",
                                                       "physical_cores", "cpu_model")}
        line["cpu_baseline"]["sample"] = "full batch per pass, >=3 s per run, median of %d pinned runs" % cb.get("rounds", 3)
        # every variant that was screened (workers or threads -> frames/s): the figure is the best of them, the others lost
        line["cpu_baseline"]["screened"] = {(k.split(":")[0].replace("forked ", "").replace(" pinned", "").replace("thread pool ", "pool ")
                                             .replace("torch.stft + abs + matmul, ", "torch ")): _r(v, 3)
                                            for k, v in (cb.get("screen") or {}).items()}
        line["gpu_over_cpu"] = _r(result.get("gpu_over_cpu"))
    if result.get("sustained"):
        line["sustained"] = {k: _r(v) for k, v in result["sustained"].items()}
    rows = []
    for a in also:
        rf_ = a.get("roofline") or {}
        # us: one buffer pair replayed (what rocprofv3 averages); us_rot / frac: consecutive steps on distinct buffers (DRAM, the
        # figure to hold against 8 TB/s); mfma: matrix-pipe busy fraction of the committed counter pass (MFMA kernels)
        row = {"w": a["workload"].split("_nfft")[0].split("_b2")[0] if a["workload"][0] in "kmd" else a["workload"].split("_nfft")[0],
               "value": _r(a["value"]), "us": _r(a.get("kernel_us")), "us_rot": _r(rf_.get("kernel_us_rotating")),
               "frac": _r(rf_.get("frac_rotating", rf_.get("frac")), 3), "kernel": rf_.get("kernel")}
        iu = ((a.get("roofline_compute") or {}).get("issue_util") or {})
        if iu.get("mfma_busy"):
            row["mfma"] = _r(iu["mfma_busy"], 3)
        if (a.get("roofline_compute") or {}).get("bound") == "mfma":
            row["mfma_frac"] = _r(a["roofline_compute"]["frac"], 3)
        rows.append(row)
    line["also"] = rows
    line["detail"] = "gpurun_out/bench_full.json"
    if len(json.dumps(line)) > cap:
        for r_ in rows:
            r_.pop("kernel", None)
    if len(json.dumps(line)) > cap:
        for r_ in rows:
            r_.pop("value", None)
    if len(json.dumps(line)) > cap:
        line["also"] = [{"w": r_["w"], "us": r_["us"], "frac": r_.get("frac")} for r_ in rows]
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=DEFAULT, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--sustain", type=float, default=10.0, help="seconds of continuous headline replay at the end (0 = off)")
    ap.add_argument("--settle", type=float, default=1.5, help="seconds of continuous replay before the headline is timed (clock settling)")
    args = ap.parse_args()

    import torch
    from kapre_amd import dist as kdist
    from kapre_amd import _ffi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    _ffi.lib()
    rank, world, local = kdist.init_from_env(backend="nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    w = WORKLOADS[args.workload]
    head = measure(args.workload, w, rank, world, device, args.steps, args.warmup, settle_s=args.settle)
    result = {
        "metric": "mel-frames/sec", "value": head["value"], "unit": head["unit"],
        "audio_sec_per_sec": head["audio_sec_per_sec"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "device_ms_per_step": head["device_ms_per_step"],
        "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None,
        "dtype": "f32", "data": "synthetic uniform(-1,1) waveforms resident in HBM", "settle_s": args.settle,
        "config": {"workload": args.workload, "per_gpu_batch": head["per_gpu_batch"], "channels": w["ch"],
                   "samples": w["t"], "sample_rate": w["sr"], "n_fft": w["n_fft"], "hop": w["hop"],
                   "n_mels": w.get("n_mels"), "return_decibel": w.get("db", False), "layout": w["fmt"],
                   "frames_per_step_per_gpu": head["frames_per_step_per_gpu"],
                   "parallelism": "batch-shard x%d" % world,
                   "constants_broadcast_bytes": head["constants_broadcast_bytes"]},
    }
    if world > 1:
        # K x ~40 us sits between two barriers: their skew (tens of us) would read as scaling loss, so the per-rank HIP-event
        # time (max over ranks) is the figure to compare across N; `value` keeps the contract's wall-clock definition
        result["primary_time"] = "device_ms_per_step"
        result["value_device_time"] = head["frames_per_step_per_gpu"] * world / (head["device_ms_per_step"] * 1e-3)
        import torch.distributed as dist
        names = [None] * world
        dist.all_gather_object(names, "rank %d: %s" % (rank, torch.cuda.get_device_name(device)))
        result["rccl_ranks"] = dist.get_world_size()
        result["rank_devices"] = names
        result["dist_backend"] = dist.get_backend()
        result["kernel_us_per_rank"] = head.get("kernel_us_per_rank")     # every rank's own hipGraph figure for the headline step
    if rank == 0:
        try:
            result["sclk_mhz"] = round(_ffi.sclk_mhz(), 1)        # under a dense packed-f32 load (2400 = spec maximum)
        except Exception:  # noqa: BLE001
            result["sclk_mhz"] = None
        result["roofline"] = head.get("roofline")
        if head.get("roofline_compute"):
            result["roofline_compute"] = head["roofline_compute"]
        result["kernel_frames_per_s"] = head["frames_per_step_per_gpu"] / (head["kernel_us"] * 1e-6)

    also = []
    if not args.no_also and args.workload == DEFAULT:
        sub_steps, sub_warm = max(10, min(args.steps, 50)), max(3, min(args.warmup, 10))
        # configs[4] strong-scaled over the ranks: every rank takes part (2048 / N items each)
        also.append(measure(STRONG, WORKLOADS[STRONG], rank, world, device, max(5, sub_steps // 5), sub_warm))
        if world == 1:
            for name in ALSO_N1:
                also.append(measure(name, WORKLOADS[name], rank, world, device, sub_steps, sub_warm))
    if rank == 0:
        for a in also:                                            # one small line per workload, BEFORE the result line
            print(json.dumps({"also": a["workload"], "value": _r(a["value"]), "unit": a["unit"],
                              "kernel_us": _r(a.get("kernel_us")), "ms_per_step": _r(a["ms_per_step"]),
                              "roofline_frac": _r((a.get("roofline") or {}).get("frac")),
                              "kernel": (a.get("roofline") or {}).get("kernel")}), flush=True)
    if rank == 0 and not args.no_cpu_baseline:
        # at every N (round 5): rank 0's host cores, while the other ranks wait at the barrier below -- outside every timed
        # region.  gpu_over_cpu stays the ONE-GPU ratio: the N = 1-equivalent rate of this rank over the CPU figure.
        result["cpu_baseline"] = cpu_baseline(w)
        one_gpu = result["value"] if world == 1 else head["frames_per_step_per_gpu"] / (head["ms_per_step"] * 1e-3)
        result["gpu_over_cpu"] = one_gpu / result["cpu_baseline"]["value"]
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if args.sustain > 0:
        sv, sec, nsteps = sustained(args.workload, w, rank, world, device, args.sustain)
        result["sustained"] = {"value": sv, "unit": head["unit"], "seconds": sec, "steps": nsteps,
                               "us_per_step": sec / nsteps * 1e6}
    if rank == 0:
        full = dict(result)
        full["also"] = also
        full["notes"] = NOTES
        try:
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", "bench_full.json"), "w") as f:
                json.dump(full, f, indent=1)
        except OSError:
            pass
        print(json.dumps(compact_line(result, also)), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
