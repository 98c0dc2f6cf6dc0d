#!/usr/bin/env python3
"""bench.py -- throughput of the Kapre time-frequency hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one pass of the hot path (get_melspectrogram_layer: STFT -> Magnitude ->
ApplyFilterbank[mel]) over one batch of synthetic waveforms that is already resident in HBM.
Default workload = BASELINE.json configs[1]: batch=64, 1ch, 44100 samples @44.1 kHz, n_fft=2048,
hop=512, n_mels=128 (per GPU: weak scaling, the batch axis shards with no data-path collective;
RCCL is used once to broadcast the filterbank).  Rank 0 prints ONE JSON line.

For N > 1 the driver launches this file with torch.distributed.run (one rank per GPU).
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
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_16x16x4_f32 dense peak

WORKLOADS = {
    # name: dict(batch per GPU, channels, samples, sr, n_fft, hop, n_mels, db, layout)
    "cfg2_mel_b64x1x44100_nfft2048_hop512_mel128": dict(
        batch=64, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=128, db=False, fmt="channels_last", seed=1235),
    "target_mel_b256x1x44100_nfft2048_hop512_mel128": dict(
        batch=256, ch=1, t=44100, sr=44100, n_fft=2048, hop=512, n_mels=128, db=False, fmt="channels_last", seed=1239),
    "cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cl": dict(
        batch=256, ch=6, t=44100, sr=44100, n_fft=2048, hop=1024, n_mels=128, db=True, fmt="channels_last", seed=1236),
    "cfg3_logmel_db_b256x6x44100_nfft2048_hop1024_mel128_cf": dict(
        batch=256, ch=6, t=44100, sr=44100, n_fft=2048, hop=1024, n_mels=128, db=True, fmt="channels_first", seed=1236),
    "cfg5_mel_b256x1x160000_nfft1024_hop160_mel80": dict(
        batch=256, ch=1, t=160000, sr=16000, n_fft=1024, hop=160, n_mels=80, db=False, fmt="channels_last", seed=1238),
}
DEFAULT = "cfg2_mel_b64x1x44100_nfft2048_hop512_mel128"
ALSO = "target_mel_b256x1x44100_nfft2048_hop512_mel128"


def frames_of(w):
    return 1 + (w["t"] - w["n_fft"]) // w["hop"]


def algorithmic(w):
    """per-frame algorithmic bytes and flops of the fused mel pipeline."""
    f = frames_of(w)
    k = w["n_fft"] // 2 + 1
    bytes_per_frame = 4.0 * w["t"] / f + 4.0 * w["n_mels"] * (3 if w["db"] else 1)
    flops_per_frame = 2.5 * w["n_fft"] * np.log2(w["n_fft"]) + 4 * k + 2.0 * k * w["n_mels"]
    return bytes_per_frame, flops_per_frame


def build_model(w):
    import kapre_amd as kapre

    return kapre.get_melspectrogram_layer(
        n_fft=w["n_fft"], hop_length=w["hop"], sample_rate=w["sr"], n_mels=w["n_mels"],
        return_decibel=w["db"], input_data_format=w["fmt"], output_data_format=w["fmt"])


def make_input(w, rank, device):
    import torch

    shape = (w["batch"], w["t"], w["ch"]) if w["fmt"] == "channels_last" else (w["batch"], w["ch"], w["t"])
    x = np.random.default_rng(w["seed"] + 1000 * rank).uniform(-1, 1, shape).astype(np.float32)
    return torch.from_numpy(x).to(device)


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
    return dt, dev_ms, y


def kernel_time_us(model, x, launches=200):
    """Average duration of ONE launch of the dominant kernel, measured with HIP events on the
    stream the kernel is launched on (torch's current stream), with the K launches captured into
    one hipGraph so that host launch overhead is not in the measurement (inter-kernel gaps of
    ~1-2 us remain and are part of the reported figure)."""
    import torch

    model(x)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            model(x)                                   # plan for this stream
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(launches):
                    model(x)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        graph.replay()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / launches, "hipGraph of %d launches, HIP events" % launches
    except Exception as e:  # noqa: BLE001  (graph capture unavailable: plain back-to-back launches)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(launches):
            model(x)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / launches, "back-to-back launches, HIP events (%s)" % type(e).__name__


def cpu_baseline(w, budget_s=12.0):
    """Kapre's op graph restated on the host CPU in float32 (oracle/cpu_graph.py), timed on a
    bounded sample of the same workload.  The oracle is used here only as the measured baseline."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import cpu_graph
    import kapre_oracle as oracle

    cores = os.cpu_count() or 1
    b = min(w["batch"], 16)
    x = np.random.default_rng(w["seed"]).uniform(-1, 1, (b, w["t"], w["ch"])).astype(np.float32)
    window = oracle.hann_window(w["n_fft"]).astype(np.float32)
    fb = oracle.filterbank_mel(w["sr"], w["n_fft"] // 2 + 1, w["n_mels"])
    db = (1.0, 1e-5, 80.0) if w["db"] else None
    frames = b * w["ch"] * frames_of(w)
    best = {}
    for name, fn in (("scipy.fft.rfft(workers)+sgemm", cpu_graph.melspectrogram_scipy),
                     ("torch.stft+matmul (CPU)", cpu_graph.melspectrogram_torch)):
        fn(x, window, fb, w["n_fft"], w["hop"], db)           # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            fn(x, window, fb, w["n_fft"], w["hop"], db)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or n >= 200:
                break
        best[name] = frames * n / el
    name = max(best, key=best.get)
    return {"value": best[name], "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "variant": name, "all_variants": {k: round(v, 1) for k, v in best.items()},
            "sample": "%d of %d batch items per pass, repeated for ~%.0f s per variant; CPU restatement "
                      "of Kapre's TF graph in float32 (TensorFlow is not installable in this image)"
                      % (b, w["batch"], budget_s / 2)}


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json:
    FETCH_SIZE x calibration factor + WRITE_SIZE); None when no pass exists for this workload."""
    best = None
    for name in sorted(os.listdir(os.path.join(REPO, "profiles"))) if os.path.isdir(os.path.join(REPO, "profiles")) else []:
        if name.endswith("_hbm_traffic.json"):
            with open(os.path.join(REPO, "profiles", name)) as f:
                d = json.load(f)
            if workload in d:
                best = d[workload]["hbm_bytes_per_launch"]
    return best


def roofline(w, kernel_us):
    bpf, fpf = algorithmic(w)
    frames = w["batch"] * w["ch"] * frames_of(w)
    gbs = bpf * frames / (kernel_us * 1e-6) / 1e9
    tfs = fpf * frames / (kernel_us * 1e-6) / 1e12
    return ({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": gbs / HBM_PEAK_GBS, "traffic": None,
             "kernel": "k_mel_ws" if w["n_fft"] in (1024, 2048) else "k_mel_fused", "kernel_us": kernel_us,
             "algorithmic_bytes_per_frame": bpf},
            {"bound": "mfma", "achieved": tfs, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
             "frac": tfs / MFMA_F32_PEAK_TF, "algorithmic_flops_per_frame": fpf,
             "note": "dense-equivalent flops (2*K*M GEMM + FFT + |.|); the kernel skips "
                     "filterbank tiles that are exactly zero, so issued MFMA flops are lower"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=DEFAULT, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
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
    model = build_model(w)
    bcast_bytes = kdist.broadcast_constants(model, src=0, device=device)     # RCCL, once
    x = make_input(w, rank, device)

    dt, dev_ms, y = timed_steps(model, x, args.steps, args.warmup, world)
    frames_step = w["batch"] * w["ch"] * frames_of(w)                       # per rank
    total_frames = frames_step * world * args.steps
    value = total_frames / dt
    audio_s = w["batch"] * w["ch"] * w["t"] / w["sr"] * world * args.steps / dt

    result = {
        "metric": "mel-frames/sec", "value": value, "unit": "mel-frames/s",
        "audio_sec_per_sec": audio_s,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "device_ms_per_step": dev_ms / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic uniform(-1,1) waveforms resident in HBM; filterbank/window built per Kapre defaults",
        "config": {"workload": args.workload, "per_gpu_batch": w["batch"], "channels": w["ch"],
                   "samples": w["t"], "sample_rate": w["sr"], "n_fft": w["n_fft"], "hop": w["hop"],
                   "n_mels": w["n_mels"], "return_decibel": w["db"], "layout": w["fmt"],
                   "frames_per_step_per_gpu": frames_step, "parallelism": "batch-shard x%d" % world,
                   "constants_broadcast_bytes": bcast_bytes},
    }
    if rank == 0:
        k_us, how = kernel_time_us(model, x)
        hbm, mfma = roofline(w, k_us)
        hbm["measured"] = how
        hbm["traffic"] = pmc_traffic(args.workload)
        hbm["traffic_source"] = "profiles/*_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
        hbm["algorithmic_bytes_per_launch"] = hbm["algorithmic_bytes_per_frame"] * frames_step
        result["roofline"] = hbm
        result["roofline_mfma_dense_equiv"] = mfma
        result["kernel_frames_per_s"] = frames_step / (k_us * 1e-6)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if rank == 0 and world == 1:
        if not args.no_also and args.workload == DEFAULT:
            w2 = WORKLOADS[ALSO]
            m2 = build_model(w2)
            x2 = make_input(w2, 0, device)
            dt2, _, _ = timed_steps(m2, x2, args.steps, args.warmup, 1)
            k2, _ = kernel_time_us(m2, x2)
            f2 = w2["batch"] * w2["ch"] * frames_of(w2)
            h2, _ = roofline(w2, k2)
            result["also"] = [{"workload": ALSO, "value": f2 * args.steps / dt2, "unit": "mel-frames/s",
                               "ms_per_step": dt2 / args.steps * 1e3, "kernel_us": k2,
                               "kernel_frames_per_s": f2 / (k2 * 1e-6), "hbm_frac": h2["frac"]}]
            del x2
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(w)
            result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
