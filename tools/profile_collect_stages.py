#!/usr/bin/env python3
"""Distil gpurun_out/prof_<round>_stages/ into profiles/<round>_stage_evidence.json: per kernel
(k_stft complex instances, k_istft_ws, k_mel_ws<1024,true> = stand-alone filterbank) the rocprofv3 average
duration, HBM bytes (FETCH_SIZE x2 + WRITE_SIZE, KiB) and the MFMA-busy share."""
import csv, glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(REPO, "gpurun_out", "prof_%s_stages" % rnd)


def rows(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    for key, tag in (("k_stft<512, 0,", "k_stft<512, complex> (cfg4 forward, 55 552 frames, n_fft 1024)"),
                     ("k_stft<1024, 0,", "k_stft<1024, complex> (256 x 44100, 21 248 frames, n_fft 2048)"),
                     ("k_istft_ws<512,", "k_istft_ws<512, 4> (cfg4 inverse, 55 552 frames, n_fft 1024, hop 256)"),
                     ("k_mel_ws<1024, true>", "k_mel_ws<1024, FROM_MAG> (stand-alone mel filterbank, 21 248 x 1025 -> 128)")):
        if key in name:
            return tag
    return None


out = {}
for r in rows(os.path.join(src, "stats"), "kernel_stats.csv"):
    t = short(r["Name"])
    if t:
        out.setdefault(t, {})["avg_us_rocprof"] = float(r["AverageNs"]) / 1e3
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = {}
    for r in rows(d, "counter_collection.csv"):
        t = short(r["Kernel_Name"])
        if t:
            acc.setdefault((t, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (t, c), v in acc.items():
        out.setdefault(t, {})[c] = sum(v) / len(v)
for t, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        if "avg_us_rocprof" in d:
            d["hbm_GBps"] = d["hbm_bytes"] / d["avg_us_rocprof"] / 1e3
            d["hbm_frac_of_8TBps"] = d["hbm_GBps"] / 8000.0
    if d.get("SQ_INSTS_MFMA") and "avg_us_rocprof" in d:
        d["mfma_issued_TFLOPs"] = d["SQ_INSTS_MFMA"] * 2048.0 / d["avg_us_rocprof"] / 1e6   # 16x16x4 f32 = 2048 flop
        d["mfma_issued_frac_of_157TF"] = d["mfma_issued_TFLOPs"] / 157.3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "avg_us_rocprof" in d:
        # busy cycles summed over SIMDs / (1024 SIMDs x kernel cycles at the 2.1 GHz the part holds)
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["avg_us_rocprof"] * 2100.0)
out["_note"] = ("rocprofv3 --kernel-trace --stats and --pmc (one group per pass) over tools/pmc_stages.py; "
                "FETCH_SIZE is doubled (gfx950 reports half, calibrated in r01_hbm_traffic.json)")
json.dump(out, open(os.path.join(REPO, "profiles", rnd + "_stage_evidence.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
