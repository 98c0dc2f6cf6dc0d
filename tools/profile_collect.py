#!/usr/bin/env python3
"""Distil gpurun_out/prof_<round>/ (written by tools/profile_round.sh on the GPU box) into
profiles/: kernel statistics CSVs, per-counter CSV extracts for the mel kernel, the HBM traffic
JSON that bench.py reads, and a counter summary.  Usage: tools/profile_collect.py r01"""
import csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(REPO, "gpurun_out", "prof_" + rnd)
dst = os.path.join(REPO, "profiles")


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    # gpurun MERGES a call's output into the local gpurun_out/: a directory may hold several runs -- take the newest
    return max(hits, key=os.path.getmtime) if hits else None


def counter_rows(d):
    f = find(d, "counter_collection.csv")
    return list(csv.DictReader(open(f))) if f else []


def mean_counter(rows, kernel_sub, counter):
    v = [float(r["Counter_Value"]) for r in rows if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


sys.path.insert(0, REPO)
import bench  # noqa: E402  (workload names and the kernel each one is priced on)

KSUB = {"mel": "k_mel", "stft": "k_stft", "stftmag": "k_stft", "istft": "k_istft", "fb": "k_fb_pw", "mag": "k_cplx_to_real", "db": "k_db_log"}


def ksub_of(spec):
    """the kernel a workload is priced on (round 6: the stand-alone filterbank = k_fb_pw for mel banks, the MFMA kernel for the
    log-frequency bank)"""
    return "k_mel_ws" if spec["kind"] == "fb" and spec.get("bank") == "log" else KSUB[spec["kind"]]
import subprocess  # noqa: E402
try:
    COMMIT = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:  # noqa: BLE001
    COMMIT = ""
try:
    LIB_SHA = open(os.path.join(src, "lib_sha16.txt")).read().strip()
except OSError:
    LIB_SHA = ""
traffic = {}
for w, spec in bench.WORKLOADS.items():
    f = find(os.path.join(src, "stats_" + w), "kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (rnd, w)))
    f = os.path.join(src, "bench_under_rocprof_%s.json" % w)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_bench_under_rocprof_%s.json" % (rnd, w)))
    ksub = ksub_of(spec)
    fr = counter_rows(os.path.join(src, "pmc_FETCH_SIZE_" + w))
    wr = counter_rows(os.path.join(src, "pmc_WRITE_SIZE_" + w))
    for rows, c in ((fr, "FETCH_SIZE"), (wr, "WRITE_SIZE")):
        keep = [r for r in rows if ksub in r["Kernel_Name"] or "k_calib" in r["Kernel_Name"]]
        if keep:
            with open(os.path.join(dst, "%s_pmc_%s_%s.csv" % (rnd, c, w)), "w", newline="") as f:
                wri = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
                wri.writeheader(); wri.writerows(keep)
    fetch, n = mean_counter(fr, ksub, "FETCH_SIZE")
    calib, _ = mean_counter(fr, "k_calib_read8", "FETCH_SIZE")
    write, _ = mean_counter(wr, ksub, "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    corr = (1 << 20) / calib if calib else 2.0          # calibration kernel reads exactly 1 GiB
    traffic[w] = {"fetch_kib_raw": fetch, "fetch_correction": corr, "write_kib": write,
                  "hbm_bytes_per_launch": (fetch * corr + write) * 1024.0, "launches_averaged": n,
                  "kernel": ksub}
if traffic:
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB), mean over the launches of the "
                        "workload's dominant kernel. FETCH_SIZE is multiplied by the correction measured with the "
                        "known-traffic kernel k_calib_read8 (1 GiB read with the same 8-byte-per-lane access width): "
                        "the counter reports 1/2 on gfx950, as MI355X_MICROARCH.md says.")
    traffic["_commit"] = COMMIT            # the tree the passes were collected into (the binary: _lib_sha16)
    traffic["_lib_sha16"] = LIB_SHA
    json.dump(traffic, open(os.path.join(dst, rnd + "_hbm_traffic.json"), "w"), indent=1)

summary = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_sq_*"))):
    if not os.path.isdir(d):
        continue
    rows = counter_rows(d)
    for c in sorted({r["Counter_Name"] for r in rows}):
        v, n = mean_counter(rows, "k_mel", c)
        if v is not None:
            summary[c] = v
if summary:
    summary["_note"] = "mean per launch of the fused mel kernel on the target workload (batch 256), rocprofv3 --pmc, one small group per pass"
    json.dump(summary, open(os.path.join(dst, rnd + "_sq_counters_target.json"), "w"), indent=1)
# per-workload issue counters (tools/profile_round.sh step 4) -> profiles/<round>_sq_counters_<workload>.json (bench.py reads them)
for w, spec in bench.WORKLOADS.items():
    rows = counter_rows(os.path.join(src, "pmc_issue_" + w))
    ksub = ksub_of(spec)
    out = {}
    for c in sorted({r["Counter_Name"] for r in rows}):
        v, n = mean_counter(rows, ksub, c)
        if v is not None:
            out[c] = v
    if out:
        out["_note"] = "mean per launch of %s* on %s, rocprofv3 --pmc (one pass)" % (ksub, w)
        out["_commit"], out["_lib_sha16"] = COMMIT, LIB_SHA
        json.dump(out, open(os.path.join(dst, "%s_sq_counters_%s.json" % (rnd, w)), "w"), indent=1)
print(json.dumps({"traffic": traffic, "counters": summary}, indent=1))

# the logs of tools/end_of_round.sh (GPU suite, stand-alone fuzz, bench line, multi-process rehearsal) of the same binary
for name in ("pytest_gpu_final.log", "fuzz_seed0.log", "fuzz_seed17.log", "fuzz_seed23.log", "bench_line.json", "bench_full.json",
             "rehearsal_n2_one_gpu.log"):
    f = os.path.join(REPO, "gpurun_out", "%s_%s" % (rnd, name))
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (rnd, name)))
