#!/usr/bin/env python3
"""Register / spill metadata of every kernel in libkapre_hip (CPU: hipcc cross-compiles kapre_hip.hip with -save-temps).
    python tools/probes/spill_report.py [path/to/existing.s]
Prints the kernels with VGPR spills, a private segment (scratch) or more than 16 SGPR spills, then totals."""
import glob, os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def metadata(text):
    out = []
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, flags=re.S):
        blk = m.group(0)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out.append(dict(name=name, vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), vspill=get("vgpr_spill_count"),
                        sspill=get("sgpr_spill_count"), scratch=get("private_segment_fixed_size")))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-DKPR_RING_DEPTH=3",
                            "-save-temps", "-c", os.path.join(REPO, "kapre_amd", "csrc", "kapre_hip.hip"), "-o", "k.o"], cwd=td, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            text = open(glob.glob(os.path.join(td, "*gfx950*.s"))[0]).read()
    md = metadata(text)
    names = demangle([k["name"] for k in md])
    bad = 0
    for k, n in zip(md, names):
        if k["vspill"] or k["scratch"] or k["sspill"] > 16:
            bad += 1
            print("%-110s vgpr %3d sgpr %3d vspill %3d sspill %3d scratch %4d" % (n[:110], k["vgpr"], k["sgpr"], k["vspill"], k["sspill"], k["scratch"]))
    print("%d kernels, %d flagged" % (len(md), bad))
