#!/usr/bin/env python3
"""Cut ONE kernel out of a `hipcc -S` device assembly (42 MB for kapre_hip.hip) into a stand-alone assembly file that
clang + ld.lld turn into a loadable code object: its text section, its kernel descriptor, its `.set` symbols and its entry of
the metadata note.  python extract_kernel.py FULL.s MANGLED_NAME OUT.s"""
import re
import sys


def extract(text, k):
    lines = text.splitlines()
    out = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6"]
    # text: from the .section line in front of `.protected k` (or `.globl k`) to .Lfunc_endN + .size
    i0 = next(i for i, l in enumerate(lines) if re.match(r"^\s*\.(protected|globl)\s+%s\b" % re.escape(k), l))
    s = i0
    while not lines[s].lstrip().startswith((".section", ".text")):
        s -= 1
    e = next(i for i in range(i0, len(lines)) if re.match(r"^\.Lfunc_end\d+:", lines[i]))
    out += lines[s:e + 1]
    out += [l for l in lines[e + 1:e + 4] if l.lstrip().startswith(".size")]
    # (the kernel descriptor -- .amdhsa_kernel ... .end_amdhsa_kernel -- sits between s_endpgm and .Lfunc_end: already copied)
    assert any(".end_amdhsa_kernel" in l for l in lines[s:e])
    out += [l for l in lines if re.match(r"^\s*\.set\s+%s\." % re.escape(k), l)]
    # metadata
    m0 = next(i for i, l in enumerate(lines) if l.strip() == ".amdgpu_metadata")
    m1 = next(i for i in range(m0, len(lines)) if l_is(lines[i], ".end_amdgpu_metadata"))
    meta = lines[m0:m1 + 1]
    ks = next(i for i, l in enumerate(meta) if l.strip() == "amdhsa.kernels:")
    starts = [i for i in range(ks + 1, len(meta)) if meta[i].startswith("  - ")]
    tail = next(i for i in range(ks + 1, len(meta)) if re.match(r"^[a-z]", meta[i]))        # amdhsa.target / version
    starts.append(tail)
    for a, b in zip(starts, starts[1:]):
        if any(re.match(r"^\s+\.name:\s+%s\s*$" % re.escape(k), l) for l in meta[a:b]):
            out += meta[:ks + 1] + meta[a:b] + meta[tail:]
            break
    else:
        raise SystemExit("kernel not in metadata")
    return "\n".join(out) + "\n"


def l_is(l, s):
    return l.strip() == s


if __name__ == "__main__":
    open(sys.argv[3], "w").write(extract(open(sys.argv[1]).read(), sys.argv[2]))
