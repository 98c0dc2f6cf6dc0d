#!/usr/bin/env python3
"""Static instruction mix of the hot loop of every kernel in a gfx950 assembly file (hipcc -save-temps .s).

    python tools/probes/isa_count.py file.s [kernel-name-substring]

The hot loop of a kernel = the span of its LAST backward branch target .. that branch with the most instructions (the
frame loop of the probes and of the product kernels is by far the largest loop).  Prints per kernel: VGPRs / spills /
occupancy from the metadata, and for the loop the counts of packed-f32 ops, other VALU, transcendental, LDS, VMEM,
SALU, s_nop, s_waitcnt, MFMA, DPP / permlane, and a VALU-time estimate in cycles (probe-measured costs on MI355X:
VOP2/VOP1 2, packed f32 and 3-source ops 4, v_sqrt/v_rcp/... 8)."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_permlane"):
        return "permlane"
    if re.match(r"v_(sqrt|rsq|rcp|exp|log|sin|cos)_", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def valu_cycles(op, rest):
    c = classify(op)
    if c == "pk":
        return 4
    if c == "trans":
        return 8
    if c == "permlane":
        return 4
    if c == "valu":
        if re.match(r"v_(fma|mad|fmac|mac|lshl_add|add3|xad|and_or|or3|bfe|perm|cndmask)", op) and op.count("_e64") + 1:
            # 3-source VOP3 ops; v_fmac/v_cndmask in VOP2 form read two VGPRs + an implicit one
            if op.startswith(("v_fma_", "v_mad_", "v_lshl_add", "v_add3", "v_xad", "v_and_or", "v_or3", "v_bfe", "v_perm")):
                return 4
        return 2
    return 0


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(\w+):\s*;?\s*@?(\w*)", line)
        if m and line.startswith("_Z") and not line.startswith("."):
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur is not None:
            body.append(line)
            if line.strip().startswith(".end_amdhsa_kernel") or line.strip() == "s_endpgm":
                pass
    if cur:
        yield cur, body


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = dict(sgpr=int(m.group(2)), sspill=int(m.group(3)), vgpr=int(m.group(4)), vspill=int(m.group(5)))
    print("| kernel | vgpr | v-spill | s-spill | loop instrs | pk | valu | trans | dpp | permlane | mfma | lds | vmem | salu | s_nop | waitcnt | VALU cycles (est) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, body in kernels(text):
        if want and want not in name:
            continue
        labels, instrs = {}, []
        for line in body:
            s = line.strip()
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = len(instrs)
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            parts = s.split(None, 1)
            instrs.append((parts[0], parts[1] if len(parts) > 1 else ""))
        best = None
        for i, (op, rest) in enumerate(instrs):
            if op.startswith("s_cbranch") or op == "s_branch":
                t = rest.split()[0].rstrip(",") if rest else ""
                if t in labels and labels[t] <= i:
                    span = (labels[t], i + 1)
                    if best is None or span[1] - span[0] > best[1] - best[0]:
                        best = span
        if best is None:
            continue
        cnt, cyc, dpp = {}, 0, 0
        for op, rest in instrs[best[0]:best[1]]:
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
            cyc += valu_cycles(op, rest)
            if "dpp" in op or "quad_perm" in rest or "row_" in rest:
                dpp += 1
        md = meta.get(name, {})
        short = re.sub(r"^_Z\d+", "", name)[:48]
        print("| %s | %s | %s | %s | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
            short, md.get("vgpr", "?"), md.get("vspill", "?"), md.get("sspill", "?"), best[1] - best[0],
            cnt.get("pk", 0), cnt.get("valu", 0), cnt.get("trans", 0), dpp, cnt.get("permlane", 0), cnt.get("mfma", 0),
            cnt.get("lds", 0), cnt.get("vmem", 0), cnt.get("salu", 0), cnt.get("s_nop", 0), cnt.get("s_waitcnt", 0), cyc))


if __name__ == "__main__":
    main()
