#!/usr/bin/env python3
"""Live-VGPR profile of one kernel from gfx950 assembly (hipcc -S / -save-temps): where is the register pressure?

    python tools/probes/vgpr_live.py file.s kernel-substring [bucket]

Backward liveness over the kernel's control-flow graph on the FINAL register assignment.  Every `bucket` (default 60)
instructions one line: max live VGPRs in the bucket, and the landmark opcodes seen there (barriers, MFMA, LDS exchange
stores, global loads / stores, sqrt, bpermute) so that the buckets can be matched to source phases.  Approximations:
a write under a partial EXEC mask is treated as a full kill; SDWA / DPP-masked / v_writelane destinations count as
read-modify-write."""
import re
import sys

RE_V = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in RE_V.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(rest):
    # operands separated by commas outside brackets
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def def_use(op, rest):
    ops = split_ops(rest.split(";")[0])
    allr = [regs(o) for o in ops]
    if not ops:
        return set(), set()
    nodef = op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "flat_store", "v_cmp", "v_readlane",
                           "v_readfirstlane", "s_", "ds_gws", "global_atomic", "ds_add", "ds_max", "ds_min", "buffer_wbl2",
                           "buffer_inv", "v_nop"))
    if op.startswith(("global_atomic", "ds_add_rtn", "ds_max_rtn", "ds_min_rtn", "ds_add_u32")) and "rtn" in op or \
            (op.startswith("global_atomic") and " sc0" in rest):
        nodef = False          # returning atomics write their first operand
    if nodef:
        u = set()
        for r in allr:
            u |= r
        return set(), u
    d = set(allr[0])
    u = set()
    for r in allr[1:]:
        u |= r
    if op.startswith(("v_permlane16_swap", "v_permlane32_swap")):
        d |= allr[1] if len(allr) > 1 else set()
        u |= d
    rmw = op.startswith(("v_writelane", "v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_movrel")) or "_sdwa" in op or \
        (("_dpp" in op or "quad_perm" in rest or "row_" in rest) and ("bank_mask:0xf" not in rest or "row_mask:0xf" not in rest)) or \
        "bound_ctrl" not in rest and "_dpp" in op
    if rmw:
        u |= d
    return d, u


LANDMARKS = ("s_barrier", "v_mfma", "ds_write_b128", "ds_write2_b32", "ds_read_b128", "global_load_dwordx2", "global_load_dwordx4",
             "global_load_dword ", "global_store", "v_sqrt", "ds_bpermute", "scratch_", "v_log", "global_atomic")


def main():
    path, want = sys.argv[1], sys.argv[2]            # [bucket [AT]]
    bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
    body, labels = [], {}
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section"):
            break
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(body)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        parts = s.split(None, 1)
        body.append((parts[0], parts[1] if len(parts) > 1 else ""))
    n = len(body)
    succ = [[] for _ in range(n)]
    for i, (op, rest) in enumerate(body):
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            t = rest.split()[0]
            if t in labels:
                succ[i].append(labels[t])
            continue
        if op.startswith("s_cbranch"):
            t = rest.split()[0]
            if t in labels and labels[t] < n:
                succ[i].append(labels[t])
        if i + 1 < n:
            succ[i].append(i + 1)
    du = [def_use(op, rest) for op, rest in body]
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            d, u = du[i]
            new = (out - d) | u
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    print("instructions %d, max live VGPRs %d" % (n, max(len(s) for s in live_in)))
    for b in range(0, n, bucket):
        seg = range(b, min(n, b + bucket))
        mx = max(len(live_in[i]) for i in seg)
        marks = []
        for i in seg:
            for k in LANDMARKS:
                if body[i][0].startswith(k.strip()) and k.strip() not in marks:
                    marks.append(k.strip())
        print("%5d  live max %3d  %s" % (b, mx, " ".join(marks)))
    if len(sys.argv) > 4:                       # AT: the live registers at instruction AT with their reaching definitions
        at = int(sys.argv[4])
        print("live at %d (%s %s):" % (at, body[at][0], body[at][1]))
        for r in sorted(live_in[at]):
            j = at - 1
            while j >= 0 and r not in du[j][0]:
                j -= 1
            print("  v%-3d <- %5d  %s %s" % (r, j, body[j][0] if j >= 0 else "?", body[j][1][:70] if j >= 0 else ""))


if __name__ == "__main__":
    main()
