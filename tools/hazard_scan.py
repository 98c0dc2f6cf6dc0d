#!/usr/bin/env python3
"""Static wait-state audit of gfx950 ISA (the `hipcc -S` text of the device code), origin-agnostic.

Why: hipcc's hazard recognizer treats an `asm` statement as one opaque instruction -- an inline-asm `v_pk_*` is neither seen
as the VALU *producer* of a register nor as a VALU *consumer* of one (LLVM `GCNHazardRecognizer`: `isVALU(INLINEASM)` is
false and an INLINEASM counts zero wait states; outside the string hipcc only adds a fixed one-state pad).  Every FFT
primitive of kpr_fft.h is such a statement, so the pairs below are correct only if the FINAL instruction stream happens to
satisfy them.  This tool checks the final stream itself: every instruction of every kernel, whoever emitted it, against the
manually-inserted-wait-state table of the CDNA3/CDNA4 ISA (MI300 ISA guide section 4.5 "Manually Inserted Wait States";
LLVM GCNHazardRecognizer.cpp for gfx940 / gfx950):

  id   producer                                  consumer                                               states
  DPP  VALU writes VGPR                          VALU with DPP reads that VGPR                          2
  DPX  VALU writes EXEC                          VALU with DPP                                          5
  RDL  VALU writes VGPR                          v_readlane / v_readfirstlane reads it as src0          1
  RWX  VALU writes EXEC                          v_readlane / v_readfirstlane / v_writelane             4
  LSL  VALU writes SGPR / VCC                    v_readlane / v_writelane using it as lane select       4
  SGV  VALU writes SGPR / VCC                    VALU reads that SGPR / VCC (gfx90a+)                   2
  SGM  VALU writes SGPR                          VMEM (buffer / global / flat / scratch) reads it       5
  TRN  trans VALU (exp/log/rcp/rsq/sqrt/sin/cos) writes VGPR   non-trans VALU reads it (gfx940+)        1
  PLS  VALU writes VGPR                          v_permlane16_swap / v_permlane32_swap reads it (gfx950) 2
  MFA  non-MFMA VALU writes VGPR                 v_mfma_* reads it as A / B / C (gfx90a+)               2
  DFM  VALU writes VCC                           v_div_fmas                                             4
  EXZ  VALU writes VCC / EXEC                    VALU reading vccz / execz as data                      5  (never emitted; listed)
  STD  global / flat / buffer store of > 64 bit  VALU writes the store's data VGPRs (gfx940: 2)         2
  M0S  SALU writes M0                            LDS add-tid / GDS / s_sendmsg / *_lds DMA / s_moverel  1

A wait state = one issued instruction of the wave (`s_nop N` = N + 1).  The walk is over the control-flow graph of the
text: a consumer's window is followed backwards through fallthrough edges and through every branch that targets the
consumer's block, so a producer at the end of a loop body and a consumer at its top are paired.

Usage:  python tools/hazard_scan.py k.s [--kernel REGEX] [--rules DPP,RDL,...] [--verbose]
Importable: scan_text(text, kernel_regex=None) -> list of Finding.
"""
import re
import sys
from collections import namedtuple

Finding = namedtuple("Finding", "kernel rule need have producer consumer pline cline pasm casm")

TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|legacy_f32)|^v_(exp|log)_legacy_f32")
DPP_TOK = re.compile(r"\b(quad_perm:|row_shl:|row_shr:|row_ror:|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|"
                     r"row_bcast:|row_newbcast:|row_share:|row_xmask:|dpp8:)")
TWO_DST = re.compile(r"^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co)_u32|^v_div_scale_|^v_mad_(u64_u32|i64_i32)")
CMP = re.compile(r"^v_cmpx?_")
RW_BOTH = re.compile(r"^v_(swap_b32|permlane16_swap|permlane32_swap)")
VMEM = re.compile(r"^(buffer|global|flat|scratch|tbuffer|image)_")
M0_CONS = re.compile(r"^(ds_\w*addtid|ds_gws|s_sendmsg|s_movrel|\w+_lds_|buffer_load_\w+ .*\blds\b)")


def _reg_set(tok):
    """register token -> set of ('v'|'s'|'a', n) plus the named ones ('vcc', 'exec', 'm0')"""
    tok = tok.strip()
    tok = re.sub(r"^(-|\|)+|\|$", "", tok)                 # neg / abs source modifiers
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), n) for n in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([vsa])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {("vcc", 0)}
    if tok in ("exec", "exec_lo", "exec_hi"):
        return {("exec", 0)}
    if tok == "m0":
        return {("m0", 0)}
    return set()


class Ins:
    __slots__ = ("op", "ops", "text", "line", "asm", "defs", "uses", "states", "is_valu", "is_dpp", "is_trans", "src0")

    def __init__(self, text, line, in_asm):
        self.text, self.line, self.asm = text, line, in_asm
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands: comma separated; modifiers (op_sel:[..], offset:..) follow the last operand after whitespace
        ops, depth, cur = [], 0, ""
        for ch in rest:
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur)
        ops = [o.strip() for o in ops]
        if ops:                                          # strip trailing modifiers of the last operand
            ops[-1] = ops[-1].split()[0] if ops[-1].split() else ops[-1]
        self.ops = ops
        op = self.op
        self.is_valu = op.startswith("v_") and not op.startswith("v_nop")
        self.is_dpp = bool(self.is_valu and (op.endswith("_dpp") or DPP_TOK.search(body)))
        self.is_trans = bool(TRANS.match(op))
        self.states = 1
        if op == "s_nop":
            self.states = int(ops[0], 0) + 1 if ops else 1
        self.defs, self.uses, self.src0 = set(), set(), set()
        regs = [_reg_set(o) for o in ops]
        if self.is_valu:
            if CMP.match(op):                             # compares print their destination (vcc / an SGPR pair) first
                self.defs |= regs[0] if regs else {("vcc", 0)}
                if op.startswith("v_cmpx"):
                    self.defs |= {("exec", 0)}
                for r in regs[1:]:
                    self.uses |= r
                if regs[1:]:
                    self.src0 = regs[1]
            elif RW_BOTH.match(op):
                for r in regs[:2]:
                    self.defs |= r
                    self.uses |= r
            else:
                nd = 2 if TWO_DST.match(op) else 1
                for r in regs[:nd]:
                    self.defs |= r
                for r in regs[nd:]:
                    self.uses |= r
                if regs[nd:]:
                    self.src0 = regs[nd]
                if op.startswith("v_cmpx"):
                    self.defs |= {("exec", 0)}
                if op.startswith("v_div_fmas") or re.match(r"^v_(cndmask_b32|addc_co_u32|subb_co_u32|subbrev_co_u32)_e32", op):
                    self.uses |= {("vcc", 0)}
                # read-modify-write destinations: fmac / mac / accumulating dot products, v_writelane, v_movrel
                if re.match(r"^v_(fmac|mac|pk_fmac|dot\w+c|writelane)", op):
                    self.uses |= regs[0] if regs else set()
        elif op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_endpgm",
                                                        "s_sleep", "s_setprio", "s_sendmsg", "s_cmp", "s_bitcmp", "s_store",
                                                        "s_dcache", "s_icache", "s_setreg", "s_sethalt", "s_trap", "s_code_end")):
            if regs:
                self.defs |= regs[0]
            for r in regs[1:]:
                self.uses |= r
        else:                                             # memory: loads define, everything is a use
            is_store = "store" in op or op.startswith(("ds_write", "ds_add", "ds_sub", "ds_min", "ds_max", "ds_or", "ds_and"))
            if not is_store and regs and (VMEM.match(op) or op.startswith("ds_")):
                self.defs |= regs[0]
                for r in regs[1:]:
                    self.uses |= r
            else:
                for r in regs:
                    self.uses |= r


def parse_kernels(text):
    """yield (name, [Ins | ('label', name)]) for every kernel of a `hipcc -S` device assembly"""
    lines = text.splitlines()
    i, n = 0, len(lines)
    while i < n:
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", lines[i])
        if m and i + 1 < n and not lines[i].startswith(".L") and (i == 0 or ".type" in "".join(lines[max(0, i - 6):i])
                                                                    or ".p2align" in "".join(lines[max(0, i - 6):i])):
            name = m.group(1)
            body, in_asm = [], False
            j = i + 1
            while j < n and not lines[j].lstrip().startswith(".Lfunc_end") and not lines[j].lstrip().startswith(".section"):
                l = lines[j].strip()
                if "ASMSTART" in l:
                    in_asm = True
                elif "ASMEND" in l:
                    in_asm = False
                elif re.match(r"^\.LBB\d+_\d+:", l):
                    body.append(("label", l.split(":")[0]))
                elif l.startswith("; kpr_lds_fence "):
                    body.append(("fence", l.split()[2], j + 1))
                elif l and not l.startswith((";", ".", "//")) and re.match(r"^[a-z]", l):
                    body.append(Ins(l, j + 1, in_asm))
                j += 1
            if any(isinstance(b, Ins) and b.op == "s_endpgm" for b in body):
                yield name, body
            i = j
        else:
            i += 1


def _preds(body):
    """index -> list of predecessor indices (instructions only; labels are transparent)"""
    label_pos = {b[1]: k for k, b in enumerate(body) if not isinstance(b, Ins)}
    preds = {k: [] for k in range(len(body))}
    prev = None
    for k, b in enumerate(body):
        if isinstance(b, Ins):
            if prev is not None:
                preds[k].append(prev)
            prev = None if b.op in ("s_branch", "s_endpgm", "s_setpc_b64") else k
            if b.op.startswith(("s_branch", "s_cbranch")) and b.ops and b.ops[0] in label_pos:
                preds.setdefault(("L", label_pos[b.ops[0]]), []).append(k)
        else:
            # a label: the next instruction inherits fallthrough `prev` and the branches that target this label
            pass
    # attach branch sources to the first instruction after their label
    for key, srcs in list(preds.items()):
        if isinstance(key, tuple):
            k = key[1] + 1
            while k < len(body) and not isinstance(body[k], Ins):
                k += 1
            if k < len(body):
                preds[k].extend(srcs)
            del preds[key]
    return preds


def _walk_back(body, preds, k, need, hit):
    """shortest wait-state distance (< need) from instruction k back to an instruction for which hit(ins) holds, else None"""
    best = None
    stack = [(p, 0) for p in preds[k]]
    seen = {}
    while stack:
        p, dist = stack.pop()
        if dist >= need or seen.get(p, 1 << 30) <= dist:
            continue
        seen[p] = dist
        ins = body[p]
        if hit(ins):
            if best is None or dist < best[0]:
                best = (dist, ins)
            continue
        for q in preds[p]:
            stack.append((q, dist + ins.states))
    return best


def scan_kernel(name, body, rules=None):
    preds = _preds(body)
    out = []

    def check(rule, k, need, hit):
        if rules and rule not in rules:
            return
        b = _walk_back(body, preds, k, need, hit)
        if b is not None:
            out.append(Finding(name, rule, need, b[0], b[1].text, body[k].text, b[1].line, body[k].line, b[1].asm, body[k].asm))

    for k, c in enumerate(body):
        if not isinstance(c, Ins):
            continue
        vuse = {r for r in c.uses if r[0] == "v"}
        suse = {r for r in c.uses if r[0] in ("s", "vcc")}
        if c.is_valu:
            if c.is_dpp:
                check("DPP", k, 2, lambda p: p.is_valu and p.defs & vuse)
                check("DPX", k, 5, lambda p: p.is_valu and ("exec", 0) in p.defs)
            if c.op.startswith(("v_readlane", "v_readfirstlane")):
                s0 = {r for r in c.src0 if r[0] == "v"}
                check("RDL", k, 1, lambda p: p.is_valu and p.defs & s0)
            if c.op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                check("RWX", k, 4, lambda p: p.is_valu and ("exec", 0) in p.defs)
            if c.op.startswith(("v_readlane", "v_writelane")):
                sel = _reg_set(c.ops[2]) if len(c.ops) > 2 else set()
                sel = {r for r in sel if r[0] in ("s", "vcc")}
                if sel:
                    check("LSL", k, 4, lambda p: p.is_valu and p.defs & sel)
            if suse:
                check("SGV", k, 2, lambda p: p.is_valu and p.defs & suse)
            if vuse and not c.is_trans:
                check("TRN", k, 1, lambda p: p.is_trans and p.defs & vuse)
            if c.op.startswith(("v_permlane16_swap", "v_permlane32_swap")):
                check("PLS", k, 2, lambda p: p.is_valu and p.defs & vuse)
            if c.op.startswith(("v_mfma", "v_smfma")):
                check("MFA", k, 2, lambda p: p.is_valu and not p.op.startswith(("v_mfma", "v_smfma")) and p.defs & vuse)
            if c.op.startswith("v_div_fmas"):
                check("DFM", k, 4, lambda p: p.is_valu and ("vcc", 0) in p.defs)
            vdef = {r for r in c.defs if r[0] == "v"}
            if vdef:
                def wide_store(p):
                    if not (VMEM.match(p.op) and "store" in p.op and re.search(r"x[34]\b", p.op)):
                        return False
                    data = set()
                    for o in p.ops:
                        r = _reg_set(o)
                        if len(r) > 2 and all(x[0] == "v" for x in r):
                            data |= r
                    return bool(data & vdef)
                check("STD", k, 2, wide_store)
        elif VMEM.match(c.op):
            sreg = {r for r in c.uses if r[0] == "s"}
            if sreg:
                check("SGM", k, 5, lambda p: p.is_valu and p.defs & sreg)
        if M0_CONS.match(c.text):
            check("M0S", k, 1, lambda p: p.op.startswith("s_") and ("m0", 0) in p.defs)
    return out


LDS_MEM_R = re.compile(r"^ds_read|^ds_load")
LDS_MEM_W = re.compile(r"^ds_write|^ds_store")


def check_lds_fences(text, kernel_regex=None):
    """The `; kpr_lds_fence W|R|X` markers of kpr_fft.h (wave-private LDS hand-overs): along every path of a kernel's
    control-flow graph, no LDS load between a W marker and the next marker, no LDS store between an R marker and the next
    marker.  (hipcc lays basic blocks out in any order -- cold blocks behind the loop, the latch in front of the body --, so the
    state is propagated along fallthrough and branch edges, not along the text.  ds_bpermute / ds_swizzle / LDS atomics touch
    no hand-over words and are ignored.)  Returns (violations, number of markers, kernels with markers)."""
    bad, nmark, kernels = [], 0, set()
    for name, body in parse_kernels(text):
        if kernel_regex and not re.search(kernel_regex, name):
            continue
        marks = [b for b in body if not isinstance(b, Ins) and b[0] == "fence"]
        if not marks:
            continue
        nmark += len(marks)
        kernels.add(name)
        n = len(body)
        label_pos = {b[1]: k for k, b in enumerate(body) if not isinstance(b, Ins) and b[0] == "label"}
        succ = [[] for _ in range(n)]
        for k, b in enumerate(body):
            if isinstance(b, Ins):
                if b.op not in ("s_branch", "s_endpgm", "s_setpc_b64") and k + 1 < n:
                    succ[k].append(k + 1)
                if b.op.startswith(("s_branch", "s_cbranch")) and b.ops and b.ops[0] in label_pos:
                    succ[k].append(label_pos[b.ops[0]])
            elif k + 1 < n:
                succ[k].append(k + 1)
        state = [set() for _ in range(n)]                 # modes that may be open when node k is reached
        state[0].add(("X", 0))
        work = [0]
        while work:
            k = work.pop()
            b = body[k]
            out = state[k]
            if not isinstance(b, Ins) and b[0] == "fence":
                out = {(b[1], b[2])}
            for q in succ[k]:
                if not out <= state[q]:
                    state[q] |= out
                    work.append(q)
        for k, b in enumerate(body):
            if not isinstance(b, Ins) or not b.op.startswith("ds_"):
                continue
            for mode, ml in state[k]:
                if (mode == "W" and LDS_MEM_R.match(b.op)) or (mode == "R" and LDS_MEM_W.match(b.op)):
                    bad.append((name, mode, ml, b.line, b.text))
    return bad, nmark, kernels


def scan_text(text, kernel_regex=None, rules=None):
    findings, nk = [], 0
    for name, body in parse_kernels(text):
        if kernel_regex and not re.search(kernel_regex, name):
            continue
        nk += 1
        findings.extend(scan_kernel(name, body, rules))
    return findings, nk


def main(argv):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--rules", default=None)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args(argv)
    rules = set(a.rules.split(",")) if a.rules else None
    findings, nk = scan_text(open(a.asm).read(), a.kernel, rules)
    by = {}
    for f in findings:
        by.setdefault((f.rule, f.pasm, f.casm), []).append(f)
    print("%d kernels scanned, %d findings" % (nk, len(findings)))
    for (rule, pasm, casm), fs in sorted(by.items()):
        print("  %s  producer %s, consumer %s: %d in %d kernels" % (rule, "asm" if pasm else "hipcc", "asm" if casm else "hipcc",
                                                                     len(fs), len({f.kernel for f in fs})))
    bad, nmark, kernels = check_lds_fences(open(a.asm).read(), a.kernel)
    print("LDS hand-over fences: %d markers in %d kernels, %d misplaced LDS accesses" % (nmark, len(kernels), len(bad)))
    for k, mode, ml, ln, l in bad[:40]:
        print("  %s: region %s opened at line %d contains line %d: %s" % (k, mode, ml, ln, l))
    if a.verbose:
        for f in findings:
            print("%s\n  %s needs %d has %d\n    L%d %s%s\n    L%d %s%s" % (f.kernel, f.rule, f.need, f.have, f.pline, f.producer,
                  "  [asm]" if f.pasm else "", f.cline, f.consumer, "  [asm]" if f.casm else ""))
    return 1 if (findings or bad) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
