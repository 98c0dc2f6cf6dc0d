"""ISA audits of the whole device code (CPU tests: hipcc cross-compiles gfx950 without a GPU).

1. Wait states (round 5): hipcc's hazard recognizer does not look inside `asm` statements -- every packed-math primitive of
   kpr_fft.h is one -- so tools/hazard_scan.py checks the FINAL instruction stream of every kernel, whoever emitted each
   instruction, against the manual-wait-state table of the CDNA3/4 ISA (LLVM's gfx940 / gfx950 rules).
2. Wave-private LDS hand-overs (round 5): the `; kpr_lds_fence` markers of kpr_fft.h -- no LDS store between a load group's
   markers, no LDS load between a store group's, along every path of the control-flow graph.  This is the class of the
   round-4 wrong-lane bug of k_istft_pw<512, 2> (profiles/r05_hazard_rootcause.md).
3. The hand-pipelined loads of k_mel_ws (rounds 2-3): the MFMA phase issues its operand loads through inline asm and waits
   with counted `s_waitcnt`; hipcc does not model those loads, so between an asm load and its wait it may legally copy / spill
   the destination registers.  Nothing but the asm loads themselves and MFMAs may touch them inside the pipelined region.
4. No kernel uses scratch; register / spill budgets of the hot kernels.
"""
import glob
import os
import re
import subprocess
import tempfile

import sys

import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))
import hazard_scan  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as td:
        # device code only, straight to assembly (the flags of kapre_amd/build.py; half the time of a full -save-temps compile)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed",
                        "-DKPR_RING_DEPTH=3", "--cuda-device-only", "-S",
                        os.path.join(REPO, "kapre_amd", "csrc", "kapre_hip.hip"), "-o", "k.s"],
                       cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(os.path.join(td, "k.s")).read()


def _kernel_bodies(text, prefix):
    for m in re.finditer(r"^(%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % prefix, text, flags=re.S | re.M):
        yield m.group(1), m.group(2)


def _regs(tok):
    """'v[22:25]' -> {22,23,24,25}; 'v13' -> {13}"""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _audit_resident_paths(name, lines, first, drain, ring, dl, is_asm, wait_re):
    """The register-resident ring is fully unrolled with a run-time chunk count: every step (issue, counted wait,
    MFMAs) is its own basic block and hipcc is free to PLACE those blocks in any textual order (seen: steps 5 and
    6 swapped in the text, linked by branches in program order).  So the lgkmcnt queue is modelled along the
    control-flow path of the longest slice -- at every conditional branch the side whose next ring instruction is
    an issue is taken (both: the side a true "chunk exists" mask selects), which is the textual order whenever the
    blocks are laid out in program order -- instead of along the text.  The walk must see every issue exactly once."""
    label_at = {}
    for i in range(first, drain + 1):
        m = re.match(r"(\.LBB\d+_\d+):", lines[i].strip())
        if m:
            label_at[m.group(1)] = i

    def next_ring_is_issue(i, hops=0):
        """first ring instruction reached from line i (following fallthrough / unconditional branches): an issue?"""
        while first <= i <= drain and hops < 400:
            l = lines[i].strip()
            hops += 1
            if i in dl:
                return True
            if is_asm(i) and wait_re.match(l):
                return False
            br = re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if br:
                if br.group(1) not in label_at:
                    return False
                i = label_at[br.group(1)]
                continue
            i += 1
        return False

    lq, i, steps, issues = [], first, 0, 0
    while first <= i <= drain:
        steps += 1
        assert steps < 200000, name
        l = lines[i].strip()
        if not l or l.startswith(";") or re.match(r"\.LBB\d+_\d+:", l):
            i += 1
            continue
        br = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if br:
            target = label_at.get(br.group(2))
            if br.group(1) == "s_branch":
                if target is None:
                    break
                i = target
            elif target is None:
                i += 1
            else:
                t, f = next_ring_is_issue(target), next_ring_is_issue(i + 1)
                take = t and not f
                if t == f:
                    # both sides lead on: the branch tests a "this chunk exists" mask (vcc = exec & ~mask / exec & mask,
                    # set by the instruction before it) -- on the longest slice every such mask is true
                    prev = next(lines[k].strip() for k in range(i - 1, first, -1)
                                if lines[k].strip() and not lines[k].strip().startswith(";"))
                    if prev.startswith("s_andn2_b64 vcc, exec, s["):
                        take = br.group(1) == "s_cbranch_vccz"
                    elif prev.startswith("s_and_b64 vcc, exec, s["):
                        take = br.group(1) == "s_cbranch_vccnz"
                i = target if take else i + 1
            continue
        if i in dl:
            lq.append(_regs(re.split(r"[\s,]+", l)[1]))
            issues += 1
            i += 1
            continue
        w = wait_re.match(l)
        if w and is_asm(i):
            kl = int(w.group(2))
            lq = lq[len(lq) - kl:] if kl else []
            i += 1
            continue
        if i in ring:
            assert not l.startswith(("scratch_", "buffer_", "global_", "flat_")), \
                "%s: foreign VMEM op inside the counted-wait region: %s" % (name, l)
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", l)
            touched = set().union(*[_regs(t) for t in toks]) if toks else set()
            inflight = set().union(*lq) if lq else set()
            assert not (touched & inflight), "%s: in-flight register touched by: %s" % (name, l)
        i += 1
    assert issues == len(dl), "%s: the walk saw %d of %d issues" % (name, issues, len(dl))


def test_ws_consumer_ring_is_never_touched_in_flight(isa):
    """Same audit for k_mel_ws, whose consumer ring prefetches MFMA operands with inline asm: the streaming
    instances (RES = false) load BOTH operands -- global_load_dwordx4 (vmcnt queue) and ds_read2_b32 (lgkmcnt
    queue, LDS returns in order), counted wait `s_waitcnt vmcnt(N) lgkmcnt(M)` --, the register-resident
    instances (RES = true) only the magnitudes (ds_read2_b32, `s_waitcnt lgkmcnt(M)`, four sets).
    Compiler-emitted LDS / scalar-memory ops in the region would only make the lgkmcnt wait more conservative,
    but VMEM ops would break the vmcnt count, and no instruction may touch an in-flight destination register."""
    seen = 0
    wait_re = re.compile(r"s_waitcnt (?:vmcnt\((\d+)\) )?lgkmcnt\((\d+)\)$")
    for name, body in _kernel_bodies(isa, "_ZN3kpr8k_mel_wsILi"):
        resident = "Lb0ELb1ELb0EEE" in name             # <NC, FROM_MAG = false, RES = true, LD8 = false>
        lines = body.splitlines()
        is_asm = lambda i: i > 0 and "ASMSTART" in lines[i - 1]
        gl = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l and is_asm(i)]
        dl = [i for i, l in enumerate(lines) if "ds_read2_b32" in l and is_asm(i)]
        assert dl and (bool(gl) != resident), name
        first = min(gl[0], dl[0]) if gl else dl[0]
        drain_pat = r"s_waitcnt lgkmcnt\(0\)$" if resident else r"s_waitcnt vmcnt\(0\) lgkmcnt\(0\)"
        drain = next(i for i in range(max(gl[-1] if gl else 0, dl[-1]), len(lines))
                     if re.search(drain_pat, lines[i].strip()) and is_asm(i))
        dests = [re.split(r"[\s,]+", lines[i].strip())[1] for i in gl + dl]
        if resident:        # fully unrolled: every issue defines fresh values, hipcc may rename a set between chunks
            assert 8 * 4 <= len(set().union(*[_regs(d) for d in dests])) <= 8 * 6, name
        else:
            assert len(dests) == len(set(dests)), "%s: a register set has two issue points" % name
            assert len(set().union(*[_regs(d) for d in dests])) == (8 + 8) * 3, name   # 3 sets x (A 8 + B 8)
        # Basic blocks (split at labels) that hold ring instructions -- asm loads or counted waits.
        # Block placement may put other blocks (epilogue pieces, debug stamps) textually between
        # the loop and its drain although they execute after it; every block of the pipeline itself
        # contains ring instructions (one issue + one counted wait per chunk), so those are audited.
        starts = [i for i in range(first, drain + 1)
                  if re.match(r"(\.LBB\d+_\d+:|; %bb\.\d+)", lines[i].strip())]
        bounds = [first] + starts + [drain + 1]
        ring = set()
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            if any((i in gl or i in dl or (is_asm(i) and wait_re.match(lines[i].strip())))
                   for i in range(b0, b1)):
                ring.update(range(b0, b1))
        if resident:
            _audit_resident_paths(name, lines, first, drain, ring, set(dl), is_asm, wait_re)
            seen += 1
            continue
        vq, lq = [], []
        for walk in range(2):                           # the streaming ring is a loop: walk it twice
            for i in range(first, drain + 1):
                if i not in ring:
                    continue
                l = lines[i].strip()
                if not l or l.startswith(";"):
                    continue
                if i in gl:
                    vq.append(_regs(re.split(r"[\s,]+", l)[1])); continue
                if i in dl:
                    lq.append(_regs(re.split(r"[\s,]+", l)[1])); continue
                m = wait_re.match(l)
                if m and is_asm(i):
                    kl = int(m.group(2))
                    if m.group(1) is not None:
                        kv = int(m.group(1))
                        vq = vq[len(vq) - kv:] if kv else []
                    lq = lq[len(lq) - kl:] if kl else []
                    continue
                assert not l.startswith(("scratch_", "buffer_", "global_", "flat_")), \
                    "%s: foreign VMEM op inside the counted-wait region: %s" % (name, l)
                toks = re.findall(r"v\[\d+:\d+\]|v\d+", l)
                touched = set().union(*[_regs(t) for t in toks]) if toks else set()
                inflight = set().union(*(vq + lq)) if (vq or lq) else set()
                assert not (touched & inflight), "%s: in-flight register touched by: %s" % (name, l)
        seen += 1
    assert seen == 6          # n_fft 2048 and 1024 (FFT producers) x (resident, streaming), loader producers (4 and 8 loaders)


def _kernel_metadata(text):
    """(mangled name, vgpr spills, sgpr spills, private segment bytes) of every kernel in the code object's metadata"""
    out = []
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, flags=re.S):
        blk = m.group(0)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        out.append((re.search(r"\.name:\s+(\S+)", blk).group(1), get("vgpr_spill_count"), get("sgpr_spill_count"),
                    get("private_segment_fixed_size")))
    return out


# kernels allowed to use scratch: none (the 27 spilling instances of the mixed-radix ring ISTFT, VERDICT r03 item 4, were
# 4 PIN prefetch registers that counted as live around the whole loop + PIN 64-bit per-lane offsets: fixed in round 4)
KNOWN_SCRATCH = ()


def test_no_kernel_uses_scratch(isa):
    """'VGPR spills are 0 in every hot kernel' was only checked for the fused mel kernels.  Every kernel of the
    library: no VGPR spill, no private segment (KNOWN_SCRATCH, the list of exceptions, is empty since round 4)."""
    md = _kernel_metadata(isa)
    assert len(md) > 200
    bad = [(n, v, p) for n, v, s, p in md if (v or p) and not any(k in n for k in KNOWN_SCRATCH)]
    assert not bad, bad
    known = [(n, v, p) for n, v, s, p in md if (v or p) and any(k in n for k in KNOWN_SCRATCH)]
    assert len(known) == 0, len(known)


def test_per_wave_mel_kernel_budgets(isa):
    """k_mel_pw: four waves per SIMD (<= 128 VGPRs is enforced by its launch bounds: a spill would be the symptom), at most
    40 SGPR spills (v_writelane / v_readlane pairs on an issue-bound kernel; 60 for the n_fft 2048 PAIR instance), and no use of M0 (the ds_write_addtid
    variant that needed it was measured and parked: tools/probes/experiments/kpr_mel_pw_addtid.h.txt)."""
    md = [(n, v, s, p) for n, v, s, p in _kernel_metadata(isa) if "k_mel_pw" in n]
    # n_fft 256 ... 2048 x 4 / 8 / 16 waves per workgroup, + the PAIR form (three waves per SIMD) for n_fft 1024 and 2048
    assert len(md) == 14, len(md)
    assert sum("Lb1E" in n for n, _, _, _ in md) == 2
    for n, v, s_, p in md:
        # (the PAIR instance of n_fft 2048 carries the slot ring of the staged channels_last store since round 5: 36 -> 58 scalar
        #  spills, and 144 instead of 151 us on cfg3 -- tools/cl_stage_ab.py)
        assert v == 0 and p == 0 and s_ <= (60 if "ILi1024ELi12ELb1E" in n else 40), (n, v, s_, p)
    seen = 0
    for name, body in _kernel_bodies(isa, "_ZN3kpr8k_mel_pwILi"):
        assert not re.search(r"\bm0\b", body), name
        seen += 1
    assert seen == 14


def test_fb_pw_keeps_its_prefetch(isa):
    """k_fb_pw (stand-alone ApplyFilterbank, round 6): two rows in flight per wave only exist if hipcc's wait-count pass waits for
    each row with a COUNT (the other row's requests stay outstanding).  Three forms of the main loop compiled to s_waitcnt
    vmcnt(0) in front of every row -- an exit flag tested at the latch, `if (ticket valid)` around the requests, a skipped
    process() -- (profiles/r06_fb_pw.md); this pins the form that works: behind the main loop's header every vmcnt wait is
    either >= 4 (a row's own four dwordx4 requests + the Nyquist word, the other slot's stay in flight; >= 8 for the ST instances,
    whose unit is two rows) or the vmcnt(0) inside the cold dense-row loop (one per slot and channel).  No scratch."""
    seen = 0
    for name, body in _kernel_bodies(isa, "_ZN3kpr7k_fb_pwILi"):
        lines = body.splitlines()
        bar = next(i for i, l in enumerate(lines) if "s_barrier" in l)
        head = next(i for i in range(bar, len(lines)) if "Loop Header: Depth=1" in lines[i])
        counts = [int(m.group(1)) for l in lines[head:] for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
        stereo = "ELb1E" in name                                          # ST: a unit = two rows = eight requests + the Nyquist pair
        floor, dense_loops = (8, 4) if stereo else (4, 2)
        assert sum(c >= floor for c in counts) >= 2, (name, counts)
        assert all(c >= floor or c == 0 for c in counts), (name, counts)
        assert sum(c == 0 for c in counts) == dense_loops, (name, counts)   # the dense recomputation of a non-finite row, per slot (and channel)
        seen += 1
    assert seen == 8                                                     # 8 / 16 / 32 / 64 lanes per row x {contiguous, two interleaved channels}
    md = [(n, v, s_, p) for n, v, s_, p in _kernel_metadata(isa) if "k_fb_pw" in n]
    assert len(md) == 8 and all(v == 0 and p == 0 for _, v, _, p in md), md


def test_fused_kernels_do_not_spill(isa):
    for kernel in ("k_mel_ws", "k_mel_ts", "k_mel_pw", "k_stft", "k_stft3", "k_irfft"):
        blocks = re.findall(r"\.name:\s+_ZN3kpr\d+%sILi\d+E.*?\.vgpr_spill_count:\s+(\d+)" % kernel,
                            isa, flags=re.S)
        assert blocks, kernel
        limit = 0
        assert all(int(b) <= limit for b in blocks), (kernel, blocks)


# ---- round 5: wait states and LDS hand-over fences, every kernel -------------------------------------------------------------
SYNTH = """
	.text
	.p2align	8
	.type	k_synth,@function
k_synth:
; %bb.0:
	v_pk_add_f32 v[2:3], v[4:5], v[6:7]
	v_mov_b32_dpp v8, v2 row_shr:1 row_mask:0xf bank_mask:0xf
	v_add_f32_e32 v9, v1, v1
	v_readfirstlane_b32 s4, v9
	v_cmp_gt_u32_e32 vcc, 16, v1
	s_nop 0
	v_cndmask_b32_e32 v10, v1, v2, vcc
	v_sqrt_f32_e32 v11, v1
	v_add_f32_e32 v12, v11, v1
	v_readfirstlane_b32 s6, v1
	s_nop 2
	global_load_dword v13, v1, s[6:7]
	v_mov_b32_e32 v20, v1
	s_nop 0
	v_permlane32_swap_b32_e32 v20, v21
	v_mov_b32_e32 v30, v1
	v_mfma_f32_16x16x4_f32 a[0:3], v30, v31, a[0:3]
	global_store_dwordx4 v[40:41], v[42:45], off
	v_mov_b32_e32 v43, v1
	v_readlane_b32 s9, v60, 3
	;;#ASMSTART
	v_pk_mul_f32 v[70:71], v[72:73], s[8:9] op_sel_hi:[1,0]
	;;#ASMEND
.LBB0_1:
	v_mov_b32_dpp v50, v51 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
	s_nop 3
	v_mov_b32_e32 v51, v1
	s_cbranch_scc1 .LBB0_1
	; kpr_lds_fence W
	ds_write_b32 v1, v2
	s_cbranch_scc1 .LBB0_3
.LBB0_2:
	; kpr_lds_fence R
	ds_read_b32 v3, v1
	ds_bpermute_b32 v4, v1, v3
	; kpr_lds_fence X
	ds_write_b32 v1, v2 offset:4
	s_endpgm
.LBB0_3:
	ds_read_b32 v5, v1 offset:8
	s_branch .LBB0_2
.Lfunc_end0:
"""


def test_hazard_scanner_positive_controls():
    """every rule of tools/hazard_scan.py fires on a hand-written stream that violates it (incl. the loop back edge and the
    asm-consumer case found in k_stft_big<2>); the fence checker follows the control-flow graph, not the text"""
    findings, nk = hazard_scan.scan_text(SYNTH)
    assert nk == 1
    rules = sorted(f.rule for f in findings)
    assert rules == ["DPP", "DPP", "MFA", "PLS", "RDL", "SGM", "SGV", "SGV", "STD", "TRN"], rules
    assert any(f.rule == "SGV" and f.casm and "v_readlane" in f.producer for f in findings)
    bad, nmark, kernels = hazard_scan.check_lds_fences(SYNTH)
    assert nmark == 3 and len(kernels) == 1
    # the load in the out-of-line block .LBB0_3 is reached with the W region open; nothing else is misplaced
    assert [b[4] for b in bad] == ["ds_read_b32 v5, v1 offset:8"], bad


def test_round4_failing_stream_has_no_wait_state_hazard():
    """the instruction stream of k_istft_pw<512, 2> that produced wrong lanes in round 4 (tools/probes/hazard/): no table
    hazard -- what was wrong with it is the ORDER of two LDS instructions (profiles/r05_hazard_rootcause.md)"""
    text = open(os.path.join(REPO, "tools", "probes", "hazard", "ipw512_2_failing.s.txt")).read()
    findings, nk = hazard_scan.scan_text(text)
    assert nk == 1 and not findings
    lds = [l.split()[0] for l in text.splitlines() if l.strip().startswith(("ds_read_b32", "ds_write2_b32"))]
    i = next(k for k, l in enumerate(text.splitlines()) if "ds_write2_b32 v51, v55, v75 offset1:1" in l)
    assert "ds_read_b32 v84, v21 offset:1980" in text.splitlines()[i + 1]       # the hoisted store sits above the last load
    assert lds


def test_no_wait_state_hazards_in_any_kernel(isa):
    findings, nk = hazard_scan.scan_text(isa)
    assert nk > 150
    assert not findings, [(f.kernel[:60], f.rule, f.producer, f.consumer) for f in findings[:10]]


def test_lds_handover_fences_in_every_kernel(isa):
    bad, nmark, kernels = hazard_scan.check_lds_fences(isa)
    assert not bad, bad[:10]
    assert nmark > 1000
    # every kernel family that hands words from lane to lane through a wave-private row carries the markers
    for fam in ("k_stftILi", "k_stft3ILi", "k_stft_mrI", "k_stft_bsILi", "k_stft_bigILi", "k_mel_pwILi", "k_fb_pwILi", "k_mel_wsILi", "k_mel_tsILi",
                "k_mel_mrI", "k_irfftILi", "k_irfft_mrI", "k_irfft_bsILi", "k_irfft_bigILi", "k_istft_fusedILi", "k_istft_wsILi",
                "k_istft_ws_mrI", "k_istft_pwILi"):
        assert any(fam in k for k in kernels), fam
