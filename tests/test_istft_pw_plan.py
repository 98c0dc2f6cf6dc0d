"""CPU emulation of k_istft_pw's work split and overlap-add (kapre_amd/csrc/kpr_istft_pw_kernels.h): the same items, segments
(with their halo frames), frame runs per stream, head / tail kinds, stashes, flags order and two-party sums -- executed stream by
stream in numpy on random frames and compared with a plain overlap-add (tf.signal.overlap_and_add: ascending frame order).

What it pins without a GPU:
  * every output sample is written as a FINAL value exactly once (and partial values only where a predecessor completes them);
  * interior hop blocks are bit-identical to the sequential sum, boundary blocks differ by association only;
  * the interleaved instances' stream -> (frame run, channel) mapping and predecessor / successor arithmetic;
  * the segment bounds the host passes as quotient / remainder.
The GPU tests (test_gpu_parity.py::test_istft_per_wave_kernel*) check the kernel itself against the oracle."""
import numpy as np
import pytest

FINAL, PARTIAL, DISCARD, RMW = 0, 1, 2, 3


def emulate(frames, hop, n_fft, segs, nstr, n_stash, t_out):
    """frames: (C, F, n_fft) float32 windowed frames of ONE batch item (C = 1: a contiguous signal; C > 1: the interleaved
    instances, streams = (run, channel)).  Returns (out (C, t_out), writes (C, t_out) = FINAL writes per sample)."""
    C, F, _ = frames.shape
    R = n_fft // hop
    cs = C                                                   # streams between a run and its successor
    runs = nstr // C
    out = np.full((C, t_out), np.nan, np.float32)
    writes = np.zeros((C, t_out), np.int32)
    q, r = divmod(F, segs)

    def store(c, t0, vals, final):
        n = max(0, min(len(vals), t_out - t0))
        out[c, t0:t0 + n] = vals[:n]
        if final:
            writes[c, t0:t0 + n] += 1

    for seg in range(segs):
        f0, f1 = seg * q + min(seg, r), (seg + 1) * q + min(seg + 1, r)
        fa = max(0, f0 - (R - 1))
        n = f1 - fa
        base, rem = divmod(n, runs)
        assert base >= R - 1, "the plan guarantees R - 1 frames per run"
        stash, partial_in_waveform, tails = {}, set(), {}
        bounds = {}
        for sid in range(nstr):                              # phase 1: every stream walks its run
            rr, c = sid // C, sid % C
            ra = fa + rr * base + min(rr, rem)
            rb = ra + base + (1 if rr < rem else 0)
            bounds[sid] = (ra, rb)
            head = FINAL if ra == 0 else (DISCARD if rr == 0 else PARTIAL)
            acc = np.zeros(n_fft, np.float32)
            for f in range(ra, rb):
                acc = acc + frames[c, f]                     # ascending frames: the reference's order
                j = f - ra
                kind = head if j < R - 1 else FINAL
                block = acc[:hop].copy()
                if kind == PARTIAL and sid - cs < n_stash:
                    stash[(sid, j)] = block
                elif kind == PARTIAL:
                    store(c, f * hop, block, final=False)
                    partial_in_waveform.add((sid, j))
                elif kind == FINAL:
                    store(c, f * hop, block, final=True)
                acc = np.concatenate([acc[hop:], np.zeros(hop, np.float32)])
            tails[sid] = acc[:n_fft - hop].copy()
        for sid in range(nstr):                              # phase 2: the tails (after the successor's flag)
            rr, c = sid // C, sid % C
            ra, rb = bounds[sid]
            kind = FINAL if rb == F else (DISCARD if rr == runs - 1 else RMW)
            if kind == FINAL:
                store(c, rb * hop, tails[sid], final=True)
            elif kind == RMW:
                succ = sid + cs
                for j in range(R - 1):
                    t0 = (rb + j) * hop
                    part = stash[(succ, j)] if (succ, j) in stash else out[c, t0:t0 + hop].copy()
                    assert (succ, j) in stash or (succ, j) in partial_in_waveform
                    store(c, t0, tails[sid][j * hop:(j + 1) * hop] + part, final=True)      # (earlier frames) + (later frames)
    return out, writes


def plain_ola(frames, hop, t_out):
    C, F, n_fft = frames.shape
    out = np.zeros((C, max(t_out, (F - 1) * hop + n_fft)), np.float32)
    for f in range(F):                                       # ascending: tf.signal.overlap_and_add
        out[:, f * hop:f * hop + n_fft] += frames[:, f]
    return out[:, :t_out]


@pytest.mark.parametrize("F, n_fft, S, segs, nstr, C, n_stash, win", [
    (431, 1024, 4, 2, 32, 1, 27, 1024),      # cfg4: two segments per signal, 27 of 31 streams with a stash
    (431, 1024, 4, 4, 32, 1, 0, 1024),       # no stash at all: every partial block through the waveform
    (100, 1024, 4, 1, 32, 1, 31, 800),       # runs of exactly R - 1 = 3 frames and a few of 4; win < n_fft
    (230, 1024, 2, 1, 32, 1, 23, 1024),      # hop = n_fft / 8: every block is a two-party sum
    (64, 1024, 8, 2, 32, 1, 31, 1023),       # hop = n_fft / 2, odd window (odd signal length)
    (50, 2048, 4, 1, 16, 1, 13, 2048),
    (777, 512, 4, 3, 64, 1, 58, 400),
    (434, 1024, 4, 9, 32, 2, 27, 1024),      # interleaved stereo: 16 runs x 2 channels, nine segments
    (97, 1024, 4, 1, 32, 4, 20, 800),        # four channels: eight runs
    (64, 2048, 8, 4, 16, 8, 8, 2048),        # eight channels, two runs per item
    (300, 512, 4, 2, 64, 2, 62, 512),
])
def test_work_split_and_two_party_sums(F, n_fft, S, segs, nstr, C, n_stash, win):
    hop = n_fft * S // 16
    R = n_fft // hop
    rng = np.random.default_rng(F + n_fft + C)
    frames = rng.standard_normal((C, F, n_fft)).astype(np.float32)
    frames[:, :, win:] = 0.0                                 # the synthesis window is zero beyond win_length
    t_out = (F - 1) * hop + win
    q = F // segs
    assert q >= (nstr // C) * (R - 1), "the host only makes such plans (launch_istft_pw: segs_max)"
    out, writes = emulate(frames, hop, n_fft, segs, nstr, n_stash, t_out)
    assert (writes == 1).all(), "every sample must be written as a final value exactly once"
    want = plain_ola(frames, hop, t_out)
    scale = np.abs(want).max()
    assert np.abs(out - want).max() <= 4e-7 * scale
    assert (out == want).mean() > (0.3 if R - 1 < (q // (nstr // C)) else 0.0)     # interior blocks: same sums, same order


def test_segment_bounds_cover_every_frame_once():
    """segment j = [j q + min(j, r), (j + 1) q + min(j + 1, r)) with F = segs q + r (IstftPwPlan.seg_q / seg_r)"""
    for F in (1, 7, 96, 431, 434, 1000, 4097):
        for segs in (1, 2, 3, 4, 9, 64):
            if segs > F:
                continue
            q, r = divmod(F, segs)
            b = [(j * q + min(j, r), (j + 1) * q + min(j + 1, r)) for j in range(segs)]
            assert b[0][0] == 0 and b[-1][1] == F
            assert all(b[j][1] == b[j + 1][0] for j in range(segs - 1))
            assert all(q <= hi - lo <= q + 1 for lo, hi in b)
