"""The slot ring of k_mel_pw's staged channels_last store (kapre_amd/csrc/kpr_mel_pw_kernels.h), as a model.

The kernel's protocol, step by step, under RANDOM interleavings of the waves of a workgroup: tickets (channel pairs) are drawn
in increasing order; a wave waits until the slot of its block has seen off the block `slots` positions back (generation word),
writes its two columns, bumps the slot's arrival counter, and the wave that finds C / 2 - 1 there copies the block out, zeroes the
counter and bumps the generation.  Checked for every schedule: no deadlock, every block stored exactly once, complete, made of
its own columns only -- for ring sizes down to ONE slot (the launcher never goes below four) and more waves than slots.
Runs on the CPU; the kernel itself is pinned by tests/test_gpu_parity.py::test_mel_pw_pair_staged_channels_last_store."""
import random

import pytest


def simulate(n_waves, cp, slots, n_blocks, rng, max_steps=200000):
    tickets = n_blocks * cp
    counter = n_waves                           # the LDS ticket counter (the first ticket of every wave is static)
    cnt = [0] * slots                           # arrivals per slot
    done = [0] * slots                          # blocks each slot has seen off
    content = [dict() for _ in range(slots)]    # slot -> {pair index: block}
    stored = {}
    # wave state: (phase, ticket); phases: wait -> write -> arrive -> [store -> release] -> draw
    waves = [["wait", w] for w in range(n_waves)]
    for step in range(max_steps):
        runnable = []
        for w, (phase, t) in enumerate(waves):
            if phase == "idle":
                continue
            if phase == "wait":
                blk = t // cp
                if t < tickets and done[blk % slots] < blk // slots:
                    continue                    # spinning on the generation word
            runnable.append(w)
        if not runnable:
            if all(p == "idle" for p, _ in waves):
                return stored
            raise AssertionError("deadlock: %r cnt %r done %r" % (waves, cnt, done))
        w = rng.choice(runnable)
        phase, t = waves[w]
        if t >= tickets:
            waves[w] = ["idle", t]
            continue
        blk, pair = divmod(t, cp)
        s, gen = blk % slots, blk // slots
        if phase == "wait":
            assert done[s] == gen, "a later generation of the slot left before this block arrived"
            waves[w][0] = "write"
        elif phase == "write":
            assert all(b == blk for b in content[s].values()), ("columns of two blocks in one slot", content[s], blk)
            assert pair not in content[s]
            content[s][pair] = blk
            waves[w][0] = "arrive"
        elif phase == "arrive":
            old = cnt[s]
            cnt[s] += 1
            assert old < cp
            waves[w][0] = "store" if old == cp - 1 else "draw"
        elif phase == "store":
            assert blk not in stored, "block stored twice"
            assert sorted(content[s]) == list(range(cp)) and set(content[s].values()) == {blk}, (content[s], blk)
            stored[blk] = dict(content[s])
            waves[w][0] = "release"
        elif phase == "release":
            content[s].clear()
            cnt[s] = 0
            done[s] = gen + 1
            waves[w][0] = "draw"
        elif phase == "draw":
            waves[w] = ["wait", counter]
            counter += 1
    raise AssertionError("no progress in %d steps" % max_steps)


@pytest.mark.parametrize("n_waves,cp,slots,n_blocks", [
    (12, 3, 16, 42),        # cfg3: six channels, twelve waves, 42 blocks per workgroup
    (12, 3, 4, 42),         # the smallest ring the launcher accepts
    (12, 2, 4, 50),
    (12, 4, 8, 30),         # eight channels
    (12, 3, 1, 20),         # one slot: everything serialises, nothing may break
    (12, 8, 2, 9),          # sixteen channels, two slots
    (3, 3, 4, 11),          # fewer waves than a block has pairs
    (12, 3, 16, 2),         # fewer tickets than waves
])
def test_slot_ring_protocol_under_random_schedules(n_waves, cp, slots, n_blocks):
    for seed in range(40):
        stored = simulate(n_waves, cp, slots, n_blocks, random.Random(1000 * seed + n_blocks))
        assert sorted(stored) == list(range(n_blocks))


def test_a_ring_without_the_generation_word_is_caught_by_the_model():
    """positive control: drop the wait and some schedule mixes two blocks in a slot"""
    def broken(n_waves, cp, slots, n_blocks, rng):
        tickets = n_blocks * cp
        counter = n_waves
        content = [dict() for _ in range(slots)]
        cnt = [0] * slots
        waves = [["write", w] for w in range(n_waves)]
        for _ in range(100000):
            live = [w for w, (p, t) in enumerate(waves) if p != "idle"]
            if not live:
                return
            w = rng.choice(live)
            phase, t = waves[w]
            if t >= tickets:
                waves[w][0] = "idle"
                continue
            blk, pair = divmod(t, cp)
            s = blk % slots
            if phase == "write":
                assert all(b == blk for b in content[s].values()), "mixed"
                content[s][pair] = blk
                waves[w][0] = "arrive"
            elif phase == "arrive":
                cnt[s] += 1
                if cnt[s] == cp:
                    content[s].clear()
                    cnt[s] = 0
                waves[w] = ["write", counter]
                counter += 1

    with pytest.raises(AssertionError, match="mixed"):
        for seed in range(200):
            broken(12, 3, 2, 30, random.Random(seed))
