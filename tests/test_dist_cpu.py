"""World-size-2 gloo test of the multi-GPU path's host logic (runs on CPU): contiguous batch
sharding, one-time broadcast of the filterbank from rank 0, optional all-gather of the outputs.
The per-rank compute is replaced by the oracle (no GPU here); the point is that
concat(rank outputs) == full-batch result and that constants become bit-identical."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO

from kapre_amd import dist as kdist


def test_shard_bounds_cover_the_batch_exactly():
    for n in (0, 1, 7, 8, 64, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [kdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        kdist.shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import torch
    import torch.distributed as dist
    import kapre_oracle as o
    import kapre_amd as kapre
    from kapre_amd import dist as kd

    r, w, _ = kd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    model = kapre.get_melspectrogram_layer(n_fft=512, hop_length=128, sample_rate=16000, n_mels=40)
    fb_layer = model.layers[2]
    if rank != 0:                      # corrupt the non-source copies: broadcast must repair them
        fb_layer.filterbank = np.zeros_like(fb_layer.filterbank)
    nbytes = kd.broadcast_constants(model, src=0)
    assert nbytes == 257 * 40 * 4
    want_fb = o.filterbank_mel(16000, 257, 40)
    assert np.array_equal(fb_layer.filterbank, want_fb)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (5, 3000, 1)).astype(np.float32)       # 5 items over 2 ranks: 3 + 2
    mine = kd.shard_batch(x, rank, world)
    assert mine.shape[0] == (3 if rank == 0 else 2)
    # a model that already holds fused-call plans and derived constants for its OLD filterbank must drop them
    # when the broadcast installs the received one: plans are keyed on the filterbank version
    import kapre_amd as kapre
    m2 = kapre.get_melspectrogram_layer(n_fft=512, hop_length=128, sample_rate=16000, n_mels=40)
    fbl = m2.layers[2]
    if rank != 0:
        fbl.filterbank = fbl.filterbank * np.float32(3.0)                 # this rank's copy differs before
    v0 = fbl._fb_version
    fbl._kranges = "stale"
    fbl._consts._cache["stale"] = object()
    kd.broadcast_constants(m2, src=0)
    assert fbl._fb_version == v0 + 1 and fbl._kranges is None and not fbl._consts._cache
    assert np.array_equal(fbl.filterbank, o.filterbank_mel(16000, 257, 40))
    y = o.kapre_melspectrogram(mine, n_fft=512, hop_length=128, sample_rate=16000, n_mels=40)
    full = kd.gather_batch(torch.from_numpy(y), world).numpy()
    want = o.kapre_melspectrogram(x, n_fft=512, hop_length=128, sample_rate=16000, n_mels=40)
    assert full.shape == want.shape and np.array_equal(full, want)
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmpdir, "ok%d" % rank), "w").write("ok")


def test_two_rank_gloo_shard_broadcast_gather(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
