"""Multi-GPU execution: one process per GPU, batch sharding, RCCL only for the constants.

The hot path shards naturally: every (batch item, channel) signal is
independent and the decibel maximum is taken inside one batch item
(/root/reference/kapre/backend.py:178-192), so ranks never exchange data in steady state.
The only shared state is the filterbank (n_freq x n_mels float32, 525 KB at 1025 x 128) and the
window(s); ``broadcast_constants`` sends them once from rank 0 with ``torch.distributed.broadcast``
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests) so that all ranks use
bit-identical constants.  No all-reduce, no all-gather unless the caller asks for the full batch.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def init_from_env(backend: str = "nccl"):
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world,
    local_rank); a single-process run needs no process group and returns (0, 1, 0)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-process path on a box with ONE GPU (tools/rehearse_multi_gpu.sh): every rank
    # on device 0 and the process group on gloo -- RCCL refuses two ranks on one device
    backend = os.environ.get("KAPRE_AMD_DIST_BACKEND", backend)
    if os.environ.get("KAPRE_AMD_SHARE_DEVICE") == "1":
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of the batch axis: the first (n_items % world) ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %r/%r" % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, rank: int, world: int):
    """This rank's slice of a (batch, ...) array / tensor (a view, no copy)."""
    lo, hi = shard_bounds(int(x.shape[0]), rank, world)
    return x[lo:hi]


def _constant_arrays(model) -> List[Tuple[object, str]]:
    """(layer, attribute) pairs of the host-built constants of a Kapre model."""
    from .keras_shim import Sequential

    layers = model._flat_layers() if isinstance(model, Sequential) else list(getattr(model, "layers", [model]))
    found = []
    for layer in layers:
        if isinstance(getattr(layer, "filterbank", None), np.ndarray):
            found.append((layer, "filterbank"))
    return found


def broadcast_constants(model, src: int = 0, device=None) -> int:
    """Broadcast every filterbank of ``model`` from rank ``src`` (one collective per matrix) and
    install the received copy on each rank.  Returns the number of bytes broadcast.  With a
    single process (no process group) this is a no-op."""
    import torch
    import torch.distributed as dist

    # (a process group of ONE rank still runs the collective: that is how tests/test_dist_gpu.py drives the RCCL path on a
    #  one-GPU box; a plain single-process run has no process group and returns here)
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    total = 0
    for layer, attr in _constant_arrays(model):
        host = np.ascontiguousarray(getattr(layer, attr), dtype=np.float32)
        t = torch.from_numpy(host.copy())
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=src)
        # ApplyFilterbank.filterbank is a property: assigning bumps the layer's filterbank version,
        # which drops its device / packed copies and k-ranges AND invalidates every fused-call plan
        # keyed on it (a model called before the broadcast is correct after it)
        setattr(layer, attr, t.cpu().numpy())
        total += host.nbytes
    return total


def gather_batch(y, world: int):
    """Optional: all-gather the per-rank outputs into the full batch on every rank (only when a
    caller needs a single-device result; outputs stay sharded by default)."""
    import torch
    import torch.distributed as dist

    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return y
    sizes = [torch.zeros(1, dtype=torch.int64, device=y.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([y.shape[0]], dtype=torch.int64, device=y.device))
    n_max = int(max(int(s) for s in sizes))
    pad = torch.zeros((n_max,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
    pad[: y.shape[0]] = y
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: int(s)] for p, s in zip(parts, sizes)], dim=0)
