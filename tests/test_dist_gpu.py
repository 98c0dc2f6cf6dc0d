"""-m gpu: the RCCL path of kapre_amd/dist.py on the one MI355X a gpurun box has (the `backend="nccl"` branch
and a broadcast of a DEVICE tensor had never executed on the hardware, so the first real multi-GPU run could have failed for
trivial reasons).  A process group of world size 1 over backend "nccl" (= RCCL on ROCm) runs the same calls a rank of an N-GPU
job runs: communicator set-up, `broadcast_constants` (device tensors), `barrier`, `all_reduce(MAX)` (bench.py's max over ranks),
`all_gather_object`, `gather_batch` -- each of them a real RCCL launch.  Runs in a child process: a process group is global state.
The N > 1 protocol itself (shards, one broadcast, no steady-state collective) is covered by the world-size-2 gloo tests in
tests/test_dist_cpu.py.  Reference: none -- Kapre has no collectives; partition per SURVEY 8(e)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, json
sys.path.insert(0, os.environ["KPR_REPO"])
import numpy as np
import torch
import torch.distributed as dist
import kapre_amd as kapre
from kapre_amd import dist as kdist

rank, world, local = kdist.init_from_env(backend="nccl")                 # WORLD_SIZE=1: no group yet (a plain single-process run)
assert (rank, world, local) == (0, 1, 0) and not dist.is_initialized()
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
model = kapre.get_melspectrogram_layer(n_fft=1024, hop_length=256, sample_rate=22050, n_mels=64)
x = torch.rand(6, 22050, 1, device=dev) * 2 - 1
before = model(x).clone()                                                # called BEFORE the broadcast: plans must survive it
nbytes = kdist.broadcast_constants(model, src=0, device=dev)             # RCCL broadcast of a device tensor
assert nbytes == 513 * 64 * 4, nbytes
after = model(x)
assert torch.equal(before, after)                                        # the received filterbank is the sent one, bit for bit
dist.barrier()
t = torch.tensor([1.25, 3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                 # bench.py: max over ranks of (wall, device) time
assert t.tolist() == [1.25, 3.5]
names = [None]
dist.all_gather_object(names, "rank 0: " + torch.cuda.get_device_name(dev))
full = kdist.gather_batch(after, 1)                                      # all_gather of the (padded) shards
assert torch.equal(full, after)
lo, hi = kdist.shard_bounds(2048, 0, 1)
assert (lo, hi) == (0, 2048)
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"ok": True, "device": names[0], "broadcast_bytes": nbytes}))
'''


def test_rccl_path_at_world_size_one(tmp_path):
    env = dict(os.environ)
    env.update({"KPR_REPO": REPO, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD)
    p = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert '"ok": true' in p.stdout, p.stdout[-2000:]


def test_bench_n1_line_under_the_driver_command(tmp_path):
    """the driver's own N = 1 command with a short K: the last stdout line parses, is below 4 KB and carries roofline (with the
    rotating-buffer figures and the provenance of `traffic`) and cpu_baseline"""
    import json
    env = dict(os.environ)
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-also",
                        "--sustain", "0", "--settle", "0.2"], env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last) < 4000
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"].startswith("k_mel_pw<1024") and rf["kernel_us_rotating"] > 0
    assert line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["screened"]
