"""RCCL on the hardware before the driver's 8-GPU run: one process, world_size 1, backend "nccl" (= RCCL on ROCm) with the GPU as
its device -- the process-group initialisation `bench.py --gpus N` performs (`init_process_group("nccl", device_id=dev)`), the
device-side `all_gather_into_tensor` of `bench.py --gather`, and `shard.all_gather_units` / `shard.flash_attention_sharded(...,
gather=True)` through the HIP operator.  World size 1 is all a one-GPU box offers; it still loads librccl, creates the communicator
and launches the collective's kernel on the stream (no data-path collective exists in this path: SURVEY.md section 8(e)).
Runs in a child process so that the process group never outlives the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
g = torch.Generator(device=dev).manual_seed(3)
B, QH, KH, L, E = 2, 4, 2, 384, 64
q = torch.randn(B, QH, L, E, generator=g, device=dev).to(torch.bfloat16)
k = torch.randn(B, KH, L, E, generator=g, device=dev).to(torch.bfloat16)
v = torch.randn(B, KH, L, E, generator=g, device=dev).to(torch.bfloat16)
o = pkg.flash_attention(q, k, v, causal=True)
# bench.py --gather (weak): all_gather_into_tensor of O on the device
full = torch.empty((1,) + tuple(o.shape), dtype=o.dtype, device=dev)
dist.all_gather_into_tensor(full, o)
torch.cuda.synchronize()
assert torch.equal(full[0], o)
# shard.all_gather_units and the sharded operator with its one optional collective
units = o.reshape(B * KH, QH // KH, L, E)
got = pkg.shard.all_gather_units(units, B * KH)
assert torch.equal(got, units)
full2 = pkg.shard.flash_attention_sharded(q, k, v, causal=True, gather=True)
torch.cuda.synchronize()
assert torch.equal(full2, o)
# bench.py --gather, overlapped form: chunk i's gather on RCCL's stream beside chunk i+1's forward (shard.forward_with_overlapped_gather)
o2 = torch.empty_like(q); ms2 = torch.empty(B, QH, L, dtype=q.dtype, device=dev); ls2 = torch.empty_like(ms2)
sl = [slice(0, 1), slice(1, 2)]
o_ch = [o2[c] for c in sl]
full_ch = [torch.empty_like(x) for x in o_ch]                 # world size 1: [1 * n, ...]
pkg.shard.forward_with_overlapped_gather(lambda i: pkg.fa_fwd_into(o_ch[i], ms2[sl[i]], ls2[sl[i]], q[sl[i]], k[sl[i]], v[sl[i]], causal=True), o_ch, full_ch)
torch.cuda.synchronize()
assert torch.equal(torch.cat(full_ch, dim=0), o)
t = torch.tensor([1.5, 2.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the timing reduction of bench.py
dist.barrier()
torch.cuda.synchronize()
assert t.tolist() == [1.5, 2.5]
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_nccl_backend_world_size_1_runs_the_device_side_collectives():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
