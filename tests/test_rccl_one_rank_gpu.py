"""RCCL executes on the one-device box too: a process group of ONE rank over backend `nccl` (= RCCL), through which every collective form the view-sharded
trainer issues (gshell_amd/train.py ViewShard: all_reduce SUM of the flat fp32 gradient bucket, all_reduce MAX of the float visibility flags,
all_gather_into_tensor of the SDF rows, reduce_scatter_tensor of their upstream gradient, broadcast of replicated parameters) runs on the device with the
tensor shapes / dtypes of the tet-res256 job.  With one rank every result is the identity -- what this proves is that the RCCL library of this image loads,
builds a communicator on an MI355X and accepts those calls; the multi-rank `nccl` parametrisations (tests/test_trainer_shard_gpu.py) need >= 2 devices.
Runs in a child process (the pytest process keeps no process-group state)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gshell_amd.train import ViewShard, flat_all_reduce_grads
sh = ViewShard(0, 1)
g = torch.Generator(device="cuda").manual_seed(0)
flat = torch.randn(23_000_000, device="cuda", generator=g)                    # ~ the 90 MB gradient bucket
ref = flat.clone()
dist.all_reduce(flat, op=dist.ReduceOp.SUM)
assert torch.equal(flat, ref)
flags = (torch.rand(225_000, device="cuda", generator=g) > 0.5).float()
ref = flags.clone()
dist.all_reduce(flags, op=dist.ReduceOp.MAX)
assert torch.equal(flags, ref)
rows = torch.randn(2_282_489, device="cuda", generator=g)                     # sdf of the tet-res256 grid
out = torch.empty_like(rows)
dist.all_gather_into_tensor(out, rows.contiguous())
assert torch.equal(out, rows)
out2 = torch.empty_like(rows)
dist.reduce_scatter_tensor(out2, rows.contiguous(), op=dist.ReduceOp.SUM)
assert torch.equal(out2, rows)
p = torch.nn.Parameter(torch.randn(1000, 3, device="cuda", generator=g))
ref = p.detach().clone()
dist.broadcast(p.data, src=0)
assert torch.equal(p.detach(), ref)
# the trainer's own helpers under an initialised nccl group
q = torch.nn.Parameter(torch.randn(4096, device="cuda", generator=g)); q.grad = torch.ones_like(q)
buf = flat_all_reduce_grads([q], sh)
assert sh.all_gather_rows(rows) is rows and sh.reduce_scatter_sum(rows) is rows
torch.cuda.synchronize()
print("rccl one-rank ok:", dist.get_backend(), torch.cuda.get_device_name(0))
dist.destroy_process_group()
'''


def test_every_collective_form_of_the_view_shard_runs_through_rccl_with_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, "-c", CHILD, root], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl one-rank ok: nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
