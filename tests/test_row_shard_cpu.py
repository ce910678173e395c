"""Row-sharded SDF-network evaluation ("shard MLP rows across ranks + all-gather sdf[N]") on CPU, gloo,
world_size 2: values and gradients (parameters and inputs) must equal the single-process network."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gshell_amd.geometry.mlp import MLP, forward_row_sharded
from gshell_amd.train import ViewShard, flat_all_reduce_grads


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(3)
    return MLP(n_freq=2, d_hidden=16, n_hidden=3, skip_in=[1]).double()


def _problem():
    g = torch.Generator().manual_seed(5)
    x = torch.rand(37, 3, generator=g, dtype=torch.float64) - 0.5          # 37 rows: not divisible by the world size
    w = torch.randn(2, 37, generator=g, dtype=torch.float64)
    w[:, ::3] = 0.0                                                        # rows without upstream gradient (row-sparse path)
    return x, w


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = ViewShard(rank, world)
    net = _net()
    x, w = _problem()
    x.requires_grad_(True)
    y = forward_row_sharded(net, x, shard)
    (y[:, 0] * w[rank]).sum().backward()                                   # every rank has its OWN loss on the full sdf vector
    params = list(net.parameters()) + [x]
    flat_all_reduce_grads(params, shard)
    out_q.put((rank, y.detach().reshape(-1).tolist(), [p.grad.reshape(-1).tolist() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_forward_backward_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _net()
    x, w = _problem()
    x.requires_grad_(True)
    y = net(x)
    (y[:, 0] * (w[0] + w[1])).sum().backward()
    ref = [p.grad.reshape(-1) for p in list(net.parameters()) + [x]]
    for rank, yv, grads in results:
        assert torch.allclose(torch.tensor(yv, dtype=torch.float64), y.detach().reshape(-1), rtol=1e-12, atol=1e-12)
        for g, r in zip(grads, ref):
            assert torch.allclose(torch.tensor(g, dtype=torch.float64), r, rtol=1e-5, atol=1e-7)   # the flat gradient bucket is fp32


def test_collectives_single_process_identity():
    s = ViewShard()
    t = torch.arange(6.0)
    assert s.all_gather_rows(t) is t and s.reduce_scatter_sum(t) is t
