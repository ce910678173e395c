"""View-sharded data parallelism (host logic) on CPU with the gloo backend, world_size 2:
round-robin view assignment, loss re-weighting and the single flat-bucket gradient all-reduce must reproduce the
single-process gradient of  mean_over_views(per-view loss) + view-independent terms."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gshell_amd.train import ViewShard, flat_all_reduce_grads, sharded_total_loss


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy(params, views):
    """per-view 'image' loss (mean over the given views) and a view-independent regulariser"""
    a, b = params
    per_view = torch.stack([((a * (v + 1)).sin() * b.sum()).pow(2).mean() for v in views]).mean()
    return per_view, (a.pow(2).sum() + b.abs().sum()) * 0.1


def _worker(rank, world, port, B, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = ViewShard(rank, world)
    g = torch.Generator().manual_seed(0)
    a = torch.randn(7, generator=g, requires_grad=True)
    b = torch.randn(3, 2, generator=g, requires_grad=True)
    unused = torch.zeros(4, requires_grad=True)            # a parameter that gets no gradient on any rank
    views = shard.local_views(B)
    per_view, glob = _toy((a, b), views)
    sharded_total_loss(per_view, glob, len(views), B, world).backward()
    flat_all_reduce_grads([a, b, unused], shard)
    flags = torch.zeros(6, dtype=torch.int32)
    flags[rank::3] = 1
    shard.all_reduce_max(flags)                            # union of per-rank visibility flags
    rep = torch.full((3,), float(rank + 1))
    shard.broadcast([rep])                                 # replicas start from rank 0's values
    assert rep.tolist() == [1.0, 1.0, 1.0]
    # plain lists: tensors would travel as shared-memory fds, which race with this process exiting
    out_q.put((rank, views, a.grad.tolist(), b.grad.tolist(), unused.grad.tolist(), flags.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_sharded_gradient_equals_single_process(B):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    a = torch.randn(7, generator=g, requires_grad=True)
    b = torch.randn(3, 2, generator=g, requires_grad=True)
    per_view, glob = _toy((a, b), list(range(B)))
    (per_view + glob).backward()
    assert sorted(results[0][1] + results[1][1]) == list(range(B))               # every view rendered exactly once
    for rank, views, ga, gb, gu, flags in results:
        ga, gb, gu = torch.tensor(ga), torch.tensor(gb), torch.tensor(gu)
        assert torch.allclose(ga, a.grad, rtol=1e-5, atol=1e-6) and torch.allclose(gb, b.grad, rtol=1e-5, atol=1e-6)
        assert (gu == 0).all()
        assert flags == [1, 1, 0, 1, 1, 0]


def test_view_shard_single_process_is_identity():
    s = ViewShard()
    assert s.local_views(4) == [0, 1, 2, 3]
    t = torch.ones(3)
    assert s.all_reduce_sum(t) is t and s.all_reduce_max(t) is t


def test_noise_stream_is_fresh_per_draw_rank_consistent_and_survives_a_render_outside_the_training_batch():
    """render.NoiseStream: (i) a second draw of the same name within an iteration differs from the first (the reference draws from
    the global RNG, render.py:55, :68, :265), (ii) re-entering the iteration repeats the sequence, (iii) rank r of W sees rows
    r, r + W, ... of the single-process draw, (iv) a render of one local view after step(global_batch=G) (validate()) works."""
    from gshell_amd.render.render import NoiseStream
    single = NoiseStream(seed=5)
    single.set_iteration(7, global_batch=4)
    a1 = single.normal('jitter', 1.0, (4, 3, 3, 2), 'cpu')
    a2 = single.normal('jitter', 1.0, (4, 3, 3, 2), 'cpu')
    assert not torch.equal(a1, a2)
    single.set_iteration(7, global_batch=4)
    assert torch.equal(single.normal('jitter', 1.0, (4, 3, 3, 2), 'cpu'), a1)
    assert torch.equal(single.normal('jitter', 1.0, (4, 3, 3, 2), 'cpu'), a2)
    for rank in range(2):
        ns = NoiseStream(seed=5, rank=rank, world=2)
        ns.set_iteration(7, global_batch=4)
        assert torch.equal(ns.normal('jitter', 1.0, (2, 3, 3, 2), 'cpu'), a1[rank::2])
        assert torch.equal(ns.normal('jitter', 1.0, (2, 3, 3, 2), 'cpu'), a2[rank::2])
        one = ns.normal('texture', 0.01, (1, 3, 3, 3), 'cpu')          # validate(): one local view although global_batch = 4
        assert one.shape == (1, 3, 3, 3)
