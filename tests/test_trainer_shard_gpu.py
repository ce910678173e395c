"""SURVEY.md 8(e) on the REAL Trainer: a view-sharded 2-rank iteration must leave the gradient of the single-process
iteration over the same global batch (B = 2) in every parameter -- same Monte-Carlo samples (the sampler hashes the
GLOBAL view index), same jitter / texture / tangent noise, same eikonal samples (NoiseStream), the SDF-MLP rows split over
the ranks with an all-gather of sdf[N], the energy-ratio regulariser formed from all-reduced sums.  On the one-device GPU box the
processes share the device and talk over gloo (RCCL refuses two ranks on one device); the collectives are the same calls.  The `nccl`
parametrisations (one rank per device over RCCL / xGMI: all_gather_into_tensor, reduce_scatter_tensor, the flat all-reduce) switch themselves
on wherever >= 2 / >= 8 devices are visible, so the first multi-GPU box that runs this suite exercises RCCL (reference: the only thing
train_gshelltet_deepfashion.py:597-606 does is init_process_group("nccl"))."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

RES, HW, NS, IT = 16, (64, 64), 2, 1000      # tiny grid, steady-state schedule (sigma 2, shadow_scale 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _grads(trainer):
    return [None if p.grad is None else p.grad.detach().clone() for p in trainer.all_params()]


def _worker(rank, world, port, shard_rows, out_q, B=2, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":          # one rank per device
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from gshell_amd import workload
    from gshell_amd.render import render
    from gshell_amd.train import ViewShard
    shard = ViewShard(rank, world)
    tr = workload.build(res=RES, n_samples=NS, batch=B, train_res=HW, shard=shard, fit_steps=60, shard_mlp_rows=shard_rows)
    state = [p.detach().clone() for p in tr.all_params()]
    seed0 = render.rnd_seed
    render.rnd_seed = seed0 + 3            # the target renders draw Monte-Carlo samples too: same seed in both runs
    target = workload.make_targets(tr, shard.local_views(B), HW)
    render.rnd_seed = seed0 + 7
    tr.it = IT
    tr.forward_backward(target, global_batch=B)
    g_sharded = _grads(tr)
    result = None
    if rank == 0:      # the single-process iteration over both views, from the same parameters, seed and iteration
        single = workload.build(res=RES, n_samples=NS, batch=B, train_res=HW, shard=ViewShard(), fit_steps=0)
        with torch.no_grad():
            for p, v in zip(single.all_params(), state):
                p.copy_(v)
        single.lgt.update_pdf()
        render.rnd_seed = seed0 + 3
        t2 = workload.make_targets(single, list(range(B)), HW)
        assert torch.allclose(t2['img'][0], target['img'][0], atol=1e-5) and torch.equal(t2['background'][0], target['background'][0])
        render.rnd_seed = seed0 + 7
        single.it = IT
        single.forward_backward(t2)
        g_single = _grads(single)
        names = [n for n, _ in single.geometry.named_parameters()]
        rows = []
        for i, (a, b) in enumerate(zip(g_sharded, g_single)):
            if a is None or b is None:
                rows.append((i, None if a is None else float(a.abs().max()), None if b is None else float(b.abs().max()), 0.0))
                continue
            rel = float((a - b).norm() / b.norm().clamp_min(1e-20))
            rows.append((i, float(a.norm()), float(b.norm()), rel))
        result = (rows, names, tuple(tr.geometry.last_mesh_sizes), tuple(single.geometry.last_mesh_sizes))
    out_q.put((rank, result))
    dist.barrier()
    dist.destroy_process_group()


def _n_devices():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.parametrize("backend,world,B,shard_rows", [
    ("gloo", 2, 2, True), ("gloo", 2, 2, False), ("gloo", 8, 8, True),
    pytest.param("nccl", 2, 2, True, marks=pytest.mark.skipif(_n_devices() < 2, reason="RCCL needs one device per rank: < 2 devices here")),
    pytest.param("nccl", 2, 4, False, marks=pytest.mark.skipif(_n_devices() < 2, reason="RCCL needs one device per rank: < 2 devices here")),
    pytest.param("nccl", 8, 8, True, marks=pytest.mark.skipif(_n_devices() < 8, reason="RCCL needs one device per rank: < 8 devices here"))])
def test_sharded_iteration_equals_single_process(backend, world, B, shard_rows):
    """(8, 8, True) = the partitioning of BASELINE.json configs[3]: 8 ranks, global batch 8 (one view per rank), the grid rows
    (a count that 8 does not divide) split over the ranks, union visibility -- gloo: eight processes on the one device; nccl: one rank per
    device over RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard_rows, q, B, backend)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows, names, mesh_sharded, mesh_single = results[0]
    assert mesh_sharded == mesh_single and mesh_single[1] > 0, (mesh_sharded, mesh_single)
    print("parameter gradients (index, |sharded|, |single|, rel L2):", rows)
    for i, na, nb, rel in rows:
        assert na is not None and nb is not None or (na in (None, 0.0) and nb in (None, 0.0)), (i, na, nb)
        # 1e-4 relative (north_star); float-atomic accumulation order is the only difference left between the two runs
        assert rel <= 1e-4, f"parameter {i}: |g_sharded| {na} |g_single| {nb} rel L2 {rel}"


@pytest.mark.skipif(_n_devices() < 2, reason="bench.py --gpus 2 needs two devices")
def test_bench_runs_two_ranks_over_rccl():
    """`python bench.py --gpus 2 --steps 3`: bench.py re-launches itself under torch.distributed.run with backend nccl (= RCCL), one rank per
    device, and rank 0 prints the one JSON line with n_gpus = 2 -- so SCALE cannot be the first time the multi-GPU path executes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--res", "64", "--fit-steps", "60", "--early-steps", "0", "--extra-steps", "0", "--no-cpu-baseline"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["value"] > 0 and r["scaling"] == "weak"


def test_bench_multi_rank_plumbing_with_two_ranks_on_one_device():
    """What the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` exercises beyond a single-GPU run -- the launcher's environment,
    the process group, sharded views and SDF rows, the barrier + max-over-ranks timing, rank 0's ONE JSON line -- on the one-device box: GSHELL_BENCH_SAME_DEVICE=1
    puts both ranks on device 0 over gloo (RCCL refuses two ranks on one device).  The line says it is not a performance number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSHELL_BENCH_SAME_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--res", "64", "--fit-steps", "60", "--early-steps", "0", "--extra-steps", "0",
                          "--no-cpu-baseline", "--no-reference-config"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"{len(lines)} JSON lines (only rank 0 prints)"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0 and r["scaling"] == "weak" and "PLUMBING TEST" in r["data"]
    assert r["config"]["global_batch"] == 2 * r["config"]["views_per_gpu"]
