"""Hand-written gradient path of the SDF network (csrc/mlp_h2.hip: save_fwd -> bwd chain -> wgrad) against torch autograd of
the SAME network in float64 (reference formulation: geometry/mlp.py:32-40 under autograd, and the eikonal term's
autograd.grad(create_graph=True) + second backward, geometry/gshell_tets_geometry.py:302-324).  Tolerance 1e-4 relative
(north_star) -- the kernels sit two orders below it."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(n_hidden=6, skip_in=(3,), seed=0):
    from gshell_amd.geometry.mlp import MLP
    torch.manual_seed(seed)
    net = MLP(skip_in=list(skip_in), n_freq=6, n_hidden=n_hidden, d_hidden=256)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    return net.to(DEV)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


@pytest.mark.parametrize("n_hidden,skip_in,need_x,wgrad_fp32", [(6, (3,), True, False), (6, (3,), False, True), (2, (), True, False)])
def test_row_sparse_backward_matches_float64_autograd(n_hidden, skip_in, need_x, wgrad_fp32, monkeypatch):
    from gshell_amd.geometry import mlp as mlp_mod
    from gshell_amd.geometry.mlp import row_sparse_backward, row_sparse_backward_torch, split_param_grads
    monkeypatch.setattr(mlp_mod, "SDF_MLP_WGRAD_FP32", wgrad_fp32)       # bf16-pair weight gradients (default) / the exact-fp32 MFMA version
    net = _net(n_hidden, skip_in)
    g = torch.Generator(device=DEV).manual_seed(1)
    N = 5000 + 13
    x = (torch.rand(N, 3, device=DEV, generator=g) * 1.4 - 0.7).contiguous()
    gy = torch.zeros(N, 1, device=DEV)
    idx = torch.arange(0, N, 3, device=DEV)
    # upstream gradients spanning six decades (means over 10^6 pixels give 1e-8 .. 1e-2 in training)
    gy[idx, 0] = torch.randn(idx.numel(), device=DEV, generator=g) * torch.pow(10.0, torch.rand(idx.numel(), device=DEV, generator=g) * 6 - 8)
    g_x, flat = row_sparse_backward(net, x, gy, need_x)
    grads = split_param_grads(net, flat)
    net64 = copy.deepcopy(net).double()
    x64 = x.double().requires_grad_(True)
    ref = torch.autograd.grad(net64(x64), [x64] + list(net64.parameters()), gy.double())
    if need_x:
        assert _rel(g_x, ref[0]) < 1e-4
        rows_without = torch.ones(N, dtype=torch.bool, device=DEV)
        rows_without[idx] = False
        assert float(g_x[rows_without].abs().max()) == 0.0
    else:
        assert g_x is None
    print("  row-sparse backward, relative L2 error vs float64 per parameter (wgrad fp32 = %s): " % wgrad_fp32
          + " ".join("%.1e" % _rel(a, b) for a, b in zip(grads, ref[1:])))
    for (name, _), a, b in zip(net.named_parameters(), grads, ref[1:]):
        assert a.shape == b.shape
        assert _rel(a, b) < 1e-4, (name, _rel(a, b))
    # and the fp32 torch formulation it replaces is no closer to float64 than a factor of a few
    _, flat_t = row_sparse_backward_torch(net, x, gy, False)
    grads_t = split_param_grads(net, flat_t)
    worst_hip = max(_rel(a, b) for a, b in zip(grads, ref[1:]))
    worst_torch = max(_rel(a, b) for a, b in zip(grads_t, ref[1:]))
    assert worst_hip < max(20 * worst_torch, 2e-6 if wgrad_fp32 else 3e-5), (worst_hip, worst_torch)


@pytest.mark.parametrize("formulation,n,n_hidden,skip_in", [("reverse", 1000 + 7, 6, (3,)), ("reverse", 16, 6, (3,)), ("reverse", 1, 6, (3,)), ("reverse", 129, 2, ()),
                                                             ("reverse-fp32-wgrad", 300, 6, (3,)), ("tangent", 1000 + 7, 6, (3,)), ("tangent", 16, 6, (3,))])
def test_eikonal_term_matches_float64_double_backward(formulation, n, n_hidden, skip_in, monkeypatch, request):
    """both formulations of the term (reverse over reverse = the reference's; forward-mode tangent rows) against float64 autograd"""
    from gshell_amd.geometry import mlp as M
    from gshell_amd.geometry.mlp import eikonal_sq_sum
    if formulation.endswith("-fp32-wgrad"):          # the exact-fp32 weight-gradient kernel over the [zbar; delta] x [a; u] planes
        monkeypatch.setattr(M, "SDF_MLP_WGRAD_FP32", True)
        formulation = formulation.split("-")[0]
    monkeypatch.setattr(M, "EIKONAL_FORMULATION", formulation)
    if formulation == "tangent":          # the forward-mode formulation's <EIK> kernels are oracle kernels (lib/variants/oracles.so)
        from gshell_amd import _lib
        import os
        if not os.path.isfile(_lib.variant_path("oracles")):
            pytest.skip("lib/variants/oracles.so not built")
        variant = _lib.use_variant("oracles")
        variant.__enter__()
        request.addfinalizer(lambda: variant.__exit__(None, None, None))
    net = _net(n_hidden, skip_in)
    g = torch.Generator(device=DEV).manual_seed(2)
    pts = (torch.rand(n, 3, device=DEV, generator=g) * 1.2 - 0.6).contiguous()
    loss = eikonal_sq_sum(net, pts) * 0.37
    grads = torch.autograd.grad(loss, list(net.parameters()))
    net64 = copy.deepcopy(net).double()
    v = pts.double().requires_grad_(True)
    gr = torch.autograd.grad(net64(v).sum(), v, create_graph=True)[0]
    loss64 = (gr.pow(2).sum(dim=-1).sqrt() - 1).pow(2).sum() * 0.37
    ref = torch.autograd.grad(loss64, list(net64.parameters()), allow_unused=True)
    assert abs(float(loss) - float(loss64)) <= 1e-5 * abs(float(loss64))
    print("  eikonal term, relative L2 error vs float64 double backward per parameter: " + " ".join("%.1e" % _rel(a, b) for a, b in zip(grads, ref) if b is not None))
    for (name, p), a, b in zip(net.named_parameters(), grads, ref):
        if b is None:                      # the output bias does not influence grad_x f
            assert float(a.abs().max()) == 0.0, name
            continue
        assert _rel(a, b) < 1e-4, (name, _rel(a, b))


def test_eikonal_reverse_formulation_gradient_of_network_output_and_empty_input():
    """grad_x f of the reverse-over-reverse pass (its first two launches) vs autograd of the fp64 network; zero samples are legal"""
    from gshell_amd.geometry import mlp as M
    net = _net()
    g = torch.Generator(device=DEV).manual_seed(3)
    n = 333
    pts = (torch.rand(n, 3, device=DEV, generator=g) - 0.5).contiguous()
    net64 = copy.deepcopy(net).double()

    class Ctx:
        pass
    ctx = Ctx()
    with torch.no_grad():
        loss = M._EikonalRRFn.forward(ctx, pts, net, None)
    v = pts.double().requires_grad_(True)
    gr = torch.autograd.grad(net64(v).sum(), v)[0]
    assert float((ctx.grad_f.double() - gr).abs().max()) < 1e-5 * float(gr.abs().max())
    assert abs(float(loss) - float((gr.norm(dim=-1) - 1).pow(2).sum())) < 1e-5 * float(loss)
    empty = M.eikonal_sq_sum(net, pts[:0])
    assert float(empty) == 0.0
    grads = torch.autograd.grad(empty, list(net.parameters()))
    assert all(float(t.abs().max()) == 0.0 for t in grads)


def test_eikonal_gradient_of_network_output_equals_autograd(oracle_kernels):
    """The tangent rows themselves: df/dx from the forward-mode pass (<EIK> planes: an oracle kernel, lib/variants/oracles.so) vs autograd of the
    fp64 network."""
    from gshell_amd.geometry.mlp import _SavedChain
    net = _net()
    g = torch.Generator(device=DEV).manual_seed(3)
    n = 333
    pts = (torch.rand(n, 3, device=DEV, generator=g) - 0.5).contiguous()
    saved = _SavedChain(net, 2, pts, None, n)
    tiles = saved.Rpad // 64
    ov = saved.out.view(tiles, 4, 16)
    J = ov[:, 1:4, :].permute(0, 2, 1).reshape(tiles * 16, 3)[:n]
    f = ov[:, 0, :].reshape(-1)[:n]
    net64 = copy.deepcopy(net).double()
    v = pts.double().requires_grad_(True)
    y = net64(v)
    gr = torch.autograd.grad(y.sum(), v)[0]
    lin = [m for m in net64.net if isinstance(m, torch.nn.Linear)]
    assert float((f.double() + lin[-1].bias - y[:, 0]).abs().max()) < 2e-6
    assert float((J.double() - gr).abs().max()) < 1e-5 * float(gr.abs().max())


def test_row_sparse_backward_without_host_sync_equals_the_synchronising_path():
    """net._gs_rows_bound (set by GShellTetsGeometry.getMesh to 2 x crossing edges) switches the row-sparse backward to the
    device-side compaction (gs_compact_rows) with planes sized by the bound and the count kept on the device: same gradients as
    the path that waits for torch.nonzero; a violated bound is reported at the next call, not silently truncated."""
    from gshell_amd.geometry.mlp import row_sparse_backward, split_param_grads
    from gshell_amd._lib import GShellHipError
    net = _net()
    g = torch.Generator(device=DEV).manual_seed(4)
    N = 9000 + 5
    x = (torch.rand(N, 3, device=DEV, generator=g) * 1.4 - 0.7).contiguous()
    gy = torch.zeros(N, 1, device=DEV)
    idx = torch.randperm(N, device=DEV, generator=g)[:1500].sort().values
    gy[idx, 0] = torch.randn(idx.numel(), device=DEV, generator=g) * 1e-4
    gx_a, flat_a = row_sparse_backward(net, x, gy, True)
    grads_a = split_param_grads(net, flat_a)
    net._gs_rows_bound = 4000                                     # > 1500 rows, < N
    gx_b, flat_b = row_sparse_backward(net, x, gy, True)
    grads_b = split_param_grads(net, flat_b)
    assert torch.allclose(gx_a, gx_b, rtol=1e-5, atol=1e-12)
    for a, b in zip(grads_a, grads_b):
        assert _rel(b, a) < 2e-5                                  # float-atomic order of the strip sums
    net._gs_rows_bound = 1000                                     # too small: flagged ...
    row_sparse_backward(net, x, gy, False)
    torch.cuda.synchronize()
    with pytest.raises(GShellHipError):
        row_sparse_backward(net, x, gy, False)                    # ... at the next call
    del net._gs_rows_bound


def test_packed_weights_are_reused_until_a_parameter_changes():
    """pack_weights_h2 hands back the SAME image while every parameter's (storage, version) is unchanged and packs again after an in-place
    update by torch OR by HipAdam (whose kernel writes through raw pointers and bumps the version counters itself)"""
    from gshell_amd.geometry.mlp import pack_weights_h2
    from gshell_amd.optim import HipAdam
    net = _net()
    a, _, _ = pack_weights_h2(net)
    b, _, _ = pack_weights_h2(net)
    assert a is b
    st = torch.zeros(2, dtype=torch.int32, device=DEV)
    c, _, _ = pack_weights_h2(net, st)                 # a call that asks for the fp16-range check always packs
    # the words the packer writes (the section of the kernel variant that is not selected and the staging pad behind the image are never read and
    # stay whatever torch.empty handed out): pack into a zero-filled and a one-filled buffer and keep the words that agree
    z, o = torch.zeros_like(a), torch.full_like(a, -1)
    pack_weights_h2(net, out=z)
    pack_weights_h2(net, out=o)
    w = z == o
    assert 0.4 * a.numel() < int(w.sum()) < a.numel()
    assert pack_weights_h2(net)[0] is c                 # packing into a caller's buffer leaves the cache alone (it holds the last packed image: c)
    assert c is not a and torch.equal(a[w], c[w]) and torch.equal(a[w], z[w])
    with torch.no_grad():
        next(iter(net.parameters())).mul_(1.5)
    d, _, _ = pack_weights_h2(net)
    assert d is not c and not torch.equal(c[w], d[w])
    opt = HipAdam(net.parameters(), lr=1e-2)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    e, _, _ = pack_weights_h2(net)
    assert e is not d and not torch.equal(d[w], e[w])
    f, _, _ = pack_weights_h2(net)
    assert f is e
