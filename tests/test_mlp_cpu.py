"""SDF network: structure / state_dict names of the reference (geometry/mlp.py:7-40, embedding.py) and exactness of the
row-sparse backward (CPU, torch)."""
import os

import pytest
import torch

from gshell_amd.geometry.mlp import MLP, forward_row_sparse_backward


def _ref_mlp():
    from oracle import refload
    if not refload.reference_available():
        pytest.skip("reference tree not present")
    import sys
    import types
    with refload.CudaToCpu():
        emb = refload.load_simple("geometry/embedding.py", "ref_embedding")
        pkg = types.ModuleType("refgeo")
        pkg.embedding = emb
        src = open(os.path.join(refload.REF_ROOT, "geometry/mlp.py")).read().replace("from .embedding import Embedding", "")
        mod = types.ModuleType("ref_mlp")
        mod.Embedding = emb.Embedding
        exec(compile(src, "ref_mlp", "exec"), mod.__dict__)
    return mod.MLP


def test_matches_reference_module_bitwise():
    RefMLP = _ref_mlp()
    torch.manual_seed(0)
    ref = RefMLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
    ours = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
    assert list(ref.state_dict().keys()) == list(ours.state_dict().keys())
    ours.load_state_dict(ref.state_dict())
    x = torch.rand(257, 3) * 1.4 - 0.7
    assert torch.equal(ours(x), ref(x))
    assert sum(p.numel() for p in ours.parameters()) == 415233


def test_row_sparse_backward_is_exact():
    torch.manual_seed(1)
    net = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=64)
    x = (torch.rand(500, 3) - 0.5).requires_grad_(True)
    g = torch.zeros(500, 1)
    g[torch.randperm(500)[:60]] = torch.randn(60, 1)
    y_ref = net(x)
    ref = torch.autograd.grad(y_ref, [x] + list(net.parameters()), g)
    x2 = x.detach().clone().requires_grad_(True)
    y = forward_row_sparse_backward(net, x2)
    assert torch.equal(y, y_ref.detach())
    got = torch.autograd.grad(y, [x2] + list(net.parameters()), g)
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    # all-zero upstream gradient
    z = torch.autograd.grad(forward_row_sparse_backward(net, x2), [x2] + list(net.parameters()), torch.zeros(500, 1))
    assert all(float(t.abs().max()) == 0.0 for t in z)


def test_split_k_linear_matches_plain_linear_through_double_backward():
    """The split-K weight-gradient formulation (geometry/mlp.py:_SplitKLinearFn) must give the gradients of the plain
    nn.Linear stack for the eikonal-style loss, which differentiates THROUGH the input gradient (double backward)."""
    import torch
    import gshell_amd.geometry.mlp as M
    torch.manual_seed(0)
    net = M.MLP(n_freq=6, d_hidden=32, n_hidden=3, skip_in=[2]).double()
    x = torch.randn(53, 3, dtype=torch.double, requires_grad=True)      # 53 rows: not a multiple of the 16 slabs

    def grads(linear):
        saved, M._linear = M._linear, linear
        try:
            y = net(x)
            g = torch.autograd.grad(y.sum(), x, create_graph=True)[0]
            loss = ((g.norm(dim=-1) - 1) ** 2).mean() + y.pow(2).mean()
            return torch.autograd.grad(loss, list(net.parameters()) + [x])
        finally:
            M._linear = saved
    a = grads(lambda m, h: M._SplitKLinearFn.apply(h, m.weight, m.bias))
    b = grads(lambda m, h: m(h))
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-12, atol=1e-14)


def test_parameter_gate_token_is_not_reused_across_grad_modes():
    """mlp.param_gate: the grid pass leaves its token on the network for the eikonal pass of the same iteration; a token created under
    no_grad (validation render) must not be handed to a pass that needs parameter gradients."""
    from gshell_amd.geometry import mlp as M
    torch.manual_seed(0)
    net = M.MLP(n_freq=2, d_hidden=16, n_hidden=2, skip_in=[])
    with torch.no_grad():
        g0 = M.param_gate(net)
    assert not g0.requires_grad and "_gs_gate" not in net.__dict__
    g1 = M.param_gate(net)
    assert g1.requires_grad and net.__dict__["_gs_gate"] is g1
    assert M.param_gate(net, reuse=True) is g1 and "_gs_gate" not in net.__dict__
    g2 = M.param_gate(net, reuse=True)                     # nothing stashed: a fresh token
    assert g2 is not g1 and g2.requires_grad
    flat = torch.arange(float(g2.numel()))
    g2.backward(flat)
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.equal(got, flat)                          # the gate hands every parameter its slice of the flat gradient
    assert all(torch.equal(a, b) for a, b in zip(M.split_param_grads(net, flat), [p.grad for p in net.parameters()]))
