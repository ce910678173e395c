"""The reverse-over-reverse formulation of the eikonal term that csrc/mlp_h2.hip (MODE_RR) implements, restated pass by pass in float64 torch and
checked against what the reference does -- autograd.grad(sdf.sum(), x, create_graph=True) followed by a second backward
(geometry/gshell_tets_geometry.py:302-324) -- on the real module (geometry/mlp.py, pinned bitwise to the reference's in test_mlp_cpu.py).
Every quantity below names the plane or kernel that holds it on the device; tests/test_mlp_grad_gpu.py checks the kernels themselves."""
import torch

from gshell_amd.geometry.mlp import MLP


def _encode(x, nf):
    cols = [x]
    for k in range(nf):
        cols += [torch.sin(2.0 ** k * x), torch.cos(2.0 ** k * x)]
    return torch.cat(cols, -1)


def _jt(x, eb, nf):
    """J_enc^T eb: adjoint of the encoding (end of k_h2_bwd<ROWS>)"""
    out = eb[:, 0:3].clone()
    for k in range(nf):
        fr = 2.0 ** k
        out += fr * (torch.cos(fr * x) * eb[:, 3 + 6 * k:6 + 6 * k] - torch.sin(fr * x) * eb[:, 6 + 6 * k:9 + 6 * k])
    return out


def _j(x, gb, nf):
    """J_enc gb: tangent of the encoding (input stage of k_h2_fwd<RR>, from the saved sin / cos)"""
    cols = [gb]
    for k in range(nf):
        fr = 2.0 ** k
        cols += [fr * torch.cos(fr * x) * gb, -fr * torch.sin(fr * x) * gb]
    return torch.cat(cols, -1)


def _case(n_hidden, skip_in, seed):
    torch.manual_seed(seed)
    net = MLP(n_freq=6, d_hidden=256, n_hidden=n_hidden, skip_in=list(skip_in)).double()
    n, nf, beta = 37, 6, 100.0
    x = torch.rand(n, 3, dtype=torch.float64) - 0.5
    g_up = torch.tensor(0.37, dtype=torch.float64)
    # ---- the reference's formulation
    v = x.clone().requires_grad_(True)
    g = torch.autograd.grad(net(v).sum(), v, create_graph=True)[0]
    loss = ((g.pow(2).sum(-1).sqrt() - 1) ** 2).sum()
    ref = torch.autograd.grad(loss * g_up, list(net.parameters()), allow_unused=True)
    ref = [torch.zeros_like(p) if r is None else r for p, r in zip(net.parameters(), ref)]
    # ---- reverse over reverse, pass by pass
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    W, b = [m.weight.detach() for m in lin], [m.bias.detach() for m in lin]
    nl = len(lin) - 1
    skip = next((i for i, m in enumerate(lin[:-1]) if m.in_features == 256 + 39), -1)
    e = _encode(x, nf)
    a, inp, h = [], [], e                                    # pass 1: k_h2_fwd<ROWS> -- a_l planes, e
    for l in range(nl):
        hin = torch.cat([h, e], -1) if l == skip else h
        inp.append(hin)
        h = torch.nn.functional.softplus(hin @ W[l].t() + b[l], beta=beta)
        a.append(h)
    slope = [1 - torch.exp(-beta * al) for al in a]          # sigma'(z) from the saved value (slope_from_value)
    G = W[nl].expand(n, -1).clone()                          # pass 2: k_h2_bwd<ROWS>, g_out = 1 -- delta_l planes, grad f
    delta, ebar = [None] * nl, torch.zeros(n, 39, dtype=torch.float64)
    for l in range(nl - 1, -1, -1):
        delta[l] = G * slope[l]
        back = delta[l] @ W[l]
        if l == 0:
            ebar += back
        elif l == skip:
            G, ebar = back[:, :256], ebar + back[:, 256:]
        else:
            G = back
    gx = _jt(x, ebar, nf)
    assert torch.allclose(gx, g.detach(), rtol=1e-7, atol=1e-10)          # (sigma' from the saved value: 1 - exp(-beta a) against the logistic)
    nrm = gx.norm(dim=-1, keepdim=True)
    assert abs(float(((nrm - 1) ** 2).sum()) - float(loss)) < 1e-7 * float(loss)
    gbar = 2 * (nrm - 1) / nrm * gx * g_up                   # k_rr_loss (x the upstream scalar)
    et = _j(x, gbar, nf)                                     # pass 3: k_h2_fwd<RR> -- u_l planes, source S_l
    u, S, uin, h = [None] * nl, [None] * nl, [None] * nl, et
    for l in range(nl):
        hin = torch.cat([h, et], -1) if l == skip else h
        uin[l] = hin
        dt = hin @ W[l].t()
        u[l] = slope[l] * dt
        S[l] = beta * (1 - slope[l]) * delta[l] * dt
        h = u[l]
    zbar, G = [None] * nl, torch.zeros(n, 256, dtype=torch.float64)      # pass 4: k_h2_bwd<RR> -- zbar_l over S_l
    for l in range(nl - 1, -1, -1):
        zbar[l] = G * slope[l] + S[l]
        back = zbar[l] @ W[l]
        G = back[:, :256] if l == skip else back
    mine = []                                                # pass 5: ONE weight-gradient launch over [zbar ; delta] x [in ; uin]
    for l in range(nl):
        mine += [zbar[l].t() @ inp[l] + delta[l].t() @ uin[l], zbar[l].sum(0)]
    mine += [u[nl - 1].sum(0, keepdim=True), torch.zeros(1, dtype=torch.float64)]
    for (name, _), r, m in zip(net.named_parameters(), ref, mine):
        assert float((r - m).abs().max()) <= 1e-7 * max(float(r.abs().max()), 1e-30), name


def test_reverse_over_reverse_equals_double_backward_with_skip_layer():
    _case(6, (3,), 0)


def test_reverse_over_reverse_equals_double_backward_without_skip_layer():
    _case(2, (), 1)
