"""gshell_amd.optim.HipAdam (one HIP launch per step, csrc/adam.hip) against torch.optim.Adam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(DEV)) for s in shapes]


SHAPES = [(3,), (4096,), (4097, 3), (257, 256), (1,), (100003,), (64, 32), (8191,)]


def _run(opt_cls, steps, contract=None, **kw):
    from gshell_amd import optim
    ps = _params(1, SHAPES)
    groups = [dict(params=ps[:3], lr=3e-3), dict(params=ps[3:], lr=1e-3)]
    if contract is not None:
        optim.CONTRACT = contract
    opt = opt_cls(groups, eps=1e-8, **kw)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: 10 ** (-it * 0.002))
    g = torch.Generator().manual_seed(2)
    for it in range(steps):
        opt.zero_grad()
        for k, p in enumerate(ps):
            if it == 3 and k == 4:
                continue                                  # a parameter without gradient in one step keeps its own step count
            scale = 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))
            p.grad = (torch.randn(p.shape, generator=g) * scale).to(DEV)
        opt.step()
        sched.step()
    return ps, opt


def test_hip_adam_is_bit_identical_to_torch_fused_adam():
    from gshell_amd import optim
    default = optim.CONTRACT
    ref, ref_opt = _run(torch.optim.Adam, 12, fused=True)
    mism = {}
    try:
        for c in (0, 1, 2):
            out, _ = _run(optim.HipAdam, 12, contract=c)
            mism[c] = sum(int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(out, ref))
    finally:
        optim.CONTRACT = default
    print("  parameters differing from torch's fused Adam after 12 steps, by rounding variant:", mism)
    assert mism[default] == 0, mism


def test_hip_adam_matches_the_reference_default_adam_and_shares_its_state_dict():
    from gshell_amd import optim
    ref, ref_opt = _run(torch.optim.Adam, 12, foreach=True)            # what the reference's torch.optim.Adam(...) runs (fp32 foreach passes)
    out, opt = _run(optim.HipAdam, 12)
    for a, b in zip(out, ref):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * max(1.0, float(b.detach().abs().max()))
    sd, sd_ref = opt.state_dict(), ref_opt.state_dict()
    assert sd["param_groups"][0]["lr"] == sd_ref["param_groups"][0]["lr"]
    for k in sd_ref["state"]:
        assert set(sd["state"][k]) == set(sd_ref["state"][k]) == {"step", "exp_avg", "exp_avg_sq"}
        assert float(sd["state"][k]["step"]) == float(sd_ref["state"][k]["step"])
        assert torch.allclose(sd["state"][k]["exp_avg_sq"], sd_ref["state"][k]["exp_avg_sq"].to(DEV), rtol=1e-5, atol=0)
    # a checkpoint written by torch's optimiser resumes in HipAdam (and continues bit-identically to torch's fused one)
    ps_a, ps_b = _params(1, SHAPES), _params(1, SHAPES)
    fused_ref, fused_opt = _run(torch.optim.Adam, 5, fused=True)
    for dst in (ps_a, ps_b):
        for p, q in zip(dst, fused_ref):
            p.data.copy_(q.data)
    oa = torch.optim.Adam([dict(params=ps_a[:3], lr=3e-3), dict(params=ps_a[3:], lr=1e-3)], eps=1e-8, fused=True)
    ob = optim.HipAdam([dict(params=ps_b[:3], lr=3e-3), dict(params=ps_b[3:], lr=1e-3)], eps=1e-8)
    import copy
    oa.load_state_dict(copy.deepcopy(fused_opt.state_dict()))          # (load_state_dict keeps the tensors it is given: one copy each)
    ob.load_state_dict(copy.deepcopy(fused_opt.state_dict()))
    g = torch.Generator().manual_seed(9)
    for _ in range(3):
        for p, q in zip(ps_a, ps_b):
            p.grad = torch.randn(p.shape, generator=g).to(DEV) * 1e-3
            q.grad = p.grad.clone()
        oa.step()
        ob.step()
    for p, q in zip(ps_a, ps_b):
        assert torch.equal(p, q)


def test_hip_adam_rejects_what_it_does_not_implement():
    from gshell_amd import optim
    p = _params(3, [(5,)])
    with pytest.raises(NotImplementedError):
        optim.HipAdam(p, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        optim.HipAdam(p, amsgrad=True)
