"""Hash-grid encoding: HIP path vs the CPU oracle (fwd, d/dparams, d/dx, masked rows)."""
import numpy as np
import pytest
import torch

from oracle import hashgrid_oracle as ho

pytestmark = pytest.mark.gpu
DEV = "cuda"
PLS = float(np.exp(np.log(4096 / 16) / 15))


@pytest.mark.parametrize("cfg", [(16, 2, 19, 16, PLS), (4, 2, 10, 4, 1.7), (3, 4, 12, 8, 2.0)])
def test_hashgrid_matches_oracle(cfg):
    from gshell_amd.render.mlptexture import HashGridEncoding, _HashGridFn
    gen = torch.Generator().manual_seed(0)
    N = 3000
    x = torch.rand(N, 3, generator=gen)
    x[:7] = torch.tensor([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0, 1, 1], [0.5, 0.5, 0.5], [1, 1, 0], [0.25, 1, 0]])
    _, total = ho.level_meta(*cfg)
    params = (torch.rand(total, generator=gen) - 0.5)
    w = torch.randn(N, cfg[0] * cfg[1], generator=gen)
    x_ref, p_ref = x.clone().requires_grad_(True), params.clone().requires_grad_(True)
    out_ref = ho.encode(x_ref, p_ref, *cfg)
    (out_ref * w).sum().backward()
    enc = HashGridEncoding(3, {"otype": "HashGrid", "n_levels": cfg[0], "n_features_per_level": cfg[1], "log2_hashmap_size": cfg[2],
                               "base_resolution": cfg[3], "per_level_scale": cfg[4]})
    assert enc.params.numel() == total and enc.n_output_dims == cfg[0] * cfg[1]
    assert float(enc.params.abs().max()) <= 1e-4
    xd, pd = x.to(DEV).requires_grad_(True), params.to(DEV).requires_grad_(True)
    out = _HashGridFn.apply(xd, pd, None, enc.cfg)
    (out * w.to(DEV)).sum().backward()
    # a 1-ulp difference in the level scale can move a point across a cell boundary at the finest levels: compare robustly
    ok = ((out.cpu() - out_ref.detach()).abs() <= 1e-4 * out_ref.detach().abs() + 1e-5)
    assert ok.float().mean() > 0.999
    print(f"  hash grid: out within 1e-4: {float(ok.float().mean()):.5f}; param-grad max err / max {float((pd.grad.cpu() - p_ref.grad).abs().max() / p_ref.grad.abs().max()):.2e}; "
          f"param-grad rel L2 {float((pd.grad.cpu() - p_ref.grad).norm() / p_ref.grad.norm()):.2e}; x-grad rel L2 {float((xd.grad.cpu() - x_ref.grad).norm() / x_ref.grad.norm()):.2e}")
    # measured on MI355X (r02): max error / max 1e-7 .. 5e-7, relative L2 <= 2.4e-7 (float-atomic order)
    assert (pd.grad.cpu() - p_ref.grad).abs().max() <= 1e-5 * p_ref.grad.abs().max()
    okx = ((xd.grad.cpu() - x_ref.grad).abs() <= 1e-4 * x_ref.grad.abs() + 1e-5 * x_ref.grad.abs().max())
    assert okx.float().mean() > 0.999
    assert float((xd.grad.cpu() - x_ref.grad).norm() / x_ref.grad.norm()) < 1e-5
    # masked rows produce zeros and no gradient
    mask = (torch.arange(N) % 3 != 0).float().to(DEV)
    xm, pm = x.to(DEV).requires_grad_(True), params.to(DEV).requires_grad_(True)
    om = _HashGridFn.apply(xm, pm, mask, enc.cfg)
    assert (om[mask == 0] == 0).all() and torch.equal(om[mask > 0], out.detach()[mask > 0])
    om.sum().backward()
    assert (xm.grad[mask == 0] == 0).all()


def test_mlptexture_sample_shapes_and_grads():
    from gshell_amd.render.mlptexture import MLPTexture3D
    aabb = (torch.tensor([-1.0, -1, -1], device=DEV), torch.tensor([1.0, 1, 1], device=DEV))
    mn = torch.tensor([0, 0, 0, 0, 0.08, 0], dtype=torch.float32, device=DEV)
    mx = torch.tensor([1, 1, 1, 0.3, 1, 1], dtype=torch.float32, device=DEV)
    tex = MLPTexture3D(aabb, channels=6, min_max=[mn, mx])
    pos = (torch.rand(2, 8, 9, 3, device=DEV) * 2 - 1).requires_grad_(True)
    out = tex.sample(pos)
    assert out.shape == (2, 8, 9, 6)
    assert (out >= mn - 1e-6).all() and (out <= mx + 1e-6).all()
    out.sum().backward()
    assert tex.encoder.params.grad is not None and torch.isfinite(tex.encoder.params.grad).all()
    assert pos.grad is not None and torch.isfinite(pos.grad).all()


@pytest.mark.parametrize("N,C,masked", [(1000, 6, True), (64 * 37 + 5, 3, False), (300000, 6, True), (17, 8, True)])
def test_texture_mlp_fused_matches_torch(N, C, masked):
    """gs_texmlp_fwd/bwd vs the same network in plain fp32 torch ops (the reference's _MLP + sigmoid mapping,
    render/mlptexture.py:18-44, :87-99).  Tolerance: 1e-5 forward, 1e-4 relative on gradients (fp32, different sum order)."""
    from gshell_amd.render.mlptexture import _TexMlpFn
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = (torch.randn(N, 32, device="cuda", generator=g) * 0.5).requires_grad_(True)
    w1 = (torch.randn(32, 32, device="cuda", generator=g) * 0.3).requires_grad_(True)
    w2 = (torch.randn(32, 32, device="cuda", generator=g) * 0.3).requires_grad_(True)
    w3 = (torch.randn(C, 32, device="cuda", generator=g) * 0.3).requires_grad_(True)
    lo = torch.rand(C, device="cuda", generator=g) * 0.2
    hi = lo + 0.5 + torch.rand(C, device="cuda", generator=g)
    mask = (torch.rand(N, device="cuda", generator=g) > 0.4).float() if masked else None
    if masked and N > 10000:
        mask[5000:9000] = 0         # whole waves of background rows
    go = torch.randn(N, C, device="cuda", generator=g)
    out = _TexMlpFn.apply(x, mask, w1, w2, w3, lo, hi)
    (out * go).sum().backward()
    got = [out.detach(), x.grad.clone(), w1.grad.clone(), w2.grad.clone(), w3.grad.clone()]
    for t in (x, w1, w2, w3):
        t.grad = None
    xm = x if mask is None else x * mask[:, None]          # masked rows behave like all-zero feature rows
    ref = torch.sigmoid(torch.relu(torch.relu(xm @ w1.t()) @ w2.t()) @ w3.t()) * (hi - lo) + lo
    (ref * go).sum().backward()
    want = [ref.detach(), x.grad, w1.grad, w2.grad, w3.grad]
    assert torch.allclose(got[0], want[0], rtol=1e-5, atol=1e-5)
    for a, b, name in zip(got[1:], want[1:], ("g_x", "g_w1", "g_w2", "g_w3")):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 1e-4 * scale, name


def _field_case(shape, smooth, seed):
    g = torch.Generator().manual_seed(seed)
    if smooth:      # neighbouring pixels = neighbouring surface points (what a g-buffer looks like): exercises the LDS combining
        B, H, W = shape
        v, u = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
        z = torch.sqrt(torch.clamp(1.3 - u * u - v * v, min=0.0)) - 0.2
        pos = torch.stack((u, v, z), -1)[None].repeat(B, 1, 1, 1) * 1.05 + 0.002 * torch.randn(B, H, W, 3, generator=g)   # a rim outside the AABB
        mask = ((u * u + v * v) < 1.1).float()[None, :, :, None].repeat(B, 1, 1, 1)
    else:
        pos = torch.rand(*shape, 3, generator=g) * 2.4 - 1.2
        mask = (torch.rand(*shape, 1, generator=g) > 0.3).float()
    return pos, mask


@pytest.mark.parametrize("shape,smooth", [((2, 32, 48), True), ((3, 64, 64), True), ((1, 16, 16), False), ((2999,), False), ((2, 17, 24), True)])
def test_texture_field_fused_path_matches_separate_operators(shape, smooth):
    """MLPTexture3D.sample / sample_many through gs_hashgrid_encode_* + gs_texmlp_*_level_major (AABB normalisation, clamp and
    the two gradient hooks folded in, level-major features, LDS-combined table atomics over 16x16 tiles) against the
    operator-by-operator path (gs_hashgrid_* + gs_texmlp_* + ATen glue), which is the one pinned to the oracle above.
    Forward: same arithmetic -> 1e-6.  Gradients: float-atomic order differs -> 1e-5 of the maximum."""
    from gshell_amd.render.mlptexture import MLPTexture3D
    aabb = (torch.tensor([-1.0, -0.9, -0.8], device=DEV), torch.tensor([1.0, 1.1, 0.9], device=DEV))
    mn = torch.tensor([0, 0, 0, 0, 0.08, 0], dtype=torch.float32, device=DEV)
    mx = torch.tensor([1, 1, 1, 0.3, 1, 1], dtype=torch.float32, device=DEV)
    tex = MLPTexture3D(aabb, channels=6, min_max=[mn, mx])
    with torch.no_grad():
        tex.encoder.params.mul_(3000.0)      # U(-0.3, 0.3): features large enough to matter
    pos, mask = _field_case(shape, smooth, seed=len(shape) + shape[0])
    if not smooth:
        mask = mask.reshape(*shape, 1)
    noise = 0.01 * torch.randn(pos.shape, generator=torch.Generator().manual_seed(5))
    go = torch.randn(2, *shape, 6, generator=torch.Generator().manual_seed(6)).to(DEV) * mask.to(DEV)
    res = {}
    for fused in (False, True):
        tex.fused_field = fused
        for p in tex.parameters():
            p.grad = None
        p_d = pos.to(DEV).requires_grad_(True)
        if fused:
            a, b = tex.sample_many([p_d + noise.to(DEV), p_d], mask.to(DEV))
        else:
            a, b = tex.sample(p_d + noise.to(DEV), mask=mask.to(DEV)), tex.sample(p_d, mask=mask.to(DEV))
        ((a * go[0]).sum() + (b * go[1]).sum()).backward()
        res[fused] = [a.detach(), b.detach(), p_d.grad.clone(), tex.encoder.params.grad.clone()] + [p.grad.clone() for p in tex.net.parameters()]
    m = mask.to(DEV) > 0
    for j in (0, 1):
        assert torch.allclose(res[True][j] * m, res[False][j] * m, rtol=1e-6, atol=1e-6)
    names = ["g_pos", "g_table", "g_w1", "g_w2", "g_w3"]
    for a, b, name in zip(res[True][2:], res[False][2:], names):
        scale = float(b.abs().max())
        assert scale > 0, name
        err = float((a - b).abs().max()) / scale
        print(f"  {name}: max err / max = {err:.2e}")
        assert err <= 1e-5, (name, err)
    # rows outside the mask receive no position gradient
    assert (res[True][2] * (~m)).abs().max() == 0
    # single-set entry point
    tex.fused_field = True
    one = tex.sample(pos.to(DEV), mask=mask.to(DEV))
    assert torch.equal(one * m, res[True][1] * m)


@pytest.mark.parametrize("capacity", [3, 64, 100000])
def test_binned_table_gradient_equals_the_atomic_path_for_any_capacity(capacity):
    """gs_hashgrid_encode_bwd_binned (hashed levels' table gradient through per-bin record arrays summed in LDS) against
    gs_hashgrid_encode_bwd (atomics): same sums in another order -> 1e-5 of the maximum; the position gradient is computed by the
    same code -> bit equal.  capacity 3 / 64: almost every reservation overflows and spills to the atomic path; the reducer must
    leave the counters at zero (they are not cleared between calls), so the call is repeated on the same scratch."""
    from gshell_amd import _lib
    from gshell_amd._lib import c_float, c_int, c_int64, check, ptr, stream
    L = _lib.lib()
    cfg = (16, 2, 19, 16, float(np.exp(np.log(4096 / 16) / 15)))
    n_par = int(L.gs_hashgrid_num_params(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4])))
    nb = int(L.gs_hashgrid_bin_count(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4])))
    assert nb == 11 * 128            # levels 5..15 of the reference's texture configuration are hashed, 2^19 entries = 128 bins each
    pos, mask = _field_case((2, 64, 64), True, seed=9)
    pos, mask = pos.reshape(-1, 3).to(DEV).contiguous(), mask.reshape(-1).to(DEV).contiguous()
    N = pos.shape[0]
    g = torch.Generator().manual_seed(10)
    params = ((torch.rand(n_par, generator=g) - 0.5) * 0.6).to(DEV)
    g_feat = torch.randn(16, N, 2, generator=g).to(DEV)
    aabb = torch.tensor([[-1.0, -0.9, -0.8], [1.0, 1.1, 0.9]], device=DEV)
    head = (c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]), ptr(pos), ptr(aabb), ptr(mask), c_int64(N), ptr(params), ptr(g_feat))
    tail = (c_float(1.0), c_float(128.0), c_int64(64), c_int64(64))
    gp_a, gx_a = torch.zeros(n_par, device=DEV), torch.empty_like(pos)
    check(L.gs_hashgrid_encode_bwd(*head, ptr(gp_a), ptr(gx_a), *tail, stream()), "gs_hashgrid_encode_bwd")
    count = torch.zeros(nb + 1, dtype=torch.int32, device=DEV)          # + the spill counter
    rec = torch.empty(nb * capacity * 3, dtype=torch.int32, device=DEV)
    scale = float(gp_a.abs().max())
    assert scale > 0
    for rep in range(2):
        gp_b, gx_b = torch.zeros(n_par, device=DEV), torch.empty_like(pos)
        check(L.gs_hashgrid_encode_bwd_binned(*head, ptr(gp_b), ptr(gx_b), *tail, ptr(count), _lib.c_void_p(count.data_ptr() + 4 * nb), ptr(rec), c_int64(capacity), stream()), "binned")
        assert int(count[:nb].abs().max()) == 0, "the reducer must leave the bin counters at zero"
        spilled = int(count[nb])                    # the word after the counters: records that took the atomic path, summed over the calls
        assert (spilled > 0) == (capacity < 1000) and (rep == 0 or spilled % 2 == 0), (capacity, spilled)
        assert torch.equal(gx_a, gx_b)
        err = float((gp_a - gp_b).abs().max()) / scale
        assert err <= 1e-5, (rep, err)
        assert torch.equal(gp_a != 0, gp_b != 0), "the two paths must touch the same table entries"
