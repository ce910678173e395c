"""Two-pass SDF forward (one-product pass over the grid + three-product refinement, csrc/mlp_h2.hip k_h1_fwd / k_h2_fwd<FIX>) and the
fp16-range guard of the h2 arithmetic."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("res,steps", [(64, 150), (256, 400), (256, 30)])
def test_two_pass_forward_equals_one_pass_where_the_reference_consumes_it(res, steps):
    """Every sign (extraction: occ = sdf > 0, ref gshell_tets.py:250) and the value at both end points of every sign-crossing edge
    (interpolation weights :277-290, sdf regulariser gshell_tets_geometry.py:33-39) equal the one-pass h2 kernel's BIT FOR BIT;
    the measured one-product error on the refined rows stays below tau / 4; the extraction of both tensors is identical.
    (256, 30): a barely fitted network -- another weight state."""
    from gshell_amd.geometry import mlp
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    from tools import two_pass
    net, verts, topo = two_pass.build(res, steps=steps)
    tau = mlp.SDF_TWO_PASS_TAU
    r = two_pass.compare(net, verts, topo, tau)
    print(r)
    assert r["sign_disagreements"] == 0 and r["occupancy_words_equal"] and r["sign_bits_match_values"]
    assert r["end_point_values_bit_identical"]
    assert not r["nonfinite"]
    assert r["max_abs_dev_one_product_on_refined_rows"] * mlp.SDF_TWO_PASS_SAFETY < tau
    assert r["crossing_edge_end_points"] <= r["refined_rows"] <= 0.25 * r["rows"]
    assert r["max_abs_diff_all_rows"] < tau          # what the off-surface values of the returned tensor are off by
    # the extraction of the two tensors: identical topology AND identical floats
    _, tets = __import__("gshell_amd.grid", fromlist=["grid"]).grid_for_res(res, device=DEV)
    msdf = (0.32 - verts[:, 1] + 0.05 * torch.sin(8.0 * verts[:, 0])).contiguous()
    ext = GShell_Tets()
    with torch.no_grad():
        y1 = mlp.fused_forward(net, verts, "h2")
        y2 = mlp.fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr(), refine_topo=topo)
        a = ext(verts, y1, msdf, tets)
        b = ext(verts, y2, msdf, tets)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) and torch.equal(a[5]["msdf"], b[5]["msdf"])
    assert a[1].shape[0] > 1000


def test_two_pass_falls_back_when_the_one_product_error_exceeds_its_budget():
    from gshell_amd import _lib
    from gshell_amd.geometry import mlp
    from tools import two_pass
    net, verts, topo = two_pass.build(64, steps=100)
    old = mlp.SDF_TWO_PASS_TAU
    try:
        mlp.SDF_TWO_PASS_TAU = 1e-5          # far below the one-product error: the a-posteriori check must object
        with pytest.raises(_lib.GShellHipError, match="one-product"):
            mlp.fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr(), refine_topo=topo)
    finally:
        mlp.SDF_TWO_PASS_TAU = old


def test_h2_range_guard_detects_overflow_and_the_exact_kernel_is_finite():
    """The reference's fp32 GEMMs (geometry/mlp.py:32-40) have no range limit; the fp16-pair arithmetic has: activations >= 65 504
    become inf in the split.  Weights scaled x 1e3: both h2 paths must say so, the exact-fp32 kernel must still be right."""
    from gshell_amd import _lib
    from gshell_amd.geometry import mlp
    from tools import two_pass
    net, verts, topo = two_pass.build(64, steps=0)
    with torch.no_grad():
        for m in net.net:
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(1e3 if m.out_features == 256 and m.in_features == 256 else 1.0)
        ref = net(verts[:4096])
        assert float(ref.abs().max()) > 1e6 and torch.isfinite(ref).all()          # finite in fp32, far beyond fp16 inside
        y32 = mlp.fused_forward(net, verts[:4096], "fp32")
        assert torch.allclose(y32, ref, rtol=1e-4, atol=1e-3 * float(ref.abs().max()))
        with pytest.raises(_lib.GShellHipError, match="fp16 range"):
            mlp.fused_forward(net, verts, "h2")
        with pytest.raises(_lib.GShellHipError, match="fp16 range"):
            mlp.fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr(), refine_topo=topo)
        # a single huge WEIGHT (clamped by the fp16 conversion of the packer) is caught even if no activation overflows
        net2, verts2, _ = two_pass.build(64, steps=0)
        lin = [m for m in net2.net if isinstance(m, torch.nn.Linear)]
        lin[2].weight[3, 5] = 7.0e4
        lin[2].weight[:, 5].mul_(0.0)
        lin[2].weight[3, 5] = 7.0e4
        with pytest.raises(_lib.GShellHipError, match="fp16 range"):
            mlp.fused_forward(net2, verts2[:256], "h2")


def test_getmesh_survives_an_fp16_overflow_by_switching_to_the_fp32_path():
    """GShellTetsGeometry.getMesh reads the status words after the extraction's own sync and re-runs: the iteration still produces
    the mesh the fp32 evaluation implies, and the event is counted (bench.py refuses to print a number when it happens)."""
    from gshell_amd import workload
    from gshell_amd.geometry import mlp
    tr = workload.build(res=16, n_samples=1, batch=1, train_res=(32, 32), fit_steps=60)
    geo = tr.geometry
    with torch.no_grad():
        out0 = geo.getMesh(tr.mat)
        lin = [m for m in geo.sdf_net.net if isinstance(m, torch.nn.Linear)]
        # an exactly cancelling pair of huge hidden units: the function is unchanged in exact arithmetic, the activations are not
        lin[1].weight[0].mul_(3e6)
        lin[1].bias[0] = 1.0
        lin[2].weight[:, 0] = 0.0
    mlp.FALLBACKS.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            out1 = geo.getMesh(tr.mat)
    assert any("overflowed" in str(x.message) for x in w)
    assert geo.sdf_net.__dict__.get("_gs_precision") == "torch" and mlp.FALLBACKS
    assert torch.isfinite(out1['sdf']).all()
    assert torch.equal(out1['imesh'].t_pos_idx, out0['imesh'].t_pos_idx)      # the dead unit does not move the surface


def test_flexicubes_getmesh_is_identical_with_the_two_pass_forward():
    """G-FlexiCubes consumes SDF signs at every grid vertex and values only at the end points of sign-changing cube edges
    (gshell_flexicubes.py:387-485), so the two-pass forward over the unique cube edges (mlp.EdgeList) must give the SAME mesh, regulariser
    and mSDF values as the one-pass h2 kernel, bit for bit, and the same sdf at every crossing-edge end point."""
    from gshell_amd import workload
    from gshell_amd.geometry import mlp
    tr = workload.build(res=48, n_samples=2, batch=1, train_res=(64, 64), fit_steps=150, geometry="flexicubes")
    geo = tr.geometry
    before = dict(mlp.FALLBACKS)
    res = {}
    for two_pass in (False, True):
        mlp.SDF_TWO_PASS = two_pass
        geo.sdf_net.__dict__.pop("_gs_two_pass_maxdev", None)
        try:
            with torch.no_grad():
                d = geo.getMesh(tr.mat)
        finally:
            mlp.SDF_TWO_PASS = True
        res[two_pass] = (d["imesh"].v_pos.clone(), d["imesh"].t_pos_idx.clone(), d["msdf"].clone(), d["sdf"].detach().clone(), float(geo.gflexi_reg_loss),
                         geo.sdf_net.__dict__.get("_gs_two_pass_maxdev"))          # the refinement's measured deviation: only a two-pass run records one
    a, b = res[False], res[True]
    assert a[5] is None and b[5] is not None and 0 < b[5] * mlp.SDF_TWO_PASS_SAFETY < mlp.SDF_TWO_PASS_TAU, "the second evaluation must have run in two passes"
    assert a[1].shape[0] > 500
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and a[4] == b[4]
    e = geo.all_edges.long()
    sa, sb = a[3].reshape(-1), b[3].reshape(-1)
    assert torch.equal(sa > 0, sb > 0)
    cross = (sa[e[:, 0]] > 0) != (sa[e[:, 1]] > 0)
    ends = torch.unique(e[cross].reshape(-1))
    assert ends.numel() > 500 and torch.equal(sa[ends], sb[ends])
    assert float((sa - sb).abs().max()) < mlp.SDF_TWO_PASS_TAU
    assert dict(mlp.FALLBACKS) == before, (before, dict(mlp.FALLBACKS))          # no evaluation of this test left the HIP kernels
