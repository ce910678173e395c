"""The pre-minted end-to-end chains (tests/golden/chain_*.npz, oracle/make_golden_chain.py) on the CPU: the network oracle they start from is the
reference's module, their regenerated inputs are reproducible, the fixtures are complete and re-minting one reproduces it."""
import os

import numpy as np
import pytest
import torch

from oracle import chain_recipe as cr, mlp_oracle

MINTED = [n for n in cr.CHAINS if os.path.isfile(os.path.join(cr.GOLDEN, f"chain_{n}.npz"))]


def test_network_oracle_is_the_reference_module_bit_for_bit():
    """oracle/mlp_oracle.forward == reference geometry/mlp.py + embedding.py (loaded from /root/reference when present) == the product's
    torch module on the CPU, on the fitted network every chain shares"""
    from gshell_amd.geometry.mlp import MLP
    state = cr.load_net()
    x = (torch.rand(513, 3, generator=torch.Generator().manual_seed(3)) - 0.5) * 1.4
    ours = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
    assert sorted(ours.state_dict().keys()) == sorted(state.keys())
    ours.load_state_dict(state)
    y = mlp_oracle.forward(state, x)
    assert torch.equal(y, ours(x))
    from oracle import refload
    if refload.reference_available():
        import types
        with refload.CudaToCpu():
            emb = refload.load_simple("geometry/embedding.py", "ref_embedding_chain")
            src = open(os.path.join(refload.REF_ROOT, "geometry/mlp.py")).read().replace("from .embedding import Embedding", "")
            mod = types.ModuleType("ref_mlp_chain")
            mod.Embedding = emb.Embedding
            exec(compile(src, "ref_mlp_chain", "exec"), mod.__dict__)
        ref = mod.MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
        ref.load_state_dict(state)
        assert torch.equal(y, ref(x))
    # float64 (the arbiter runs) agrees with float32 to float32 round-off
    y64 = mlp_oracle.forward({k: v.double() for k, v in state.items()}, x.double())
    assert float((y64 - y.double()).abs().max()) < 2e-6
    # the chunked no-graph evaluation is the same function
    assert float((mlp_oracle.forward_chunked(state, x, chunk=100) - y).abs().max()) < 1e-6


def test_network_oracle_softplus_and_skip_follow_the_reference():
    """known answers: Softplus(beta = 100) with torch's linear branch above 20 / beta; the encoding re-enters at hidden layer 3 only"""
    x = torch.tensor([-1.0, 0.0, 0.1, 0.19, 0.21, 3.0])
    ref = torch.where(100 * x > 20, x, torch.log1p(torch.exp(100 * x)) / 100)
    assert torch.allclose(mlp_oracle.softplus100(x), ref, atol=1e-7)
    state = cr.load_net()
    widths = [state[f"net.{2 * j}.weight"].shape[1] for j in range(8)]
    assert widths == [39, 256, 256, 256, 295, 256, 256, 256]
    e = mlp_oracle.embed(torch.tensor([[0.25, -0.5, 1.0]]))
    assert e.shape == (1, 39)
    assert torch.allclose(e[0, 3:6], torch.sin(torch.tensor([0.25, -0.5, 1.0]))) and torch.allclose(e[0, 36:39], torch.cos(32 * torch.tensor([0.25, -0.5, 1.0])))


@pytest.mark.parametrize("name", MINTED)
def test_fixture_is_complete_and_its_inputs_regenerate(name):
    z = np.load(os.path.join(cr.GOLDEN, f"chain_{name}.npz"))
    c = cr.CHAINS[name]
    if c["res"] <= 128:                                         # the res-256 grid (13.4 M tets) is rebuilt by the GPU test only
        sc = cr.inputs(name)
        cs = cr.checksums(sc)
        for k, v in zip(z["checksums_keys"], z["checksums_vals"]):
            assert cs[str(k)] == float(v), (name, str(k))
    B, H = c["B"], c["frame"]
    assert z["target_img"].shape == (B, H, H, 4) and z["target_img"].dtype == np.float16
    assert z["sampled_pts"].shape == (50000, 3)
    assert z["faces"].min() >= 0 and z["faces"].max() < int(z["n_verts"])
    assert np.array_equal(np.unique(z["faces"]), z["used_idx"])
    names = [str(s) for s in z["grad_names"]]
    assert len(names) == len(z["grad_rel32_vals"]) and np.isfinite(z["grad_rel32_vals"]).all()
    assert {"deform", "msdf", "light"} <= set(names) and sum(n.startswith("sdf_net.") for n in names) == 16
    assert ("per_cube_weights" in names) == (c["kind"] == "flexicubes")
    assert ("tex_params" in names) == c["textured"] and ("material" in names) == (not c["textured"])
    for n in names:
        assert any(f"{p}_{n}" in z.files for p in ("grad", "gradrows", "gradsk")), n
    for key in ("shaded", "z_grad", "normal", "geometric_normal", "kd", "ks", "kd_grad", "ks_grad", "normal_grad", "diffuse_light", "specular_light", "msdf_image"):
        q, (lo, hi) = z[f"buf_{key}"], z[f"buf_{key}_range"]
        assert q.dtype == np.uint16 and q.shape[:3] == (B, H, H)
        assert cr.quant_half_step(lo, hi) <= 1.53e-5
    # the two runs of the oracle agree on the values (the float64 run is pinned to the float32 run's discrete configuration)
    assert abs(float(z["img_loss32"]) - float(z["img_loss64"])) <= 2e-6 * abs(float(z["img_loss64"]))
    assert abs(float(z["reg_loss32"]) - float(z["reg_loss64"])) <= 2e-6 * abs(float(z["reg_loss64"]))
    if not c["textured"] or c.get("tex_levels", 16) <= 6:
        # without the fine hash-grid levels float32 defines every gradient to a few 1e-4 (the output bias: one signed sum over all rows)
        worst = max(v for n, v in zip(names, z["grad_rel32_vals"]) if not n.endswith("14.bias"))
        assert worst < 1e-3, (name, worst)


def test_quantised_buffers_round_trip():
    g = torch.Generator().manual_seed(1)
    for buf in (torch.rand(1, 8, 8, 4, generator=g) * 3.0, torch.randn(1, 8, 8, 3, generator=g) * 10.0, torch.zeros(1, 4, 4, 2)):
        q, lo, hi = cr.quantise(buf)
        back = cr.dequantise(q, lo, hi)
        scale = float(buf.abs().max()) or 1.0
        assert float((back - buf).abs().max()) / scale <= cr.quant_half_step(lo, hi) + 1e-7


def test_count_sketch_estimates_a_relative_distance():
    from oracle import make_golden_chain as mg
    g = torch.Generator().manual_seed(2)
    a = torch.randn(300_000, generator=g)
    b = a + 1e-3 * torch.randn(300_000, generator=g)
    plan = mg.sketch_plan(a.numel())
    sa, sb = mg.sketch(a, plan), mg.sketch(b, plan)
    est = float((sa - sb).norm() / sa.norm())
    true = float((a - b).norm() / a.norm())
    assert abs(est - true) <= 0.03 * true


@pytest.mark.skipif("flexi32" not in MINTED, reason="fixture not minted")
def test_reminting_a_chain_reproduces_the_committed_fixture(tmp_path):
    """the committed fixture IS what the committed script writes (G-FlexiCubes res 32: ~15 s of CPU)"""
    from oracle import make_golden_chain as mg
    path = mg.mint("flexi32", str(tmp_path), log=lambda *a: None)
    new, old = np.load(path), np.load(os.path.join(cr.GOLDEN, "chain_flexi32.npz"))
    assert np.array_equal(new["faces"], old["faces"]) and np.array_equal(new["target_img"], old["target_img"])
    assert float(np.abs(new["verts_used"] - old["verts_used"]).max()) <= 1e-6
    assert abs(float(new["img_loss32"]) - float(old["img_loss32"])) <= 1e-6 and abs(float(new["reg_loss32"]) - float(old["reg_loss32"])) <= 1e-6
    for n in ("deform", "msdf"):
        assert np.array_equal(new[f"gradrows_{n}"], old[f"gradrows_{n}"])
    assert float(np.abs(new["grad_light"] - old["grad_light"]).max()) <= 1e-5 * float(np.abs(old["grad_light"]).max())
    assert int(np.abs(new["buf_normal"].astype(np.int64) - old["buf_normal"].astype(np.int64)).max()) <= 1
