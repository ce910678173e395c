"""Mint the end-to-end chain fixtures tests/golden/chain_<name>.npz -- ON THE CPU, no GPU anywhere (TEST INFRASTRUCTURE).

    python -m oracle.make_golden_chain --fit-net            # once: tests/golden/chain_sdf_net.npz (the SDF network all chains share)
    python -m oracle.make_golden_chain config0_a flexi32    # some chains
    python -m oracle.make_golden_chain --all                # every chain of oracle/chain_recipe.CHAINS (the res-256 headline chain: tens of minutes)

One chain = one training iteration's `tick` of the reference (geometry/gshell_tets_geometry.py:257-384; FlexiCubes: gshell_flexicubes_geometry
.py:237-364) on the state oracle/chain_recipe.py builds from the chain's name, evaluated by the oracle chain

    oracle/mlp_oracle.forward  ->  oracle/mtets_oracle.extract | flexi_oracle.extract  ->  oracle/pipeline_oracle.render_mesh  ->  oracle/tick_oracle.tick

TWICE: in float32 (what the parity tests compare values with: mesh, buffers, losses) and in float64 (the ARBITER of the gradient bars:
wherever |float32 oracle - float64 oracle| of a parameter gradient exceeds the north-star 1e-4 because the function itself is not defined
better than that in float32 -- piecewise-trilinear texture slopes, cancelling float sums -- the test's bar is a fixed multiple of that MEASURED
distance, not a hand-set number).  The float64 run evaluates the SAME discrete configuration as the float32 run: SDF values, mesh vertices and the sampler's
decision inputs are pinned to the float32 run's values (straight-through), so the two runs differ by arithmetic only.

The SDF network runs over every grid vertex without a graph; its Jacobian is applied afterwards on the rows that receive gradient (end points
of sign-crossing edges) -- the chain rule split at `sdf`, exact (rows without upstream gradient contribute exactly zero), which is what lets
the res-256 chain (2.28 M rows) fit this machine.

Stored per chain: inputs that cannot be regenerated (target images, eikonal samples; float16), the float32 mesh (faces, referenced vertices,
augmented mSDF), the float32 buffers (16-bit codes over each buffer's range), both runs' losses and terms, the float64 gradient of every
parameter tensor (float32; sparse rows for the grid-sized ones; count-sketches for tensors above 4 M entries), the float32-vs-float64
distance of each, and d loss / d v_pos of the float64 run (for the flipped-sample accounting of the tests)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import chain_recipe as cr                     # noqa: E402
from oracle import mlp_oracle as mlp                      # noqa: E402
from oracle import mtets_oracle, pipeline_oracle as pl, pixel_oracle as po, raster_oracle as ro_mod, shade_oracle as so, tick_oracle   # noqa: E402

SKETCH_BUCKETS = 1 << 16
SKETCH_ABOVE = 4_000_000


# ---- the shared SDF network -----------------------------------------------------------------------------------------------------------
def fit_net(steps=300, batch=32768, seed=0):
    """Reference-shaped network (geometry/mlp.py: 39 -> 256 x 7 -> 1, skip at hidden layer 3) fitted to the benchmark's capped-cone state at
    random points of the grid's box (independent of any grid), nn.Linear's default initialisation."""
    torch.manual_seed(seed)
    dims = [(39, 256)] + [(256 + (39 if i == 3 else 0), 256) for i in range(6)] + [(256, 1)]
    lin = [torch.nn.Linear(i, o) for i, o in dims]
    state = {}
    for j, m in enumerate(lin):
        state[f"net.{2 * j}.weight"], state[f"net.{2 * j}.bias"] = m.weight, m.bias
    opt = torch.optim.Adam(list(state.values()), lr=1e-3)
    g = torch.Generator().manual_seed(seed + 1)
    t0 = time.time()
    for it in range(steps):
        x = (torch.rand(batch, 3, generator=g) - 0.5) * cr.MESH_SCALE * 1.02
        loss = (mlp.forward(state, x)[:, 0] - cr.skirt_sdf(x)).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 50 == 0 or it + 1 == steps:
            print(f"  fit step {it}: loss {float(loss):.3e} ({time.time() - t0:.0f} s)", flush=True)
    np.savez(cr.NET_FILE, **{k: v.detach().numpy() for k, v in state.items()})
    print(f"wrote {cr.NET_FILE}")


# ---- one run of the chain ---------------------------------------------------------------------------------------------------------------
def _leaf(t, dt):
    return t.detach().to(dt).clone().requires_grad_(True)


def _texture(sc, dt, leaves):
    if sc["textured"]:
        p = _leaf(sc["tex_params"], dt)
        ws = [_leaf(w, dt) for w in sc["tex_w"]]
        leaves["tex_params"], leaves["tex_w"] = p, ws
        return pl.TextureOracle((sc["aabb"][0].to(dt), sc["aabb"][1].to(dt)), cr.TEX_CFG, p, ws, sc["min_max"][0].to(dt), sc["min_max"][1].to(dt))
    v = _leaf(sc["material"], dt)
    leaves["material"] = v
    return pl.ConstantTextureOracle(v)


def extract(sc, pos, sdf, msdf, cube_w):
    if sc["kind"] == "flexicubes":
        from oracle import flexi_oracle as fo
        fv, ff, L_dev, fex = fo.extract(pos, sdf, msdf, sc["indices"], sc["res"], cube_w[:, :12], cube_w[:, 12:20], cube_w[:, 20])
        return {"verts_aug": fv, "faces_aug": ff, "msdf": fex["msdf"], "msdf_boundary": fex["msdf_boundary"], "n_verts_watertight": fex["n_verts_watertight"],
                "L_dev": L_dev}
    return mtets_oracle.extract(pos, sdf, msdf, sc["indices"], with_tangents=False)


def all_edges(sc):
    """sorted unique (min, max) grid edges (gshell_tets_geometry.py:141-156 / gshell_flexicubes_geometry.py:124-127)"""
    if "all_edges" not in sc:
        if sc["kind"] == "flexicubes":
            from oracle import flexi_oracle as fo
            e = fo.build_topology(sc["indices"], sc["N"])["edges"]
            e = torch.sort(e, dim=1).values
            sc["all_edges"] = torch.unique(e, dim=0)
        else:
            sc["all_edges"] = mtets_oracle.build_topology(sc["indices"])["edges"]
    return sc["all_edges"]


def run_chain(sc, dt, pins, target_img, sampled_pts, log=print):
    """One evaluation of the chain in dtype `dt`.  float32: fills `pins` (sdf values, mesh values, sampler decision inputs).  float64: reads them."""
    first = dt == torch.float32
    t0 = time.time()
    leaves = {}
    state = {k: _leaf(v, dt) for k, v in sc["sdf_net"].items()}
    leaves["sdf_net"] = state
    deform, msdf, light = _leaf(sc["deform"], dt), _leaf(sc["msdf"], dt), _leaf(sc["light"], dt)
    leaves.update(deform=deform, msdf=msdf, light=light)
    cube_w = None
    if sc["kind"] == "flexicubes":
        cube_w = _leaf(sc["cube_w"], dt)
        leaves["per_cube_weights"] = cube_w
    tex = _texture(sc, dt, leaves)
    v_def = sc["verts"].to(dt) + sc["max_displacement"] * deform

    # ---- A: SDF of every grid vertex, no graph
    if first:
        pins["sdf"] = mlp.forward_chunked(sc["sdf_net"], v_def.detach())
    sdf_leaf = pins["sdf"].to(dt).clone().requires_grad_(True)
    v_leaf = v_def.detach().clone().requires_grad_(True)
    log(f"    [{dt}] SDF over {sc['N']} rows: {time.time() - t0:.0f} s")

    # ---- B: extraction -> render -> tick, from (v_leaf, sdf_leaf)
    ex = extract(sc, v_leaf, sdf_leaf, msdf, cube_w)
    v, f = ex["verts_aug"], ex["faces_aug"]
    if first:
        pins.update(v=v.detach().clone(), f=f.clone(), msdf_aug=ex["msdf"].detach().clone())
    else:
        assert torch.equal(f, pins["f"]), "the float64 extraction chose a different topology"
        v = pins["v"].to(dt) + (v - v.detach())                     # values of the float32 mesh, graph of this run
        ex["msdf"] = pins["msdf_aug"].to(dt) + (ex["msdf"] - ex["msdf"].detach())
    v.retain_grad()
    msdf_aug = ex["msdf"]
    so.DECISION_PIN = {"mode": "record" if first else "replay", "calls": [] if first else list(pins["decisions"])}
    old_hit = so.ANY_HIT
    so.ANY_HIT, so.CHECKPOINT = so.any_hit_c, True          # (memory: the graph of one sample batch / one filter row alive at a time; same arithmetic)
    try:
        out = pl.render_mesh(v, f, po.auto_normals(v, f), msdf_aug, sc["mvp"].to(dt), sc["campos"].to(dt), light, sc["background"].to(dt),
                             {k: t.to(dt) for k, t in sc["noise"].items()}, tex, sc["n"], sc["seed"], sc["shadow"], sc["perms"].numpy(), bsdf="pbr",
                             denoise_sigma=sc["sigma"], resolution=(sc["H"], sc["W"]), xfm=ro_mod.xfm_points_kernel_order, covered_texture=True)
    finally:
        if first:
            pins["decisions"] = so.DECISION_PIN["calls"]
        so.ANY_HIT, so.CHECKPOINT, so.DECISION_PIN = old_hit, False, None
    log(f"    [{dt}] render: {time.time() - t0:.0f} s; V_aug {v.shape[0]} T {f.shape[0]}")
    d_o = {"buffers": out, "imesh_faces": f, "msdf": msdf_aug, "msdf_boundary": ex["msdf_boundary"], "n_verts_watertight": ex["n_verts_watertight"],
           "sdf": sdf_leaf, "sampled_pts": sampled_pts.to(dt)}

    class _Net:                                             # what tick_oracle's eikonal term calls
        def __call__(self, x):
            return mlp.forward(state, x)
    img_o, _, reg_o, terms = tick_oracle.tick(sc["flags"], sc["res"], _Net(), all_edges(sc), d_o, {"img": target_img.to(dt)}, sc["iteration"])
    if "L_dev" in ex:
        terms["L_dev"] = ex["L_dev"].mean() * 0.25           # gshell_flexicubes_geometry.py:358
        reg_o = reg_o + terms["L_dev"]
    (img_o + reg_o).backward()
    log(f"    [{dt}] tick + backward: {time.time() - t0:.0f} s")

    # ---- C: the network's Jacobian on the rows that carry gradient
    g_sdf = sdf_leaf.grad.reshape(-1)
    rows = torch.nonzero(g_sdf != 0).reshape(-1)
    g_v = v_leaf.grad.clone() if v_leaf.grad is not None else torch.zeros_like(v_leaf)
    for i in range(0, rows.numel(), 65536):
        r = rows[i:i + 65536]
        x = v_def.detach()[r].clone().requires_grad_(True)
        s = mlp.forward(state, x)
        s.backward(g_sdf[r].reshape(s.shape))
        g_v[r] += x.grad
    deform.backward(g_v * sc["max_displacement"])            # v_def = verts + max_displacement * deform
    log(f"    [{dt}] network Jacobian on {rows.numel()} rows: {time.time() - t0:.0f} s")

    grads = {f"sdf_net.{k}": p.grad for k, p in state.items()}
    grads.update(deform=deform.grad, msdf=msdf.grad, light=light.grad)
    if cube_w is not None:
        grads["per_cube_weights"] = cube_w.grad
    if sc["textured"]:
        grads["tex_params"] = leaves["tex_params"].grad
        for i, w in enumerate(leaves["tex_w"]):
            grads[f"tex_w{i}"] = w.grad
    else:
        grads["material"] = leaves["material"].grad
    for k, g in grads.items():
        assert g is not None and torch.isfinite(g).all(), k
    res = {"out": {k: t.detach() for k, t in out.items()}, "img": float(img_o), "reg": float(reg_o), "terms": {k: float(t) for k, t in terms.items()},
           "grads": {k: g.detach() for k, g in grads.items()}, "g_v_pos": v.grad.detach(), "g_sdf": g_sdf.detach(), "rows": int(rows.numel()),
           "ex": {"msdf_boundary": ex["msdf_boundary"].detach(), "n_verts_watertight": int(ex["n_verts_watertight"])}}
    return res


# ---- target + eikonal samples ---------------------------------------------------------------------------------------------------------
def make_target(sc, pins_mesh):
    """The fixed target of the iteration: the SAME mesh seen from cameras yawed by 1.5 degrees under a 1.6 x brighter probe, 2 shadow rays per
    pixel, denoised (forward only).  Silhouettes and colours differ from the iteration's own render, so every loss term has a gradient."""
    v, f, msdf_aug = pins_mesh
    n_t = 1
    g = cr._gen(13)
    B, H, W = sc["B"], sc["H"], sc["W"]
    noise = {"jitter": torch.zeros(B, H, W, 2), "texture": torch.zeros(B, H, W, 3), "tangent": torch.randn(B, H, W, 3, generator=g)}
    perms = torch.argsort(torch.rand(cr.PERM_ROWS, n_t * n_t, generator=g), dim=-1).int()
    if sc["textured"]:
        tex = pl.TextureOracle(sc["aabb"], cr.TEX_CFG, sc["tex_params"], sc["tex_w"], *sc["min_max"])
    else:
        tex = pl.ConstantTextureOracle(sc["material"])
    old_hit = so.ANY_HIT
    so.ANY_HIT = so.any_hit_c
    try:
        with torch.no_grad():
            out = pl.render_mesh(v, f, po.auto_normals(v, f), msdf_aug, sc["target_mvp"], sc["target_campos"], sc["light"] * 1.6, sc["background"], noise, tex,
                                 n_t, 7, 1.0, perms.numpy(), bsdf="pbr", denoise_sigma=1.0, resolution=(H, W), xfm=ro_mod.xfm_points_kernel_order, covered_texture=True)
    finally:
        so.ANY_HIT = old_hit
    img = torch.cat((out["shaded"][..., 0:3].clamp(0, 4), (out["shaded"][..., 3:4] > 0.5).float()), -1)
    return img.half()


def sample_points(v, f, n, seed):
    """area-weighted surface samples (what kaolin.ops.mesh.sample_points draws, gshell_tets_geometry.py:236), rounded to float16"""
    g = cr._gen(seed)
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = torch.linalg.cross(v1 - v0, v2 - v0).norm(dim=-1).double()
    cdf = torch.cumsum(area, 0)
    r = torch.rand(n, 3, generator=g, dtype=torch.float64)
    fid = torch.searchsorted(cdf, r[:, 2] * cdf[-1]).clamp(max=f.shape[0] - 1)
    u, w = r[:, 0:1].sqrt(), r[:, 1:2]
    p = (1 - u) * v0[fid].double() + u * (1 - w) * v1[fid].double() + u * w * v2[fid].double()
    return p.float().half()


# ---- encoding -------------------------------------------------------------------------------------------------------------------------
def sketch_plan(numel, seed=17):
    """count-sketch of a tensor too large to store: entry i goes to bucket h(i) with sign s(i); E |sketch(e)|^2 = |e|^2 with relative standard
    deviation sqrt(2 / buckets) = 0.55 %, so the relative L2 distance of two tensors is read off their sketches to better than 1 %"""
    g = cr._gen(seed)
    h = torch.randint(0, SKETCH_BUCKETS, (numel,), generator=g)
    s = torch.randint(0, 2, (numel,), generator=g, dtype=torch.int8) * 2 - 1
    return h, s


def sketch(t, plan):
    h, s = plan
    flat = t.detach().reshape(-1).double().cpu()
    return torch.zeros(SKETCH_BUCKETS, dtype=torch.float64).index_add_(0, h, flat * s.double())


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def mint(name, out_dir=cr.GOLDEN, log=print):
    t_start = time.time()
    sc = cr.inputs(name)
    assert sc["sdf_net"] is not None, "run --fit-net first"
    log(f"chain {name}: {sc['kind']} res {sc['res']}, N {sc['N']}, {sc['B']} x {sc['H']}^2, n {sc['n']}")
    # the mesh of the state (float32 extraction), for the target and the eikonal samples
    with torch.no_grad():
        v_def = sc["verts"] + sc["max_displacement"] * sc["deform"]
        sdf0 = mlp.forward_chunked(sc["sdf_net"], v_def)
        ex0 = extract(sc, v_def, sdf0, sc["msdf"], sc.get("cube_w"))
    v0, f0 = ex0["verts_aug"], ex0["faces_aug"]
    log(f"  mesh: V_aug {v0.shape[0]}, T {f0.shape[0]} ({time.time() - t_start:.0f} s)")
    target = make_target(sc, (v0, f0, ex0["msdf"]))
    pts = sample_points(v0, f0, 50000, 19)
    log(f"  target rendered ({time.time() - t_start:.0f} s); alpha coverage {float(target[..., 3].float().mean()):.3f}")

    pins = {}
    r32 = run_chain(sc, torch.float32, pins, target.float(), pts.float(), log)
    r64 = run_chain(sc, torch.float64, pins, target.float(), pts.float(), log)

    z = {"name": name, "checksums_keys": [], "target_img": target.numpy(), "sampled_pts": pts.numpy(), "mvp": sc["mvp"].numpy(), "campos": sc["campos"].numpy()}
    cs = cr.checksums(sc)
    z["checksums_keys"] = np.array(list(cs.keys()))
    z["checksums_vals"] = np.array(list(cs.values()), dtype=np.float64)
    # mesh of the float32 run
    v, f = pins["v"], pins["f"]
    used = torch.zeros(v.shape[0], dtype=torch.bool)
    used[f.reshape(-1)] = True
    z.update(faces=f.numpy().astype(np.int32), n_verts=v.shape[0], used_idx=torch.nonzero(used).reshape(-1).numpy().astype(np.int32),
             verts_used=v[used].numpy(), msdf_aug=pins["msdf_aug"].numpy(), n_verts_watertight=r32["ex"]["n_verts_watertight"],
             g_v_pos64=r64["g_v_pos"][used].float().numpy(), g_v_pos_rel32=rel(r32["g_v_pos"], r64["g_v_pos"]))
    if not bool((v[~used] == 0).all()):
        z["verts_unused"] = v[~used].numpy()               # FlexiCubes: boundary slots no face references keep their (extrapolated) values
    # buffers of the float32 run; the pixels where anything is drawn
    out = r32["out"]
    z["visible_triangles"] = out["visible_triangles"].numpy().astype(np.int32)
    for key, buf in out.items():
        if key == "visible_triangles":
            continue
        q, lo, hi = cr.quantise(buf)
        z[f"buf_{key}"] = q
        z[f"buf_{key}_range"] = np.array([lo, hi], dtype=np.float64)
        z[f"buf_{key}_f64_maxdev"] = float(((buf.double() - r64["out"][key]).abs().amax() / buf.abs().amax().clamp_min(1e-30)))
    # losses
    for tag, r in (("32", r32), ("64", r64)):
        z[f"img_loss{tag}"], z[f"reg_loss{tag}"] = r["img"], r["reg"]
        z[f"terms{tag}_keys"] = np.array(list(r["terms"].keys()))
        z[f"terms{tag}_vals"] = np.array(list(r["terms"].values()), dtype=np.float64)
    # gradients: float64 run (stored float32), distance of the float32 run
    names = list(r64["grads"].keys())
    z["grad_names"] = np.array(names)
    rel32 = {}
    for k in names:
        g64, g32 = r64["grads"][k], r32["grads"][k]
        rel32[k] = rel(g32, g64)
        if g64.numel() > SKETCH_ABOVE:
            plan = sketch_plan(g64.numel())
            z[f"gradsk_{k}"] = sketch(g64, plan).numpy()
            z[f"gradsk32_{k}"] = sketch(g32, plan).numpy()
            z[f"gradnorm_{k}"] = float(g64.double().norm())
        elif k in ("deform", "msdf", "per_cube_weights", "tex_params"):
            g2 = g64.reshape(g64.shape[0], -1)
            nz = torch.nonzero((g2 != 0).any(-1)).reshape(-1)
            z[f"gradrows_{k}"] = nz.numpy().astype(np.int32)
            z[f"gradvals_{k}"] = g2[nz].float().numpy()
            z[f"gradshape_{k}"] = np.array(g64.shape)
        else:
            z[f"grad_{k}"] = g64.float().numpy()
    z["grad_rel32_vals"] = np.array([rel32[k] for k in names], dtype=np.float64)
    z["rows_with_sdf_gradient"] = r64["rows"]
    nz = torch.nonzero(r64["g_sdf"] != 0).reshape(-1)                       # d loss / d sdf of the float64 run (extraction + sign regulariser), sparse
    z["g_sdf64_rows"], z["g_sdf64_vals"] = nz.numpy().astype(np.int32), r64["g_sdf"][nz].float().numpy()
    z["g_sdf_rel32"] = rel(r32["g_sdf"], r64["g_sdf"])
    # the float32 SDF values the chain was evaluated at, where a value is consumed (the rows above = end points of sign-crossing edges), and the
    # sign of every row: the tests substitute them for the product's own (2e-7 away) so that both sides extract from IDENTICAL fields
    z["sdf32_vals"] = pins["sdf"].reshape(-1)[nz].numpy()
    z["sdf_sign_bits"] = np.packbits((pins["sdf"].reshape(-1) > 0).numpy())
    z["cond_bias"] = float(r64["g_sdf"].abs().sum() / r64["g_sdf"].sum().abs().clamp_min(1e-300))     # the output bias's gradient = this signed sum
    z["mint_seconds"] = time.time() - t_start
    path = os.path.join(out_dir, f"chain_{name}.npz")
    np.savez_compressed(path, **z)
    log(f"  losses: img {r32['img']:.6f} (f64 {r64['img']:.6f}), reg {r32['reg']:.6f} (f64 {r64['reg']:.6f})")
    log("  float32 vs float64 oracle, relative L2 per gradient: " + ", ".join(f"{k} {v:.1e}" for k, v in rel32.items()))
    log(f"  d/d v_pos float32 vs float64: {z['g_v_pos_rel32']:.1e}")
    log(f"wrote {path}: {os.path.getsize(path) / 1e6:.1f} MB in {time.time() - t_start:.0f} s")
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--fit-net", action="store_true")
    ap.add_argument("--out", default=cr.GOLDEN)
    a = ap.parse_args()
    if a.fit_net:
        fit_net()
    names = list(cr.CHAINS) if a.all else a.names
    if len(names) == 1:
        mint(names[0], a.out)
    else:
        # one fresh process per chain: the res-256 chain's float64 run peaks above 40 GB and must not inherit the heap of the chains before it
        import subprocess
        for name in names:
            subprocess.run([sys.executable, "-u", "-W", "ignore", "-m", "oracle.make_golden_chain", name, "--out", a.out], check=True, cwd=ROOT)


if __name__ == "__main__":
    main()
