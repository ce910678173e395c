"""One whole training iteration's `tick` END TO END against PRE-MINTED oracle chains (tests/golden/chain_<name>.npz, written on the CPU by
oracle/make_golden_chain.py -- no GPU was involved in making them):

    SDF network over the grid -> G-MarchingTets / G-FlexiCubes -> normals -> BVH -> rasterise / interpolate -> (hash-grid texture) -> shading
    normal -> MC environment shading with shadow rays -> bilateral denoiser -> composite -> antialias -> every loss term of
    geometry/gshell_tets_geometry.py:257-384 (FlexiCubes: + L_dev, gshell_flexicubes_geometry.py:358) -> backward to EVERY trainable tensor

HIP through the drop-in API vs  oracle/mlp_oracle -> mtets_oracle | flexi_oracle -> pipeline_oracle.render_mesh -> tick_oracle.tick  on the
state oracle/chain_recipe.py rebuilds from the chain's name (regenerated tensors are re-checked against the fixture's checksums), with the same
noise tensors, sampler seed, stratification table, target and eikonal samples on both sides.  Chains: BASELINE configs[0] (two schedule points),
FlexiCubes res 32 (with / without the open regulariser), a small two-view textured scene, configs[1] at its REAL size (6 / 16 texture levels),
the configs[4] extractor at its grid size (res 80) and configs[2] -- the HEADLINE: tet-res256, 4 x 512^2, n = 8, 16-level texture.

Checked per chain
  * the SDF network's forward: the sign of EVERY grid row equals the fixture's, the values that are consumed (end points of sign-crossing
    edges) agree to 1e-6.  Those rows then carry the fixture's float32 values and the product's graph (straight-through substitution
    s_fix + (s - s.detach()) -- exactly s_fix in float32), so both sides extract from IDENTICAL fields: with each side's own values (2e-7 apart)
    the reference's L_dev = | |u_e - v_d| - mean | regulariser (gshell_flexicubes.py:232-240) flips the sign of a few of its ~10^4 kinks and
    d loss / d sdf of two float32 evaluations differs by 9e-4 at res 32 (measured, gpurun_out r06b; tools/flexi_grad_diag.py shows the
    extraction kernels' gradients agree with float64 as well as the float32 oracle's on identical inputs);
  * mesh: topology bit-exact, vertices / augmented mSDF to 1e-6 (G-FlexiCubes: float-atomic sums, 2e-5).  The render stages are then compared
    on the SAME mesh values (the same substitution at the extractor's output), so every later difference is the render stages' own;
  * every buffer pixel by pixel: 1e-4 of the buffer's scale + the fixture's 16-bit rounding (<= 1.53e-5, chain_recipe.quantise), with the
    Monte-Carlo stage's discrete flips counted as filter footprints against an a-priori rate;
  * both losses to 1e-4;
  * every parameter gradient against the FLOAT64 run of the oracle chain: relative L2 <= max(1e-4, REL_FACTOR x the float32 oracle's own distance from the
    float64 run) -- the bar is the north star's 1e-4 wherever float32 defines the quantity that well, and where it does not (the 16-level texture's
    slope jumps at cell faces of 1/4096 of the box; the network's output bias = one signed sum over all rows) it is MEASURED, not hand-set.
    Flipped Monte-Carlo samples (counted above) add, for the tensors that are linear images of d loss / d v_pos, the error share measured within 3 px
    of them (capped by twice the gradient that lives there), for the appearance tensors an a-priori 4 sqrt(flips / covered pixels) / (2 n^2); the ten
    worst vertices of the heavy-tailed position gradient are set aside, printed, and carried only where they exceed what the float32 oracle explains.

GSHELL_LIVE_ORACLE=1 re-mints each fixture in-process (minutes of CPU per config-size chain) instead of reading the committed file."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import chain_recipe as cr

pytestmark = pytest.mark.gpu
DEV = "cuda"
LIVE = os.environ.get("GSHELL_LIVE_ORACLE") == "1"
# A gradient must agree with the float64 chain to 1e-4, or -- where the float32 ORACLE itself is further than that from float64 -- to within this factor
# of the float32 oracle's own distance ("the same order"): the oracle sums in torch's pairwise / sequential order on the CPU, the kernels with float
# atomics in arbitrary order, so the two float32 evaluations are not equally far from float64 (measured ratio on MI355X: 0.9 - 2.1, see
# profiles/r06_chain_parity.txt).
REL_FACTOR = 4.0


class _ConstantMaterial(torch.nn.Module):
    """configs[0] 'constant kd': duck-types MLPTexture3D.sample (render/mlptexture.py:87) with a fixed kd|ks vector."""

    def __init__(self, value):
        super().__init__()
        self.value = torch.nn.Parameter(value.to(DEV).clone())
        self.encoder = type("E", (), {"params": self.value})()

    def sample(self, texc, mask=None):
        return self.value.expand(*texc.shape[:-1], 6)


class _SubstitutingExtractor:
    """The product's extractor with its float outputs' VALUES replaced by the fixture's (graph kept); records what it really produced."""

    def __init__(self, inner, fx, flexi):
        self.inner, self.fx, self.flexi, self.seen = inner, fx, flexi, None

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def __call__(self, *a, **k):
        out = self.inner(*a, **k)
        verts, faces, extra = (out[0], out[1], out[3]) if self.flexi else (out[0], out[1], out[5])
        self.seen = (verts.detach().clone(), faces.clone(), extra['msdf'].detach().clone())
        if tuple(verts.shape) != tuple(self.fx['v'].shape) or not torch.equal(faces.cpu().long(), self.fx['f']):
            return out                                   # reported by the caller as a topology mismatch
        v_sub = self.fx['v'].to(verts.device) + (verts - verts.detach())
        extra = dict(extra)
        extra['msdf'] = self.fx['msdf_aug'].to(verts.device).reshape(extra['msdf'].shape) + (extra['msdf'] - extra['msdf'].detach())
        self.v_sub, self.m_sub = v_sub, extra['msdf']
        v_sub.retain_grad()
        return (v_sub, faces, out[2], extra) if self.flexi else (v_sub, faces, out[2], out[3], out[4], extra)


def load_fixture(name):
    if LIVE:
        from oracle import make_golden_chain as mg
        path = mg.mint(name, tempfile.mkdtemp(prefix="gshell_chain_"))
    else:
        path = os.path.join(cr.GOLDEN, f"chain_{name}.npz")
        if not os.path.isfile(path):
            pytest.skip(f"{path} not minted (python -m oracle.make_golden_chain {name})")
    z = np.load(path)
    fx = {'z': z}
    n_v = int(z['n_verts'])
    v = torch.zeros(n_v, 3)
    used = torch.from_numpy(z['used_idx'].astype(np.int64))
    v[used] = torch.from_numpy(z['verts_used'])
    if 'verts_unused' in z.files:
        mask = torch.ones(n_v, dtype=torch.bool)
        mask[used] = False
        v[mask] = torch.from_numpy(z['verts_unused'])
    g64 = torch.zeros(n_v, 3)
    g64[used] = torch.from_numpy(z['g_v_pos64'])
    fx.update(v=v, used=used, f=torch.from_numpy(z['faces'].astype(np.int64)), msdf_aug=torch.from_numpy(z['msdf_aug']), g_v64=g64)
    return fx


def fixture_grad(z, name):
    """float64-run gradient `name` as stored: ('dense', tensor) | ('sketch', sketch64, sketch32, norm64)"""
    if f"gradsk_{name}" in z.files:
        return 'sketch', torch.from_numpy(z[f"gradsk_{name}"]), torch.from_numpy(z[f"gradsk32_{name}"]), float(z[f"gradnorm_{name}"])
    if f"gradrows_{name}" in z.files:
        shape = tuple(int(s) for s in z[f"gradshape_{name}"])
        g = torch.zeros(shape[0], int(np.prod(shape[1:])) if len(shape) > 1 else 1)
        g[torch.from_numpy(z[f"gradrows_{name}"].astype(np.int64))] = torch.from_numpy(z[f"gradvals_{name}"]).reshape(-1, g.shape[1])
        return 'dense', g.reshape(shape)
    return 'dense', torch.from_numpy(z[f"grad_{name}"])


def build_product(sc, fx):
    """The product's trainer in the fixture's state."""
    from gshell_amd import workload
    torch.manual_seed(0)
    overrides = dict(cr.CHAINS[sc['name']].get('flags', {}))
    tr = workload.build(res=sc['res'], n_samples=sc['n'], batch=sc['B'], train_res=(sc['H'], sc['W']), fit_steps=0, geometry=sc['kind'], **overrides)
    for k, v in vars(sc['flags']).items():
        assert getattr(tr.FLAGS, k) == v, f"FLAGS.{k}: product {getattr(tr.FLAGS, k)} vs recipe {v}"
    g = tr.geometry
    with torch.no_grad():
        assert tuple(g.verts.shape) == tuple(sc['verts'].shape) and torch.equal(g.indices.cpu(), sc['indices'])
        assert float((g.verts.cpu() - sc['verts']).abs().max()) <= 1e-6            # the product centres the grid with a device reduction: last bits
        g.verts.copy_(sc['verts'])
        md = float(g.max_displacement)
        assert abs(md - sc['max_displacement']) <= 1e-6 * abs(md)
        g.max_displacement = torch.tensor(sc['max_displacement'], dtype=torch.float32, device=DEV) if torch.is_tensor(g.max_displacement) else sc['max_displacement']
        missing = g.sdf_net.load_state_dict({k: v for k, v in sc['sdf_net'].items()})
        assert not missing.missing_keys and not missing.unexpected_keys
        g.deform.copy_(sc['deform'])
        g.msdf.copy_(sc['msdf'])
        if sc['kind'] == 'flexicubes':
            g.per_cube_weights.copy_(sc['cube_w'])
        tr.lgt.base.copy_(sc['light'])
        if sc['textured']:
            tex = tr.mat['kd_ks']
            assert tuple(tex.encoder.cfg) == tuple(cr.TEX_CFG) or np.allclose(tex.encoder.cfg, cr.TEX_CFG)
            tex.encoder.params.copy_(sc['tex_params'])
            lin = [mm for mm in tex.net.net if isinstance(mm, torch.nn.Linear)]
            for mm, w in zip(lin, sc['tex_w']):
                mm.weight.copy_(w)
            for a, b in zip(tex.AABB, sc['aabb']):
                assert float((a.cpu() - b).abs().max()) <= 1e-6
            tex.AABB = tuple(b.to(DEV) for b in sc['aabb'])
            for a, b in zip(tex.min_max, sc['min_max']):
                assert torch.equal(a.cpu(), b)
        else:
            tr.mat['kd_ks'] = _ConstantMaterial(sc['material'])
            tr.mat_params = list(tr.mat['kd_ks'].parameters())
    tr.lgt.update_pdf()
    return tr


def _run_chain(name):
    from gshell_amd.geometry import gshell_tets_geometry as geo_mod, mlp as mlp_mod
    from gshell_amd.render import optixutils as ou, render
    fx = load_fixture(name)
    z = fx['z']
    sc = cr.inputs(name)
    cs = cr.checksums(sc)
    for k, val in zip(z['checksums_keys'], z['checksums_vals']):
        if k == 'mvp':
            continue                                                            # cameras are read from the fixture
        assert cs[str(k)] == float(val), f"regenerated input `{k}` differs from the one the fixture was minted with ({cs[str(k)]!r} vs {float(val)!r})"
    kind, B, n, H, W, iteration = sc['kind'], sc['B'], sc['n'], sc['H'], sc['W'], sc['iteration']
    flexi = kind == 'flexicubes'
    tr = build_product(sc, fx)
    g = tr.geometry
    target = {'mvp': torch.from_numpy(z['mvp']).to(DEV), 'campos': torch.from_numpy(z['campos']).to(DEV), 'resolution': [H, W], 'spp': 1,
              'background': sc['background'].to(DEV), 'img': torch.from_numpy(z['target_img']).float().to(DEV)}
    pts = torch.from_numpy(z['sampled_pts']).float().to(DEV)
    sigma = sc['sigma']

    # ---- HIP: the product's tick through the drop-in API
    ext_name = 'gflexicubes' if flexi else 'gshell_tets'
    proxy = _SubstitutingExtractor(getattr(g, ext_name), fx, flexi)
    setattr(g, ext_name, proxy)
    captured = {}
    inner_render = g.render

    def spy(*a, **k):
        captured['d'] = inner_render(*a, **k)
        return captured['d']
    g.render = spy
    # the SDF values where a value is consumed (end points of sign-crossing edges) carry the fixture's float32 values, the product's graph
    rows = torch.from_numpy(z['g_sdf64_rows'].astype(np.int64)).to(DEV)
    vals = torch.from_numpy(z['sdf32_vals']).to(DEV)
    seen_sdf = {}
    inner_sdf = g._sdf_values

    def sdf_with_fixture_values(v_deformed):
        sdf_own = inner_sdf(v_deformed)
        seen_sdf['sdf'] = sdf_own.detach().clone()
        fixed = sdf_own.detach().clone()
        fixed.view(-1)[rows] = vals
        return fixed + (sdf_own - sdf_own.detach())
    g._sdf_values = sdf_with_fixture_values
    old_sampler = geo_mod.sample_points_detached
    geo_mod.sample_points_detached = lambda v_pos, faces, n_pts, generator=None: (pts, None)
    ou.set_random_perm(n, sc['perms'].to(DEV))
    render.noise_override = {k: v.to(DEV) for k, v in sc['noise'].items()}
    render.rnd_seed = sc['seed']
    tr.FLAGS.noise_stream.set_iteration(iteration, None)
    for p in tr.all_params() + tr.mat_params:
        p.grad = None
    try:
        img, depth, reg = g.tick(tr.glctx, target, tr.lgt, tr.mat, tr.loss_fn, iteration, denoiser=tr.denoiser)
    finally:
        render.noise_override = None
        g.render = inner_render
        del g._sdf_values                        # back to the class's method
        geo_mod.sample_points_detached = old_sampler
        setattr(g, ext_name, proxy.inner)
    d = captured['d']
    assert not mlp_mod.FALLBACKS, mlp_mod.FALLBACKS

    # ---- the SDF network's own forward values: every sign, and the values that are consumed
    sdf_own = seen_sdf['sdf'].reshape(-1).cpu()
    sign_fix = torch.from_numpy(np.unpackbits(z['sdf_sign_bits'])[:sc['N']].astype(bool))
    n_sign = int(((sdf_own > 0) != sign_fix).sum())
    d_sdf = float((sdf_own[rows.cpu()] - vals.cpu()).abs().max())
    print(f"\n  chain {name}: SDF network over {sc['N']} rows: {n_sign} sign differences; max |sdf - oracle| on the {rows.numel()} rows whose value is consumed = {d_sdf:.2e}")
    assert n_sign == 0 and d_sdf <= 1e-6, (n_sign, d_sdf)

    # ---- mesh (what the extractor really produced from those values, before the mesh substitution)
    v_hip, f_hip, m_hip = proxy.seen
    assert torch.equal(f_hip.cpu().long(), fx['f']), "the extracted topology differs from the oracle chain's"
    assert int(d['n_verts_watertight']) == int(z['n_verts_watertight'])
    used = fx['used']
    dv = float((v_hip.cpu() - fx['v'])[used].abs().max())
    dm = float((m_hip.cpu().reshape(-1) - fx['msdf_aug'].reshape(-1))[used].abs().max())
    print(f"  mesh: V_aug={fx['v'].shape[0]} T={fx['f'].shape[0]}; max |v_pos - oracle| = {dv:.2e}, max |msdf - oracle| = {dm:.2e} (referenced vertices)")
    if flexi:
        # dual vertices are ratios of float-atomic sums; boundary vertices on edges whose end points lie on one side of the cut extrapolate
        # without bound and are referenced by no face (tests/test_flexi_gpu.py): what a face references must agree
        assert dv <= 2e-5 and dm <= 5e-5, (dv, dm)
    else:
        # identical SDF values in, the extraction's own float32 arithmetic out (tests/test_fullsize_parity_gpu.py: bit-identical at res 128 / 256)
        bound = 1e-6
        assert dv <= bound and dm <= bound, (dv, dm)
        assert float((v_hip.cpu() - fx['v']).abs().max()) <= bound               # unreferenced slots: zero on both sides
    if d['sdf'].requires_grad and not d['sdf'].is_leaf:
        d['sdf'].retain_grad()                                  # intermediate gradient, to say WHERE a parameter gradient's error comes from
    (img + reg).backward()

    # ---- buffers, pixel by pixel
    bufs = d['buffers']
    assert torch.equal(bufs['visible_triangles'].cpu().long(), torch.from_numpy(z['visible_triangles'].astype(np.int64)))
    keys = [k[4:] for k in z.files if k.startswith('buf_') and not k.endswith('_range') and not k.endswith('_maxdev')]
    shaded = cr.dequantise(z['buf_shaded'], *z['buf_shaded_range'])
    n_cov = int((shaded[..., 3] > 0).sum())
    print(f"  covered pixels {n_cov} of {B * H * W}")
    assert n_cov > (3000 if H >= 256 else 1500)
    # discrete flips of the Monte-Carlo stage (a sample landing in the neighbouring probe texel, choosing the other lobe, a shadow ray grazing an
    # edge): 9e-6 per sample against the reference's own kernel on IDENTICAL g-buffers (tests/test_ray_stage_fullsize_parity_gpu.py); here each side
    # shades its own g-buffer (normals from float-atomic sums; with the hash-grid texture also kd / ks from two float32 evaluations, which steer
    # the lobe choice): a-priori allowance 1e-5 per sample, 4.5e-5 with the texture (measured 2.5e-5 at configs[1]'s size), + 1
    textured = sc['textured']
    max_roots = 1 + int((4.5e-5 if textured else 1e-5) * n_cov * 2 * n * n)
    R = int(np.ceil(2.5 * sigma))                       # the bilateral filter's radius: one differently placed sample reaches (2R+1)^2 pixels
    failures, all_roots = [], []
    for key in keys:
        lo, hi = z[f'buf_{key}_range']
        b = cr.dequantise(z[f'buf_{key}'], lo, hi)
        a = bufs[key].detach().cpu()
        scale = float(b.abs().max()) or 1.0
        bar = 1e-4 + cr.quant_half_step(lo, hi)
        dev_px = ((a - b).abs() - 1e-4 * b.abs()).amax(-1) / scale
        bad = dev_px > bar
        roots, rest = [], bad.clone()
        while rest.any() and len(roots) < max_roots + 8:
            idx = int(torch.argmax(torch.where(rest, dev_px, torch.zeros_like(dev_px))))
            bb, y, x = idx // (H * W), (idx // W) % H, idx % W
            roots.append((bb, y, x, float(dev_px.reshape(-1)[idx])))
            rest[bb, max(0, y - R - 1):y + R + 2, max(0, x - R - 1):x + R + 2] = False
        print(f"  buffer {key}: pixels outside {bar:.2e}: {int(bad.sum())} in {len(roots)} filter footprint(s) {[(bb, y, x, f'{e:.1e}') for bb, y, x, e in roots[:6]]} "
              f"(allowed: {max_roots}); oracle float32 vs float64 max {float(z[f'buf_{key}_f64_maxdev']):.1e}")
        mc = key in ('shaded', 'diffuse_light', 'specular_light')
        if mc:
            all_roots += [r for r in roots if r[:3] not in [q[:3] for q in all_roots]]
        if len(roots) > max_roots or (not mc and int(bad.sum()) > 2 * max_roots):
            failures.append(key)
    assert not failures, failures

    # ---- losses
    img_o, reg_o = float(z['img_loss32']), float(z['reg_loss32'])
    print(f"  img_loss {float(img):.6f} vs {img_o:.6f} (float64 {float(z['img_loss64']):.6f}); reg_loss {float(reg):.6f} vs {reg_o:.6f} (float64 {float(z['reg_loss64']):.6f})")
    print("  oracle terms: " + ", ".join(f"{k} {v:.3e}" for k, v in zip(z['terms32_keys'], z['terms32_vals'])))
    assert abs(float(img) - img_o) <= 1e-4 * abs(img_o)
    assert abs(float(reg) - reg_o) <= 1e-4 * abs(reg_o)
    assert float(depth) == 0.0

    # ---- the position gradient against the float64 run: heavy tail, flipped samples, everything else
    # d loss / d v_pos is heavy-tailed: ~10 vertices with near-singular slopes (an antialiased edge almost parallel to its pixel pair, a texture cell face
    # under the vertex) carry 0.3 - 0.98 of the squared error of ANY float32 evaluation, and which value they take varies from run to run with the order of
    # the float atomics (measured on the headline chain: total 4.8e-2 in one run, 2.1e-1 in the next, 1.8e-2 without those ten in both).  They are set
    # aside (printed, and carried into the bars below where they exceed what the float32 oracle's own distance explains); what is asserted is the rest.
    g_v = proxy.v_sub.grad.detach().cpu()
    g64 = fx['g_v64']
    used_mask = torch.zeros(fx['v'].shape[0], dtype=torch.bool).index_fill_(0, used, True)
    e2 = torch.where(used_mask, (g_v - g64).square().sum(-1), torch.zeros(fx['v'].shape[0]))
    tot = float(g64.square().sum())
    top = torch.topk(e2, min(10, e2.numel()))
    top_share = float((top.values.sum() / tot) ** 0.5)
    e2r = e2.clone()
    e2r[top.indices] = 0.0
    # vertices whose samples a flipped Monte-Carlo sample belongs to: the flagged pixel +- 3 px (its own samples' gradients land there; the denoiser
    # spreads the changed RADIANCE over its footprint, which moves the neighbours' gradients only through the loss's curvature)
    vh = torch.cat((fx['v'], torch.ones(fx['v'].shape[0], 1)), -1)
    near = torch.zeros(fx['v'].shape[0], dtype=torch.bool)
    mvp_c = torch.from_numpy(z['mvp'])
    for bb, y, x, _ in all_roots:
        c = vh @ mvp_c[bb].t()
        pxy = ((c[:, :2] / c[:, 3:4]) * 0.5 + 0.5) * torch.tensor([W, H])
        near |= ((pxy[:, 0] - (x + 0.5)).abs() <= 3) & ((pxy[:, 1] - (y + 0.5)).abs() <= 3) & (c[:, 3] > 0)
    near &= used_mask
    flip_share = float((e2r[near].sum() / tot) ** 0.5)
    flip_cap = 2.0 * float((g64[near].square().sum() / tot) ** 0.5)
    far_rel = float((e2r[~near].sum() / tot) ** 0.5)
    e32_v = float(z['g_v_pos_rel32'])
    tail_excess = max(0.0, 2.0 * top_share - REL_FACTOR * e32_v)      # (2 x: a parameter tensor is a linear image of these ten rows, not a norm-preserving one)
    print(f"  d/d v_pos vs float64: total {float((e2.sum() / tot) ** 0.5):.2e} (float32 oracle: {e32_v:.2e}); its 10 worst vertices carry {top_share:.2e} "
          f"({float(top.values.sum() / e2.sum().clamp_min(1e-300)):.2f} of the squared error; |g| / max |g| there: {[round(float(t), 3) for t in (g64[top.indices].norm(dim=-1) / g64.norm(dim=-1).max())]}); "
          f"without them: {int(near.sum())} vertices within 3 px of the {len(all_roots)} flipped samples carry {flip_share:.2e} (cap: 2 x the gradient living there = {flip_cap:.2e}), "
          f"all others {far_rel:.2e}")
    if 'g_sdf64_rows' in z.files and d['sdf'].grad is not None:
        gs64 = torch.zeros(sc['N'])
        gs64[torch.from_numpy(z['g_sdf64_rows'].astype(np.int64))] = torch.from_numpy(z['g_sdf64_vals'])
        gs = d['sdf'].grad.detach().cpu().reshape(-1)
        print(f"  intermediate d/d sdf (extraction + sign regulariser) vs float64: relative L2 {float((gs - gs64).norm() / gs64.norm()):.2e} (float32 oracle "
              f"{float(z['g_sdf_rel32']):.2e}); sum {float(gs.double().sum()):.6e} vs {float(gs64.double().sum()):.6e}; rows with gradient {int((gs != 0).sum())} vs {int((gs64 != 0).sum())}")
    assert flip_share <= flip_cap + 1e-12, "the error next to the flipped samples exceeds twice the gradient that lives there"
    assert far_rel <= max(1e-4, REL_FACTOR * e32_v), (far_rel, e32_v)
    # what the bars below may add: tensors that are linear images of d loss / d v_pos inherit the flipped samples' measured share and the part of the
    # ten worst vertices the float32 oracle's own distance does not explain; appearance tensors (probe, texture, material) see a flipped sample as ONE of
    # the 2 n^2 samples of one of the n_cov pixels: 4 sqrt(flips / n_cov) / (2 n^2), zero when nothing flipped
    allow_pos = 2.0 * flip_share + tail_excess
    allow_app = 4.0 * (len(all_roots) / max(n_cov, 1)) ** 0.5 / (2 * n * n)
    print(f"  allowances on top of max(1e-4, {REL_FACTOR:.0f} x float32-vs-float64): position-linked tensors + {allow_pos:.2e} (flips {2 * flip_share:.2e}, heavy tail {tail_excess:.2e}); appearance tensors + {allow_app:.2e}")

    # ---- every parameter gradient against the float64 run
    names = [str(s) for s in z['grad_names']]
    e32 = dict(zip(names, [float(x) for x in z['grad_rel32_vals']]))
    prod = {f"sdf_net.{k}": p.grad for k, p in g.sdf_net.named_parameters()}
    prod.update(deform=g.deform.grad, msdf=g.msdf.grad, light=tr.lgt.base.grad)
    if flexi:
        prod['per_cube_weights'] = g.per_cube_weights.grad
    if textured:
        tex = tr.mat['kd_ks']
        prod['tex_params'] = tex.encoder.params.grad / 128.0                     # the reference's x128 backward hook (render/mlptexture.py:31)
        for i, mm in enumerate(mm for mm in tex.net.net if isinstance(mm, torch.nn.Linear)):
            prod[f'tex_w{i}'] = mm.weight.grad
    else:
        prod['material'] = tr.mat['kd_ks'].value.grad
    assert sorted(prod) == sorted(names), (sorted(prod), sorted(names))
    failures = []
    for name_t in names:
        a = prod[name_t]
        assert a is not None, name_t
        a = a.detach().cpu()
        assert torch.isfinite(a).all(), name_t
        stored = fixture_grad(z, name_t)
        pos_linked = name_t.startswith('sdf_net.') or name_t in ('deform', 'msdf', 'per_cube_weights')
        tol = max(1e-4, REL_FACTOR * e32[name_t]) + (allow_pos if pos_linked else allow_app)
        note = ""
        if stored[0] == 'sketch':
            from oracle import make_golden_chain as mg
            sk = mg.sketch(a, mg.sketch_plan(a.numel()))
            relerr = float((sk - stored[1]).norm() / stored[1].norm())
            note = f" (count-sketch of {a.numel()} entries, +-1 %; float32 oracle by the same sketch {float((stored[2] - stored[1]).norm() / stored[1].norm()):.2e})"
        else:
            b = stored[1].reshape(a.shape)
            assert float(b.abs().max()) > 0, name_t
            relerr = float((a - b).norm() / b.norm())
            if name_t == 'light' and all_roots:
                # a flipped sample moves ONE sample's light gradient from a probe texel to its neighbour (or removes it): two texels per flip are set
                # aside -- each bounded by the largest texel gradient of the frame (one sample cannot carry more than the heaviest texel's total)
                e_t = (a - b).reshape(-1, 3).square().sum(-1)
                worst = torch.topk(e_t, min(2 * len(all_roots), e_t.numel()))
                assert float(worst.values.max().sqrt()) <= float(b.reshape(-1, 3).norm(dim=-1).max()), "a set-aside probe texel deviates by more than the heaviest texel"
                e_t[worst.indices] = 0.0
                note = f" (with the {worst.indices.numel()} texels of the {len(all_roots)} flipped samples: {relerr:.2e})"
                relerr = float(e_t.sum().sqrt() / b.norm())
            if b.numel() == 1:
                # the output bias's gradient is the plain SUM of d loss / d sdf over all rows (signs cancel): round-off relative to sum |.|
                cond = float(z['cond_bias']) if 'cond_bias' in z.files else 0.0
                tol = max(tol, 1e-5 * cond)
                note = f" (one signed sum, condition {cond:.0f})"
        print(f"  gradient {name_t}: relative L2 vs float64 {relerr:.2e}; float32 oracle {e32[name_t]:.2e}; bar {tol:.2e}{note}")
        if relerr > tol:
            failures.append((name_t, relerr, tol))
    assert not failures, failures


@pytest.mark.parametrize("name", ["config0_a", "config0_b"])
def test_config0_tick_and_every_parameter_gradient_match_the_oracle_chain(name):
    """BASELINE configs[0]: tet-res64 (BCC 26, 202 800 tets), 1 view 256 x 256, 1 Monte-Carlo sample, constant kd / ks; iteration 500 / 1500."""
    _run_chain(name)


@pytest.mark.parametrize("name", ["flexi32", "flexi32_open"])
def test_flexicubes_tick_and_every_parameter_gradient_match_the_oracle_chain(name):
    """G-FlexiCubes (BASELINE configs[4]'s extractor) at res 32 with the per-cube weights among the parameters.  `_open`: the mSDF "open" Huber
    term (tick :330-336) sums over EVERY entry of the augmented mSDF vector, also over boundary vertices whose value u_a j_a + u_b j_b is
    analytically 0 but whose gradient is a difference of nearly equal float32 sums: the float32 oracle's own distance from float64 is large there
    and the bar follows it (printed)."""
    _run_chain(name)


def test_textured_two_view_tick_and_every_parameter_gradient_match_the_oracle_chain():
    """configs[1]-like at a small size: tet-res 32, TWO views of 128^2, n = 2, the hash-grid + MLP texture of the training path (6 levels)."""
    _run_chain("textured2v")


@pytest.mark.parametrize("name", ["config1_l6", "config1_l16"])
def test_config1_tick_and_every_parameter_gradient_match_the_oracle_chain_at_its_real_size(name):
    """BASELINE configs[1] (reference configs/nerf_chair.json:7-13) AT ITS OWN SIZE: tet-res128, 2 views 512 x 512, n = 4, hash-grid + MLP texture
    with 6 levels / the config's own 16 (piecewise trilinear with cells of 1/4096 of the box: d texture / d position jumps at every cell face, two
    float32 evaluations of the position gradient agree to ~1e-2 only -- the float64 arbiter measures exactly that)."""
    _run_chain(name)


def test_flexicubes_tick_chain_at_config4_grid_size_res80():
    """BASELINE configs[4]'s extractor at ITS grid size (reference configs/deepfashion_mc_80.json:17: 80^3 cubes), one 512 x 512 view, n = 2."""
    _run_chain("flexi80")


@pytest.mark.parametrize("name", ["config2", "config2_l6"])
def test_headline_config2_tick_and_every_parameter_gradient_match_the_oracle_chain(name):
    """BASELINE configs[2], the config the benchmark's number is quoted on: tet-res256 (2.28 M grid vertices, 13.4 M tets), 4 views 512 x 512,
    n = 8 (128 shadow rays per covered pixel and pass), 16-level hash-grid texture, steady-state schedule (iteration 1500: full shadows, 23 x 23
    denoiser).  `_l6`: the same state and frames with the texture's 6 coarse levels only -- there float32 defines the position-linked gradients to a few 1e-4,
    so the headline geometry is held to bars with power; with the config's own 16 levels the float64 arbiter measures ~3e-2 for them."""
    _run_chain(name)
