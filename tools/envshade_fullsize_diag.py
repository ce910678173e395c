"""Diagnostics for tests/test_ray_stage_fullsize_parity_gpu.py (b): which pixels of a 512 x 512 env-shade output / gradient differ from the
reference kernel, and what the float64 evaluation of the same samples (oracle/shade_oracle.env_shade on float64 leaves: the sampling
decisions stay float32) says about each side.      python tools/envshade_fullsize_diag.py [res]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import shade_oracle as so       # noqa: E402
import tests.test_ray_stage_fullsize_parity_gpu as T   # noqa: E402
from gshell_amd.render import optixutils as ou   # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
DEV = "cuda"
s = T.build_scene(res)
g, t_ref = T.reference_golden(s)
print(f"reference kernel: {t_ref:.1f} s")
N = T.N
t = {k: torch.tensor(g[k], device=DEV) for k in ("mask", "ro", "gb_pos", "gb_normal", "view_pos", "gb_kd", "gb_ks", "light", "pdf", "rows", "cols")}
ou.set_random_perm(N, torch.tensor(g["perms"].astype(np.int32), device=DEV))
names = ("gb_pos", "gb_normal", "gb_kd", "gb_ks", "light")
leaves = [t[k].clone().requires_grad_(True) for k in names]
args = (s["ctx"], t["mask"], t["ro"], leaves[0], leaves[1], t["view_pos"], leaves[2], leaves[3], leaves[4], t["pdf"], t["rows"], t["cols"])
d, sp = ou.optix_env_shade(*args, BSDF="pbr", n_samples_x=N, rnd_seed=int(g["seed"]), shadow_scale=1.0)
((d * torch.tensor(g["diff_grad"], device=DEV)).sum() + (sp * torch.tensor(g["spec_grad"], device=DEV)).sum()).backward()
mine = dict(diff=d.detach().cpu().numpy(), spec=sp.detach().cpu().numpy())
mine.update({f"g_{k}": l.grad.cpu().numpy() for k, l in zip(names, leaves)})

worst = {}
for k in ("diff", "spec", "g_gb_pos", "g_gb_normal", "g_gb_kd", "g_gb_ks"):
    a, b = mine[k].reshape(-1, 3), g[k].reshape(-1, 3)
    sc = float(np.abs(b).max())
    dev = np.abs(a - b).max(-1) / sc
    order = np.argsort(-dev)[:6]
    worst[k] = order
    print(f"{k}: scale {sc:.4g}; pixels > 1e-4: {int((dev > 1e-4).sum())}, > 2e-4: {int((dev > 2e-4).sum())}, > 1e-3: {int((dev > 1e-3).sum())}; worst {[(int(i), float(dev[i])) for i in order]}")

# float64 evaluation of the worst gradient pixels (same sampling decisions)
pick = np.unique(np.concatenate([worst["g_gb_pos"], worst["g_gb_normal"]]))
m64 = np.zeros_like(g["mask"]).reshape(-1)
m64[pick] = 1
tt = {k: torch.tensor(g[k]).double() for k in ("ro", "gb_pos", "gb_normal", "view_pos", "gb_kd", "gb_ks", "light")}
lv = [tt[k].clone().requires_grad_(True) for k in names]
so.ANY_HIT = so.any_hit_c
d64, s64 = so.env_shade(torch.tensor(m64.reshape(g["mask"].shape)), tt["ro"], lv[0], lv[1], tt["view_pos"], lv[2], lv[3], lv[4], torch.tensor(g["pdf"]),
                                 torch.tensor(g["rows"]), torch.tensor(g["cols"]), g["perms"].astype(np.int32), 0, N, int(g["seed"]), 1.0, g["verts"], g["tris"].astype(np.int64))
((d64 * torch.tensor(g["diff_grad"]).double()).sum() + (s64 * torch.tensor(g["spec_grad"]).double()).sum()).backward()
for k, l in zip(names[:2], lv[:2]):
    g64 = l.grad.numpy().reshape(-1, 3)
    sc = float(np.abs(g[f"g_{k}"]).max())
    for p in pick:
        a, b = mine[f"g_{k}"].reshape(-1, 3)[p], g[f"g_{k}"].reshape(-1, 3)[p]
        print(f"g_{k} pixel {int(p)}: |hip - ref| {np.abs(a - b).max() / sc:.2e}  |hip - f64| {np.abs(a - g64[p]).max() / sc:.2e}  |ref - f64| {np.abs(b - g64[p]).max() / sc:.2e}  "
              f"value {np.abs(g64[p]).max() / sc:.3g} of max; ks {g['gb_ks'].reshape(-1, 3)[p].tolist()} n.v "
              f"{float((g['gb_normal'].reshape(-1, 3)[p] * (g['view_pos'].reshape(-1, 3)[0] - g['gb_pos'].reshape(-1, 3)[p])).sum()):.3g}")

# ---- per-sample anatomy of the worst forward pixels: the same float64 formula fed with either side's (direction, k, visibility) ----------
import math
from tests.test_ref_parity_gpu import _texel
pixs, dirs, kk, live, vis = (x.cpu().numpy() for x in ou.optix_env_shade_samples(s["ctx"], t["mask"], t["ro"], t["gb_pos"], t["gb_normal"], t["view_pos"], t["gb_kd"],
                                                                                 t["gb_ks"], t["light"], t["pdf"], t["rows"], t["cols"], BSDF="pbr", n_samples_x=N,
                                                                                 rnd_seed=int(g["seed"]), shadow_scale=1.0))
S = N * N
ref = g["samples"].reshape(-1, S, 2, 6).transpose(0, 2, 1, 3)
Hl, Wl = g["pdf"].shape
light64 = torch.tensor(g["light"]).double()


def contrib(p, d, k, v):
    """float64 contribution (diff, spec) [2S, 3] of pixel p's samples given directions d [2S,3] (float32 values), weights k, visibility v"""
    nrm = torch.tensor(g["gb_normal"].reshape(-1, 3)[p]).double()[None]
    pos = torch.tensor(g["gb_pos"].reshape(-1, 3)[p]).double()[None]
    kd = torch.tensor(g["gb_kd"].reshape(-1, 3)[p]).double()[None]
    ks = torch.tensor(g["gb_ks"].reshape(-1, 3)[p]).double()[None]
    view = torch.tensor(g["view_pos"].reshape(-1, 3)[0]).double()[None]
    wo = so.t_safe_normalize(view - pos)
    wi = torch.tensor(d).double()
    tex, border = _texel(d, Hl, Wl)
    col = light64.reshape(-1, 3)[torch.as_tensor(tex)]
    dl = so.lambert(nrm, wi).expand(-1, 3)
    spec_col = (0.04 * (1.0 - ks[:, 2:3]) + kd * ks[:, 2:3]) * (1.0 - ks[:, 0:1])
    sl = so.pbr_specular(spec_col, nrm, wo, wi, ks[:, 1:2] * ks[:, 1:2])
    w = torch.tensor(k).double()[:, None] * torch.tensor(v.astype(np.float64))[:, None]
    return (dl * col * w).numpy(), (sl * col * w).numpy(), tex, border


for p in worst["diff"][:3]:
    i = int(np.searchsorted(pixs, p))
    assert pixs[i] == p
    dh, kh, vh, lh = dirs[i].reshape(-1, 3), kk[i].reshape(-1), vis[i].reshape(-1), live[i].reshape(-1)
    r = ref[i].reshape(-1, 6)
    dr_, kr, vr = r[:, :3], (np.float32(1.0) / np.maximum(r[:, 3] + r[:, 4], np.float32(1e-4))) * np.float32(1.0 / S), r[:, 5] > 0
    a_d, a_s, tex_h, bor_h = contrib(p, dh, kh, vh & lh)
    b_d, b_s, tex_r, bor_r = contrib(p, dr_, kr, vr)
    print(f"pixel {int(p)}: product diff {mine['diff'].reshape(-1, 3)[p]} ref {g['diff'].reshape(-1, 3)[p]}; float64 from product samples {a_d.sum(0)} from ref samples {b_d.sum(0)}")
    dev = np.abs(a_d - b_d).max(-1) + np.abs(a_s - b_s).max(-1)
    for j in np.argsort(-dev)[:4]:
        print(f"   sample {int(j)} ({'light' if j < S else 'bsdf'}): |d contribution| {dev[j]:.3e}  dir diff {np.abs(dh[j] - dr_[j]).max():.2e}  k {kh[j]:.6g} / {kr[j]:.6g}  "
              f"vis {bool(vh[j])}/{bool(vr[j])} live {bool(lh[j])}  texel {int(tex_h[j])}/{int(tex_r[j])} border {bor_h[j]:.2e}/{bor_r[j]:.2e}  n.wi {float((g['gb_normal'].reshape(-1, 3)[p] * dr_[j]).sum()):.3e}")

# ---- light gradient: which texels differ, and what float64 sums over either side's samples say -----------------------------------------
gl, gl_ref = mine["g_light"].reshape(-1, 3), g["g_light"].reshape(-1, 3)
sc = float(np.abs(gl_ref).max())
dev = np.abs(gl - gl_ref).max(-1) / sc
print(f"g_light: scale {sc:.4g}; texels > 1e-4: {int((dev > 1e-4).sum())}, > 1e-3: {int((dev > 1e-3).sum())}; worst {[(int(i), float(dev[i])) for i in np.argsort(-dev)[:6]]}")


def light_grad_f64(dd, kk_, vv):
    """float64 light gradient [Hl*Wl, 3] from per-sample (dir [n_cov,2,S,3], k, visibility)"""
    n_cov = dd.shape[0]
    out = np.zeros((Hl * Wl, 3))
    gd, gs = g["diff_grad"].reshape(-1, 3)[pixs].astype(np.float64), g["spec_grad"].reshape(-1, 3)[pixs].astype(np.float64)
    CH = 2048
    for a in range(0, n_cov, CH):
        b = min(a + CH, n_cov)
        P = pixs[a:b]
        nrm = torch.tensor(g["gb_normal"].reshape(-1, 3)[P]).double()[:, None, :]
        pos = torch.tensor(g["gb_pos"].reshape(-1, 3)[P]).double()[:, None, :]
        kd = torch.tensor(g["gb_kd"].reshape(-1, 3)[P]).double()[:, None, :]
        ks = torch.tensor(g["gb_ks"].reshape(-1, 3)[P]).double()[:, None, :]
        view = torch.tensor(g["view_pos"].reshape(-1, 3)[0]).double()[None, None]
        wo = so.t_safe_normalize(view - pos)
        wi = torch.tensor(dd[a:b].reshape(b - a, -1, 3)).double()
        dl = so.lambert(nrm, wi).expand(-1, -1, 3)
        spec_col = (0.04 * (1.0 - ks[..., 2:3]) + kd * ks[..., 2:3]) * (1.0 - ks[..., 0:1])
        sl = so.pbr_specular(spec_col, nrm, wo, wi, ks[..., 1:2] * ks[..., 1:2])
        w = torch.tensor(kk_[a:b].reshape(b - a, -1)).double()[..., None] * torch.tensor(vv[a:b].reshape(b - a, -1).astype(np.float64))[..., None]
        lg = ((torch.tensor(gd[a:b])[:, None] * dl + torch.tensor(gs[a:b])[:, None] * sl) * w).numpy().reshape(-1, 3)
        tex, _ = _texel(dd[a:b].reshape(-1, 3), Hl, Wl)
        np.add.at(out, tex, lg)
    return out


r_dir, r_pl, r_pb, r_vis = ref[..., :3], ref[..., 3], ref[..., 4], ref[..., 5] > 0
r_k = (np.float32(1.0) / np.maximum(r_pl + r_pb, np.float32(1e-4))) * np.float32(1.0 / S)
L_h = light_grad_f64(dirs, kk, vis & live)
L_r = light_grad_f64(r_dir, r_k, r_vis)
for i in np.argsort(-dev)[:8]:
    print(f"  texel {int(i)} (row {int(i) // Wl}, col {int(i) % Wl}): product {gl[i]} ref {gl_ref[i]}; float64 over product samples {L_h[i]} over ref samples {L_r[i]}")
print(f"  |product - f64(product samples)| max {np.abs(gl - L_h).max() / sc:.2e};  |ref - f64(ref samples)| max {np.abs(gl_ref - L_r).max() / sc:.2e};  "
      f"|f64(product samples) - f64(ref samples)| texels > 1e-4: {int((np.abs(L_h - L_r).max(-1) / sc > 1e-4).sum())}")
