"""Scenes of the end-to-end oracle chains (TEST INFRASTRUCTURE -- checker only; never imported by gshell_amd/).

One training iteration's `tick` (reference geometry/gshell_tets_geometry.py:257-384 / gshell_flexicubes_geometry.py:237-364) needs a complete
state: grid, SDF network, deform, mSDF, (per-cube weights), probe, material, cameras, target, the three noise tensors of render/render.py
:55,:68,:265, the stratification table of optixutils/ops.py:89 and the eikonal surface samples.  This module builds that state ON THE CPU,
bit-reproducibly (IEEE +,-,*,/ in numpy float64 rounded once to float32, and torch's CPU generator), from a chain's NAME:

    oracle/make_golden_chain.py   mints tests/golden/chain_<name>.npz from it (no GPU anywhere),
    tests/test_config0_end_to_end_gpu.py   rebuilds the same state, loads it into the product and compares the HIP `tick` with the fixture.

Only what cannot be regenerated is stored in a fixture: the fitted SDF network (tests/golden/chain_sdf_net.npz, shared by all chains), the
target images and the eikonal sample points (both rounded to float16: they are INPUTS of both sides)."""
import math
import os
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NET_FILE = os.path.join(GOLDEN, "chain_sdf_net.npz")

# name -> what the chain runs.  `iteration` picks the schedule point (shadow ramp, denoiser sigma, eikonal / sdf-regulariser weights),
# `seed` the sampler's rnd_seed.  configs[k] = BASELINE.json's list.
CHAINS = {
    # configs[0]: tet-res64, 1 view 256^2, 1 MC sample, constant kd
    "config0_a": dict(kind="tets", res=64, iteration=500, seed=23, B=1, n=1, frame=256, textured=False),
    "config0_b": dict(kind="tets", res=64, iteration=1500, seed=5, B=1, n=1, frame=256, textured=False),
    # G-FlexiCubes res 32 with / without the mSDF "open" regulariser (tick :330-336)
    "flexi32_open": dict(kind="flexicubes", res=32, iteration=500, seed=31, B=1, n=1, frame=256, textured=False),
    "flexi32": dict(kind="flexicubes", res=32, iteration=500, seed=31, B=1, n=1, frame=256, textured=False, flags=dict(msdf_reg_open_scale=0.0)),
    # configs[1]-like at a small size: two views, hash-grid + MLP texture
    "textured2v": dict(kind="tets", res=32, iteration=500, seed=41, B=2, n=2, frame=128, textured=True, tex_levels=6),
    # configs[1] at its real size (reference configs/nerf_chair.json:7-13), texture with 6 / all 16 hash-grid levels
    "config1_l6": dict(kind="tets", res=128, iteration=500, seed=43, B=2, n=4, frame=512, textured=True, tex_levels=6),
    "config1_l16": dict(kind="tets", res=128, iteration=500, seed=43, B=2, n=4, frame=512, textured=True, tex_levels=16),
    # configs[4]'s extractor at its grid size (reference configs/deepfashion_mc_80.json:17)
    "flexi80": dict(kind="flexicubes", res=80, iteration=500, seed=37, B=1, n=2, frame=512, textured=False, flags=dict(msdf_reg_open_scale=0.0)),
    # configs[2], the HEADLINE: tet-res256, 4 views 512^2, n = 8 (128 shadow rays / covered pixel / pass), the config's own 16-level texture
    "config2": dict(kind="tets", res=256, iteration=1500, seed=47, B=4, n=8, frame=512, textured=True, tex_levels=16),
    # the same frames with the texture's 6 coarse levels only: position-linked gradients defined to ~3e-4 instead of ~3e-2, i.e. the headline GEOMETRY
    # (13.4 M tets, 2.3 10^5 one-pixel triangles, 18.8 M samples) under bars that have power
    "config2_l6": dict(kind="tets", res=256, iteration=1500, seed=47, B=4, n=8, frame=512, textured=True, tex_levels=6),
}
VIEW_IDS = [3, 11, 20, 41]
PERM_ROWS = 32768                      # optixutils/ops.py:89
MESH_SCALE = 1.4                       # train_gshelltet_deepfashion.py:559
BCC_CELLS = {64: 26, 128: 52, 256: 104}
CONSTANT_MATERIAL = [0.6, 0.5, 0.4, 0.0, 0.4, 0.1]
TEX_CFG = (16, 2, 19, 16, float(np.exp(np.log(4096 / 16) / 15)))        # render/mlptexture.py:57-69
KD_KS_MIN = [0.0, 0.0, 0.0, 0.0, 0.001, 0.0]                             # train script :571-574 (kd_min[0:3] | ks_min)
KD_KS_MAX = [1.0, 1.0, 1.0, 0.0, 1.0, 1.0]


def flags(**overrides):
    """The FLAGS fields the reference's `tick` reads (train_gshelltet_deepfashion.py:540-594)."""
    F = types.SimpleNamespace(iter=5000, use_sdf_mlp=True, use_eikonal=True, eikonal_scale=None, use_mesh_msdf_reg=True, msdf_reg_open_scale=1e-6,
                              msdf_reg_close_scale=3e-6, sdf_regularizer=0.2, lambda_kd=0.1, lambda_ks=0.05, lambda_nrm=0.025, lambda_chroma=0.0,
                              lambda_diffuse=0.15, lambda_specular=0.0025)
    for k, v in overrides.items():
        setattr(F, k, v)
    return F


def _gen(seed):
    return torch.Generator(device="cpu").manual_seed(int(seed))


def _f32(a64):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a64, dtype=np.float64).astype(np.float32)))


def skirt_sdf(x):
    """capped cone, positive outside (the state the benchmark's network is fitted to: gshell_amd/workload.py)"""
    r = torch.sqrt(x[:, 0] ** 2 + x[:, 2] ** 2)
    return -torch.minimum(0.42 - 0.22 * x[:, 1] - r, 0.5 - x[:, 1].abs())


def grid(kind, res):
    """-> verts [N,3] float32 AS THE GEOMETRY MODULE HOLDS THEM (centred, x mesh_scale: gshell_tets_geometry.py:62-64), indices, max_displacement"""
    if kind == "tets":
        from gshell_amd import grid as gridlib             # pure torch-CPU generator of the synthetic tet grids (also the oracle goldens' source)
        v, idx = gridlib.bcc_grid(BCC_CELLS.get(res, max(2, int(round(res * 26 / 64)))))
        v64 = v.numpy().astype(np.float64)
        mean32 = v64.mean(0).astype(np.float32)            # float64 pairwise mean, rounded once
        verts = torch.from_numpy((v.numpy() - mean32[None]) * np.float32(MESH_SCALE))
        return verts, idx.long(), 1.0 / res * MESH_SCALE / 2.1          # gshell_tets_geometry.py:155
    from oracle import flexi_oracle as fo
    v, cubes = fo.construct_voxel_grid(res)
    verts = v * MESH_SCALE
    topo = fo.build_topology(cubes, verts.shape[0])
    e = topo["edges"]
    d = (verts[e[:, 0]].double() - verts[e[:, 1]].double()).norm(dim=-1)
    return verts, cubes, float(np.float32(float(d.mean()) / 4))         # gshell_flexicubes_geometry.py:129


def msdf_field(verts):
    """open top + a wavy cut line: 0.32 - y + 0.05 T5(x / 0.7), T5 the Chebyshev polynomial (+,-,* only: bit-reproducible)"""
    v = verts.numpy().astype(np.float64)
    u = v[:, 0] / 0.7
    t5 = ((16.0 * u * u - 20.0) * u * u + 5.0) * u
    return _f32(np.clip(0.32 - v[:, 1] + 0.05 * t5, -2.0, 2.0))


def camera(k, radius=2.2, fovy_deg=60.0, yaw=0.0):
    """pose k of the benchmark's 72-pose orbit (gshell_amd/workload.py:camera), optionally yawed (the target's camera)"""
    k = k % 72
    az = 2 * math.pi * ((k * 0.61803398875) % 1.0) + yaw
    el = math.radians(-25.0 + 50.0 * ((k * 0.41421356237 + 0.25) % 1.0))
    eye = radius * np.array([math.cos(el) * math.sin(az), math.sin(el), math.cos(el) * math.cos(az)])
    y = math.tan(math.radians(fovy_deg) / 2)
    n, f = 0.1, 1000.0
    proj = np.array([[1 / y, 0, 0, 0], [0, -1 / y, 0, 0], [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]], dtype=np.float64)
    w = eye / np.linalg.norm(eye)
    u = np.cross(np.array([0.0, 1.0, 0.0]), w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    mv = np.eye(4)
    mv[0, :3], mv[1, :3], mv[2, :3] = u, v, w
    mv[:3, 3] = -mv[:3, :3] @ eye
    return (proj @ mv).astype(np.float32), eye.astype(np.float32)


def load_net():
    z = np.load(NET_FILE)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def inputs(name):
    """Everything of chain `name` that is a pure function of the name (see the module docstring).  All tensors torch CPU float32 / int."""
    c = dict(CHAINS[name])
    kind, res, B, n, frame = c["kind"], c["res"], c["B"], c["n"], c["frame"]
    H = W = frame
    verts, indices, max_disp = grid(kind, res)
    N = verts.shape[0]
    sc = dict(c, name=name, H=H, W=W, verts=verts, indices=indices, max_displacement=max_disp, N=N)
    sc["flags"] = flags(**c.get("flags", {}))
    sc["deform"] = (torch.rand(N, 3, generator=_gen(1)) * 2 - 1) * 0.3
    sc["msdf"] = msdf_field(verts)
    sc["sdf_net"] = load_net() if os.path.isfile(NET_FILE) else None
    if kind == "flexicubes":
        sc["cube_w"] = torch.ones(indices.shape[0], 21)                    # gshell_flexicubes_geometry.py:95 (the reference's initial value)
    sc["light"] = torch.rand(256, 256, 3, generator=_gen(5)) * 0.8 + 0.2
    if c["textured"]:
        from oracle import hashgrid_oracle as ho
        metas, total = ho.level_meta(*TEX_CFG)
        g = _gen(7)
        p = (torch.rand(total, generator=g) * 2 - 1) * 0.3                 # tcnn's U(-1e-4, 1e-4) x 3000: a texture with visible structure
        L = c["tex_levels"]
        if L < len(metas):
            p[metas[L][2] * TEX_CFG[1]:] = 0.0
        sc["tex_params"] = p
        sc["tex_w"] = [(torch.rand(o, i, generator=g) * 2 - 1) * math.sqrt(6.0 / i) for o, i in ((32, 32), (32, 32), (6, 32))]   # kaiming_uniform(relu)
        sc["aabb"] = (verts.min(0).values, verts.max(0).values)            # geometry.getAABB()
        sc["min_max"] = (torch.tensor(KD_KS_MIN), torch.tensor(KD_KS_MAX))
    else:
        sc["material"] = torch.tensor(CONSTANT_MATERIAL)
    g = _gen(11)
    sc["noise"] = {"jitter": torch.randn(B, H, W, 2, generator=g) * 0.005, "texture": torch.randn(B, H, W, 3, generator=g) * 0.01,
                   "tangent": torch.randn(B, H, W, 3, generator=g)}
    sc["perms"] = torch.argsort(torch.rand(PERM_ROWS, n * n, generator=g), dim=-1).int()
    cams = [camera(k) for k in VIEW_IDS[:B]]
    sc["mvp"] = torch.from_numpy(np.stack([m for m, _ in cams]))
    sc["campos"] = torch.from_numpy(np.stack([e for _, e in cams]))
    tcams = [camera(k, yaw=math.radians(1.5)) for k in VIEW_IDS[:B]]
    sc["target_mvp"] = torch.from_numpy(np.stack([m for m, _ in tcams]))
    sc["target_campos"] = torch.from_numpy(np.stack([e for _, e in tcams]))
    sc["background"] = torch.stack([torch.rand(1, 1, 3, generator=_gen(7919 + v)) for v in VIEW_IDS[:B]]).expand(B, H, W, 3).contiguous()
    sc["shadow"] = min(c["iteration"] / 1000, 1.0)
    sc["sigma"] = 2.0 * sc["shadow"]                                       # denoiser.py: sigma = max(2 * influence, 1e-4)
    return sc


def checksums(sc):
    """float64 sums of the regenerated tensors: stored in the fixture at mint time, re-checked where the fixture is used (a different
    torch build drawing different random numbers must fail HERE, not as a parity error 300 lines later)"""
    out = {}
    for k in ("verts", "deform", "msdf", "light", "tex_params", "perms", "background", "mvp"):
        if k in sc and sc[k] is not None:
            out[k] = float(sc[k].double().sum())
    for k, v in sc["noise"].items():
        out["noise_" + k] = float(v.double().abs().sum())
    if "tex_w" in sc:
        out["tex_w"] = float(sum(w.double().abs().sum() for w in sc["tex_w"]))
    return out


# ---- fixture encoding ---------------------------------------------------------------------------------------------------------------
def quantise(buf):
    """[B,H,W,C] float -> (uint16 codes, lo, hi): value = lo + code / 65535 * (hi - lo).  The chain tests compare a buffer relative to its
    largest magnitude, so a uniform 16-bit grid over the buffer's own range costs at most half a step = (hi - lo) / 131070 <= 1.53e-5 of that
    scale (7.6e-6 for the non-negative buffers) -- added to the 1e-4 bar where the fixtures are compared (QUANT_HALF_STEP)."""
    a = buf.detach().double().numpy()
    lo, hi = float(a.min()), float(a.max())
    if hi <= lo:
        hi = lo + 1.0
    q = np.rint((a - lo) / (hi - lo) * 65535.0).astype(np.uint16)
    return q, lo, hi


def dequantise(q, lo, hi):
    return torch.from_numpy((lo + q.astype(np.float64) / 65535.0 * (hi - lo)).astype(np.float32))


def quant_half_step(lo, hi):
    """worst-case rounding error of `quantise`, relative to the buffer's largest magnitude"""
    return (hi - lo) / 131070.0 / (max(abs(lo), abs(hi)) or 1.0)
