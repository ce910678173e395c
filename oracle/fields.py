"""Deterministic synthetic SDF / mSDF fields for parity tests and benchmarks.

TEST/BENCH INFRASTRUCTURE (inputs only; contains no reference algorithm).
All randomness comes from numpy's PCG64 `default_rng(seed)` which is bit-stable
across machines, so the same (grid, kind, seed) gives the same arrays in the
build container (where goldens are minted) and on the GPU box.
"""
import numpy as np

SDF_KINDS = ("sphere", "sphere_noise", "two_spheres", "plane", "skirt")
MSDF_KINDS = ("positive", "rand", "halfspace", "negative", "wavy")


def make_sdf(verts: np.ndarray, kind: str, seed: int = 0, zeros: int = 0) -> np.ndarray:
    """Positive = inside (reference gshell_tets.py:250 `occ_n = sdf_n > 0`)."""
    v = verts.astype(np.float64)
    rng = np.random.default_rng(seed)
    if kind == "sphere":
        s = 0.31 - np.linalg.norm(v, axis=1)
    elif kind == "sphere_noise":
        s = 0.33 - np.linalg.norm(v, axis=1) + 0.04 * rng.standard_normal(v.shape[0])
    elif kind == "two_spheres":
        a = 0.22 - np.linalg.norm(v - np.array([0.14, 0.02, -0.03]), axis=1)
        b = 0.19 - np.linalg.norm(v + np.array([0.15, 0.05, 0.01]), axis=1)
        s = np.maximum(a, b)
    elif kind == "plane":
        s = 0.013 + v @ np.array([0.31, -0.52, 0.79])
    elif kind == "skirt":
        # capped cylinder / "skirt": radius grows towards -y (SURVEY 8d state B)
        r = np.sqrt(v[:, 0] ** 2 + v[:, 2] ** 2)
        s = np.minimum(0.26 - 0.18 * v[:, 1] - r, 0.36 - np.abs(v[:, 1]))
    else:
        raise ValueError(kind)
    s = s.astype(np.float32)
    if zeros:
        idx = rng.choice(s.shape[0], size=min(zeros, s.shape[0]), replace=False)
        s[idx] = 0.0
    return s


def make_msdf(verts: np.ndarray, kind: str, seed: int = 0, zeros: int = 0) -> np.ndarray:
    v = verts.astype(np.float64)
    rng = np.random.default_rng(seed + 7919)
    if kind == "positive":
        m = 0.5 + 0.25 * rng.random(v.shape[0])
    elif kind == "rand":
        # reference init: (rand - 0.01).clamp(-1, 1)  (gshell_tets_geometry.py:139)
        m = np.clip(rng.random(v.shape[0]) - 0.01, -1, 1)
    elif kind == "halfspace":
        m = 0.07 - v[:, 1] + 0.3 * v[:, 0]
    elif kind == "negative":
        m = -0.5 - 0.25 * rng.random(v.shape[0])
    elif kind == "wavy":
        m = 0.12 - v[:, 1] + 0.05 * np.sin(8.0 * v[:, 0]) + 0.02 * rng.standard_normal(v.shape[0])
    else:
        raise ValueError(kind)
    m = m.astype(np.float32)
    if zeros:
        idx = rng.choice(m.shape[0], size=min(zeros, m.shape[0]), replace=False)
        m[idx] = 0.0
    return m


def make_deform(verts: np.ndarray, cell: float, seed: int = 0, amp: float = 0.3) -> np.ndarray:
    rng = np.random.default_rng(seed + 104729)
    return ((rng.random(verts.shape) * 2 - 1) * amp * cell).astype(np.float32)


def loss_weights(n_aug: int, n_wt: int, seed: int = 0):
    """Random cotangents for verts_aug, msdf_aug, vertices_watertight."""
    rng = np.random.default_rng(seed + 15485863)
    return (rng.standard_normal((n_aug, 3)).astype(np.float32),
            rng.standard_normal((n_aug,)).astype(np.float32),
            rng.standard_normal((n_wt, 3)).astype(np.float32))


def discretize_verts(verts: np.ndarray) -> np.ndarray:
    """Integer cells of the 2x denser cubic grid (reference gshell_tets_geometry.py:72-78,
    GMeshDiffusion/metadata/save_tet_info.py:39-44): dx = half the smallest coordinate step."""
    v = verts.astype(np.float32)
    u = np.unique(v.reshape(-1))
    dx = (u[1] - u[0]) / np.float32(2.0)
    return np.floor((v - v.min()) / dx + np.float32(1e-3)).astype(np.int64)


def make_aug_grids(G: int, seed: int, msdf_kind: str = "sign"):
    """Seeded cubic-grid inputs of marching_from_auggrid: crossing coefficients [G^3]
    (deliberately spilling outside [0,1] to exercise the clamp), mSDF sign grid [G^3]
    and the boundary occupancy grid [(2G)^3] in [-1,1]."""
    rng = np.random.RandomState(1000 + seed)
    coeff = rng.uniform(-0.2, 1.2, size=(G, G, G)).astype(np.float32)
    occ = rng.uniform(-1.0, 1.0, size=(2 * G, 2 * G, 2 * G)).astype(np.float32)
    if msdf_kind == "sign":
        m = np.sign(rng.uniform(-0.6, 1.0, size=(G, G, G))).astype(np.float32)
        m[rng.uniform(size=(G, G, G)) < 0.02] = 0.0
    elif msdf_kind == "halfspace":
        z = np.linspace(-1, 1, G, dtype=np.float32)
        m = np.sign(0.2 - z[None, :, None] + 0.3 * z[:, None, None] + 0 * z[None, None, :]).astype(np.float32)
    elif msdf_kind == "positive":
        m = np.ones((G, G, G), np.float32)
    elif msdf_kind == "negative":
        m = -np.ones((G, G, G), np.float32)
    else:
        raise ValueError(msdf_kind)
    return coeff, m, occ
