"""Mint golden vectors for G-FlexiCubes by running the REAL reference (geometry/gshell_flexicubes.py:136-230) on CPU
(build container only):   python -m oracle.make_golden_flexi
Writes tests/golden/flexi_*.npz: inputs (recipe), every output of `__call__`, and input gradients of a fixed weighted sum."""
import os

import numpy as np
import torch

from oracle import refload

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_inputs(res, sdf_kind, msdf_kind, weights_kind, seed):
    """Deterministic inputs (numpy PCG64) on the reference's own voxel grid layout."""
    rng = np.random.default_rng(seed)
    n = res + 1
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    grid = (np.stack([i, j, k], -1).reshape(-1, 3) / res - 0.5).astype(np.float32)      # same order as construct_voxel_grid
    x = grid + ((rng.random(grid.shape) - 0.5) * 0.3 / res).astype(np.float32)
    r = np.linalg.norm(grid.astype(np.float64), axis=1)
    if sdf_kind == "sphere":
        s = r - 0.33
    elif sdf_kind == "noisy":
        s = r - 0.31 + 0.06 * rng.standard_normal(r.shape)
    elif sdf_kind == "two":
        s = np.minimum(np.linalg.norm(grid - np.array([0.17, 0.02, -0.03]), axis=1) - 0.2, np.linalg.norm(grid + np.array([0.16, 0.04, 0.0]), axis=1) - 0.18)
    elif sdf_kind == "empty":
        s = np.ones_like(r)
    s = s.astype(np.float32)
    if msdf_kind == "positive":
        nu = 0.5 + 0.2 * rng.random(r.shape)
    elif msdf_kind == "half":
        nu = 0.05 - grid[:, 1] + 0.2 * grid[:, 0]
    elif msdf_kind == "rand":
        nu = np.clip(rng.random(r.shape) - 0.3, -1, 1)
    elif msdf_kind == "negative":
        nu = -0.5 - 0.1 * rng.random(r.shape)
    nu = nu.astype(np.float32)
    F = res ** 3
    w = np.zeros((F, 21), np.float32) + 1.0 if weights_kind == "ones" else (rng.standard_normal((F, 21)) * 0.8).astype(np.float32)
    return x, s, nu, w


CASES = [("r6_sphere_pos_ones", 6, "sphere", "positive", "ones", 1), ("r6_noisy_half_rand", 6, "noisy", "half", "rand", 2),
         ("r8_two_rand_rand", 8, "two", "rand", "rand", 3), ("r10_noisy_rand_rand", 10, "noisy", "rand", "rand", 4),
         ("r7_sphere_negative_ones", 7, "sphere", "negative", "ones", 5), ("r5_empty", 5, "empty", "positive", "ones", 6)]


# training=True (gshell_flexicubes.py:523-551: every quad becomes a fan of four triangles around a gamma-weighted centre vertex) -> tests/golden/flexitrain_*.npz
TRAIN_CASES = [("r6_noisy_half_rand", 6, "noisy", "half", "rand", 2), ("r8_two_rand_rand", 8, "two", "rand", "rand", 3), ("r10_noisy_rand_rand", 10, "noisy", "rand", "rand", 4),
               ("r7_sphere_negative_ones", 7, "sphere", "negative", "ones", 5)]


def main():
    mod = refload.load_flexicubes()
    with refload.CudaToCpu():
        fc = mod.GShellFlexiCubes(device="cpu")
        for name, res, sk, mk, wk, seed, training in [c + (False,) for c in CASES] + [c + (True,) for c in TRAIN_CASES]:
            verts, cubes = fc.construct_voxel_grid(res)
            x, s, nu, w = make_inputs(res, sk, mk, wk, seed)
            # the recipe's grid order must be the reference's own vertex order
            assert np.allclose(verts.numpy(), (x * 0 + (np.stack(np.meshgrid(*(np.arange(res + 1),) * 3, indexing="ij"), -1).reshape(-1, 3) / res - 0.5)), atol=1e-5)
            X, S, NU, Wt = (torch.tensor(a, requires_grad=True) for a in (x, s[:, None], nu, w))
            out = fc(X, S, NU, cubes, res, Wt[:, :12], Wt[:, 12:20], Wt[:, 20], training=training)
            rec = dict(res=res, sdf_kind=sk, msdf_kind=mk, weights_kind=wk, seed=seed, cubes=cubes.numpy())
            if len(out) == 3:           # empty surface early return (reference :193-202)
                rec.update(empty=True, vertices_open=out[0].numpy(), faces_open=out[1].numpy(), L_dev=out[2].numpy())
            else:
                v, f, L, ex = out
                g = torch.Generator().manual_seed(seed)
                wv, wm, wl = torch.randn(v.shape, generator=g), torch.randn(ex['msdf'].shape, generator=g), torch.randn(L.shape, generator=g)
                loss = (v * wv).sum() + (ex['msdf'] * wm).sum() + (L * wl).sum() + (ex['msdf_watertight'] * 0.3).sum()
                loss.backward()
                rec.update(empty=False, vertices_open=v.detach().numpy(), faces_open=f.numpy(), L_dev=L.detach().numpy(),
                           n_verts_watertight=ex['n_verts_watertight'], vertices_watertight=ex['vertices_watertight'].detach().numpy(),
                           faces_watertight=ex['faces_watertight'].numpy(), msdf=ex['msdf'].detach().numpy(),
                           msdf_watertight=ex['msdf_watertight'].detach().numpy(), msdf_boundary=ex['msdf_boundary'].detach().numpy(),
                           w_v=wv.numpy(), w_m=wm.numpy(), w_l=wl.numpy(),
                           g_x=X.grad.numpy(), g_s=S.grad.numpy(), g_nu=NU.grad.numpy() if NU.grad is not None else np.zeros_like(nu),
                           g_w=Wt.grad.numpy() if Wt.grad is not None else np.zeros_like(w))
                print("training" if training else "", name, "V", v.shape[0], "T", f.shape[0], "Vwt", ex['n_verts_watertight'], "Twt", ex['faces_watertight'].shape[0], "L", L.shape[0])
            np.savez_compressed(os.path.join(OUT, f"{'flexitrain' if training else 'flexi'}_{name}.npz"), **rec)


if __name__ == "__main__":
    main()
