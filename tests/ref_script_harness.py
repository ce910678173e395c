"""Runs the REFERENCE'S OWN training script (train_gshelltet_deepfashion.py, from /root/reference, unmodified, as __main__) against
gshell_amd.compat on a box WITHOUT a GPU (TEST INFRASTRUCTURE; used by tests/test_compat_cpu.py in a subprocess).

What is stubbed: the out-of-scope imports (xatlas, dataset.*: SURVEY.md section 2) -- a synthetic dataset with the reference's batch
keys and collate (dataset/dataset_deepfashion.py:59-138) stands in --, and `device='cuda'` / `.cuda()` are redirected to the CPU so
that the script's host-side set-up runs.  `render.material`, `render.texture` and `render.obj` are the reference's own python files,
loaded into the shimmed `render` package.  The script then runs: argument parsing, FLAGS, rasteriser context, datasets, trainable
env light, denoiser, GShellTetsGeometry(grid, scale, FLAGS) incl. the SDF-network pre-fit, initial_guess_material, optimize_mesh:
three Adam optimisers over the parameter groups picked BY NAME, the DataLoader, prepare_batch, zero_grad, lgt.update_pdf(),
geometry.tick(...) -- where the first HIP entry point refuses the CPU tensors (there is no CPU fallback) and raises GShellHipError.
usage: python tests/ref_script_harness.py <reference root> <work dir> [cuda [iterations]]

`cuda` (GPU box, tools/run_ref_train_script_gpu.sh): nothing is redirected -- the unmodified script trains on the MI355X through the shim for `iterations`
steps (optimize_mesh :278-497) and runs its own validate() (:227-272).  The synthetic dataset then holds CONSISTENT views: images of a ground-truth
state (the benchmark's capped-cone garment, rendered once by this package's renderer from the dataset's cameras), so the losses fall and the PSNR
of validate() means something.  The reference tree does not exist on the GPU box: <reference root> is a git-ignored scratch copy of the three files
the script needs (train_gshelltet_deepfashion.py, render/material.py, render/texture.py), removed after the run."""
import json
import os
import runpy
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ref_root, work = sys.argv[1], sys.argv[2]
CUDA = len(sys.argv) > 3 and sys.argv[3] == "cuda"
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 60

import gshell_amd.compat as compat  # noqa: E402
from gshell_amd import grid  # noqa: E402
if not CUDA:
    from oracle import refload  # noqa: E402

compat.install()
import render as shim_render  # noqa: E402  (= gshell_amd.render)


def load_ref_module(name, rel):
    """exec a reference python file as module `name` (its relative imports resolve inside the shimmed package)"""
    mod = types.ModuleType(name)
    mod.__file__ = os.path.join(ref_root, rel)
    mod.__package__ = name.rpartition(".")[0]
    sys.modules[name] = mod
    exec(compile(open(mod.__file__).read(), mod.__file__, "exec"), mod.__dict__)
    setattr(sys.modules[mod.__package__], name.rpartition(".")[2], mod)
    return mod


for n in ("texture", "material"):
    if not hasattr(shim_render, n):
        load_ref_module("render." + n, f"render/{n}.py")
sys.modules.setdefault("xatlas", types.ModuleType("xatlas"))


class _Dataset(torch.utils.data.Dataset):
    """synthetic stand-in with the batch layout of dataset/dataset_deepfashion.py (mv, mvp, campos, resolution, spp, img)"""
    _gt = None          # cuda mode: (mvp [K,4,4], campos [K,3], img [K,H,W,4]) of the ground-truth state

    def __init__(self, root, FLAGS, examples=None):
        self.FLAGS = FLAGS
        if CUDA:
            self.n = ITERS * FLAGS.batch + FLAGS.batch if examples is not None else 8          # the train loader runs one epoch: FLAGS.iter + 1 batches
            if _Dataset._gt is None:
                _Dataset._gt = ground_truth_views(FLAGS)
        else:
            self.n = 4 if examples is None else min(int(examples), 4)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if CUDA:
            mvp, cam, img = _Dataset._gt
            k = i % mvp.shape[0]
            return {'mv': torch.eye(4)[None], 'mvp': mvp[k:k + 1], 'campos': cam[k:k + 1], 'resolution': self.FLAGS.train_res, 'spp': self.FLAGS.spp, 'img': img[k:k + 1]}
        from oracle import scenes
        mvp, cam = scenes.orbit_views(1, first=i)
        H, W = self.FLAGS.train_res
        g = torch.Generator().manual_seed(i)
        return {'mv': torch.eye(4)[None], 'mvp': torch.tensor(mvp), 'campos': torch.tensor(cam), 'resolution': self.FLAGS.train_res, 'spp': self.FLAGS.spp,
                'img': torch.rand(1, H, W, 4, generator=g)}

    def collate(self, batch):
        return {'mv': torch.cat([b['mv'] for b in batch]), 'mvp': torch.cat([b['mvp'] for b in batch]), 'campos': torch.cat([b['campos'] for b in batch]),
                'resolution': batch[0]['resolution'], 'spp': batch[0]['spp'], 'img': torch.cat([b['img'] for b in batch])}


def ground_truth_views(FLAGS, K=24):
    """K views of the benchmark's "mid-training" garment state on the same grid, rendered by this package (host tensors, as a dataset hands them over)"""
    from gshell_amd import workload
    H, W = FLAGS.train_res
    tr = workload.build(res=FLAGS.gshell_grid, n_samples=4, batch=1, train_res=(H, W), fit_steps=400)
    imgs, mvps, cams = [], [], []
    for k in range(K):
        t = workload.make_targets(tr, [k * 3], (H, W))
        imgs.append(t['img'].cpu())
        mvps.append(t['mvp'].cpu())
        cams.append(t['campos'].cpu())
    del tr
    torch.cuda.empty_cache()
    return torch.cat(mvps), torch.cat(cams), torch.cat(imgs)


ds = types.ModuleType("dataset")
for sub, cls in (("dataset_deepfashion", "DatasetDeepFashion"), ("dataset_deepfashion_testset", "DatasetDeepFashionTestset")):
    m = types.ModuleType("dataset." + sub)
    setattr(m, cls, _Dataset)
    sys.modules["dataset." + sub] = m
    setattr(ds, sub, m)
sys.modules["dataset"] = ds

# a tet grid where the script looks for it (data/tets/{res}_tets.npz, keys of data/tets/generate_tets.py:47)
os.makedirs(os.path.join(work, "data", "tets"), exist_ok=True)
if CUDA:
    # out-of-scope IO the script calls after training (render/util.py image IO needs imageio; render/light.py:save_env_map): minimal writers
    import gshell_amd.render.util as _util
    import gshell_amd.render.light as _light
    if not hasattr(_util, "save_image"):
        def _save_image(fn, x):
            from PIL import Image
            Image.fromarray(np.clip(np.rint(np.asarray(x) * 255.0), 0, 255).astype(np.uint8)).save(fn)
        _util.save_image = _save_image
    if not hasattr(_light, "save_env_map"):
        _light.save_env_map = lambda fn, lgt: np.save(fn + ".npy", lgt.base.detach().cpu().numpy())
    verts, tets = grid.grid_for_res(64)
    np.savez(os.path.join(work, "data", "tets", "64_tets.npz"), vertices=verts.numpy(), indices=tets.numpy())
    cfg = {"gshell_grid": 64, "sdf_mlp_pretrain_steps": 300, "train_res": [256, 256], "batch": 2, "n_samples": 4, "iter": ITERS, "out_dir": os.path.join(work, "out"),
           "trainset_path": work, "index": 0, "validate": True, "save_interval": 0, "display_interval": 0,
           "boxscale": [1, 1, 1], "aabb": [-1, -1, -1, 1, 1, 1], "learning_rate": [0.03, 0.005], "background": "white", "denoiser": "bilateral"}   # as configs/deepfashion_mc_256.json
else:
    verts, tets = grid.bcc_grid(4)
    np.savez(os.path.join(work, "data", "tets", "8_tets.npz"), vertices=verts.numpy(), indices=tets.numpy())
    cfg = {"gshell_grid": 8, "sdf_mlp_pretrain_steps": 2, "train_res": [32, 32], "batch": 2, "n_samples": 1, "iter": 2, "out_dir": os.path.join(work, "out"),
           "trainset_path": work, "index": 0, "validate": False, "save_interval": 0,
           "boxscale": [1, 1, 1], "aabb": [-1, -1, -1, 1, 1, 1], "learning_rate": [0.03, 0.005], "background": "white", "denoiser": "bilateral"}   # as configs/deepfashion_mc_256.json
json.dump(cfg, open(os.path.join(work, "cfg.json"), "w"))
os.chdir(work)
sys.argv = ["train_gshelltet_deepfashion.py", "--config", os.path.join(work, "cfg.json"), "--trainset_path", work, "--index", "0", "-o", cfg["out_dir"]]
if CUDA:
    runpy.run_path(os.path.join(ref_root, "train_gshelltet_deepfashion.py"), run_name="__main__")
    print(open(os.path.join(cfg["out_dir"], "30", "validate", "metrics.txt")).read())
else:
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(synchronize=lambda: None, cuda_stream=0)
    with refload.CudaToCpu():
        runpy.run_path(os.path.join(ref_root, "train_gshelltet_deepfashion.py"), run_name="__main__")
