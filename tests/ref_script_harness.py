"""Runs the REFERENCE'S OWN training script (train_gshelltet_deepfashion.py, from /root/reference, unmodified, as __main__) against
gshell_amd.compat on a box WITHOUT a GPU (TEST INFRASTRUCTURE; used by tests/test_compat_cpu.py in a subprocess).

What is stubbed: the out-of-scope imports (xatlas, dataset.*: SURVEY.md section 2) -- a synthetic dataset with the reference's batch
keys and collate (dataset/dataset_deepfashion.py:59-138) stands in --, and `device='cuda'` / `.cuda()` are redirected to the CPU so
that the script's host-side set-up runs.  `render.material`, `render.texture` and `render.obj` are the reference's own python files,
loaded into the shimmed `render` package.  The script then runs: argument parsing, FLAGS, rasteriser context, datasets, trainable
env light, denoiser, GShellTetsGeometry(grid, scale, FLAGS) incl. the SDF-network pre-fit, initial_guess_material, optimize_mesh:
three Adam optimisers over the parameter groups picked BY NAME, the DataLoader, prepare_batch, zero_grad, lgt.update_pdf(),
geometry.tick(...) -- where the first HIP entry point refuses the CPU tensors (there is no CPU fallback) and raises GShellHipError.
usage: python tests/ref_script_harness.py <reference root> <work dir>"""
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

import gshell_amd.compat as compat  # noqa: E402
from gshell_amd import grid  # noqa: E402
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

    def __init__(self, root, FLAGS, examples=None):
        self.FLAGS, self.n = FLAGS, 4 if examples is None else min(int(examples), 4)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from oracle import scenes
        mvp, cam = scenes.orbit_views(1, first=i)
        H, W = self.FLAGS.train_res
        g = torch.Generator().manual_seed(i)
        return {'mv': torch.eye(4)[None], 'mvp': torch.tensor(mvp), 'campos': torch.tensor(cam), 'resolution': self.FLAGS.train_res, 'spp': self.FLAGS.spp,
                'img': torch.rand(1, H, W, 4, generator=g)}

    def collate(self, batch):
        return {'mv': torch.cat([b['mv'] for b in batch]), 'mvp': torch.cat([b['mvp'] for b in batch]), 'campos': torch.cat([b['campos'] for b in batch]),
                'resolution': batch[0]['resolution'], 'spp': batch[0]['spp'], 'img': torch.cat([b['img'] for b in batch])}


ds = types.ModuleType("dataset")
for sub, cls in (("dataset_deepfashion", "DatasetDeepFashion"), ("dataset_deepfashion_testset", "DatasetDeepFashionTestset")):
    m = types.ModuleType("dataset." + sub)
    setattr(m, cls, _Dataset)
    sys.modules["dataset." + sub] = m
    setattr(ds, sub, m)
sys.modules["dataset"] = ds

# a tiny tet grid where the script looks for it (data/tets/{res}_tets.npz, keys of data/tets/generate_tets.py:47)
os.makedirs(os.path.join(work, "data", "tets"), exist_ok=True)
verts, tets = grid.bcc_grid(4)
np.savez(os.path.join(work, "data", "tets", "8_tets.npz"), vertices=verts.numpy(), indices=tets.numpy())
cfg = {"gshell_grid": 8, "sdf_mlp_pretrain_steps": 2, "train_res": [32, 32], "batch": 2, "n_samples": 1, "iter": 2, "out_dir": os.path.join(work, "out"),
       "trainset_path": work, "index": 0, "validate": False, "save_interval": 0,
       "boxscale": [1, 1, 1], "aabb": [-1, -1, -1, 1, 1, 1], "learning_rate": [0.03, 0.005], "background": "white", "denoiser": "bilateral"}   # as configs/deepfashion_mc_256.json
json.dump(cfg, open(os.path.join(work, "cfg.json"), "w"))
os.chdir(work)
sys.argv = ["train_gshelltet_deepfashion.py", "--config", os.path.join(work, "cfg.json"), "--trainset_path", work, "--index", "0", "-o", cfg["out_dir"]]
torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(synchronize=lambda: None, cuda_stream=0)
with refload.CudaToCpu():
    runpy.run_path(os.path.join(ref_root, "train_gshelltet_deepfashion.py"), run_name="__main__")
