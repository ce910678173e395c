"""One optimisation iteration of G-Shell reconstruction, as the reference's `optimize_mesh` body does it
(train_gshelltet_deepfashion.py:278-497: zero_grad x3 -> lgt.update_pdf -> geometry.tick -> backward -> gradient
rescaling -> 3 Adam steps + LambdaLR -> clamps), plus what the reference lacks: view-sharded data parallelism with ONE
RCCL all-reduce of the flattened gradient per iteration.

    flags   = default_flags(gshell_grid=..., n_samples=..., batch=..., train_res=[H, W])
    trainer = Trainer(flags, tet_grid=(verts, indices))
    losses  = trainer.step(target)          # target: dict(mvp [B,4,4], campos [B,3], img [B,H,W,4], background [B,H,W,3], resolution, spp)
"""
import os
import types

import numpy as np
import torch
import torch.distributed as dist

from .denoiser.denoiser import BilateralDenoiser
from .optim import HipAdam
from .geometry.gshell_tets_geometry import GShellTetsGeometry
from .render import light, mlptexture, render
from .render import rast as dr
from .render import renderutils as ru


def default_flags(**overrides):
    """FLAGS of train_gshelltet_deepfashion.py:504-594 (argparse defaults + hard-coded assignments)."""
    F = types.SimpleNamespace(
        iter=5000, batch=1, spp=1, layers=1, train_res=[512, 512], learning_rate=0.01, min_roughness=0.08, loss='logl1', background='checker',
        n_samples=4, bsdf='pbr', denoiser='bilateral', denoiser_demodulate=True, msdf_reg_open_scale=1e-6, msdf_reg_close_scale=3e-6,
        eikonal_scale=None, sdf_regularizer=0.2,
        gshell_grid=64, mesh_scale=1.4, probe_res=256, learn_lighting=True, no_perturbed_nrm=False, decorrelated=False,
        kd_min=[0.0, 0.0, 0.0, 0.0], kd_max=[1.0, 1.0, 1.0, 1.0], ks_min=[0.0, 0.001, 0.0], ks_max=[0.0, 1.0, 1.0], clip_max_norm=0.0,
        lambda_kd=0.1, lambda_ks=0.05, lambda_nrm=0.025, lambda_chroma=0.0, lambda_diffuse=0.15, lambda_specular=0.0025,
        use_tanh_deform=False, use_sdf_mlp=True, use_msdf_mlp=False, use_eikonal=True, sdf_mlp_pretrain_steps=1000, use_mesh_msdf_reg=True,
        sphere_init=False, sphere_init_norm=0.5, n_hidden=6, d_hidden=256, n_freq=6, skip_in=[3], use_float16=False, visualize_watertight=False,
        boxscale=[1, 1, 1], use_depth=False, use_img_2nd_layer=False, view_shard=None, shard_mlp_rows=True, seed=0, eikonal_side_stream=False, fused_assemble=True, sync_free_rows=False)
    if os.environ.get("GSHELL_EIKONAL_SIDE_STREAM") in ("0", "1"):          # diagnostics: A / B of the side stream without touching the callers
        F.eikonal_side_stream = os.environ["GSHELL_EIKONAL_SIDE_STREAM"] == "1"
    for k, v in overrides.items():
        setattr(F, k, v)
    return F


def create_loss(FLAGS):
    table = {"smape": ('smape', 'none'), "mse": ('mse', 'none'), "logl1": ('l1', 'log_srgb'), "logl2": ('mse', 'log_srgb'), "relmse": ('relmse', 'none')}
    l, tm = table[FLAGS.loss]
    return ImageLoss(l, tm)


class ImageLoss:
    """The reference's createLoss closure (train script: ru.image_loss with a fixed loss / tonemapper) as an object: `gs_spec`
    lets GShellTetsGeometry.tick fold the colour term into its one pass over the frame (regularizer.frame_sums)."""

    def __init__(self, loss, tonemapper):
        self.loss, self.tonemapper = loss, tonemapper
        self.gs_spec = (ru._LOSS[loss], ru._TONEMAP[tonemapper])

    def __call__(self, img, ref):
        return ru.image_loss(img, ref, loss=self.loss, tonemapper=self.tonemapper)


class ViewShard:
    """Views of the global batch are dealt round-robin to the ranks of one node; geometry is replicated
    (deterministic integer topology => identical meshes on every rank)."""

    def __init__(self, rank=0, world=1, group=None):
        self.rank, self.world, self.group = rank, world, group

    def local_views(self, B):
        return list(range(self.rank, B, self.world))

    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def _backend(self):
        return dist.get_backend(self.group) if self.world > 1 else None

    def all_gather_rows(self, local):
        """[per] on every rank -> [per * world] (rank-major).  One RCCL all-gather; list form on gloo (CPU tests)."""
        if self.world == 1:
            return local
        out = torch.empty(local.numel() * self.world, dtype=local.dtype, device=local.device)
        if self._backend() == "nccl":
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        else:
            dist.all_gather(list(out.chunk(self.world)), local.contiguous(), group=self.group)
        return out

    def reduce_scatter_sum(self, full):
        """[per * world] on every rank -> sum over ranks of this rank's [per] slice."""
        if self.world == 1:
            return full
        per = full.numel() // self.world
        if self._backend() == "nccl":
            out = torch.empty(per, dtype=full.dtype, device=full.device)
            dist.reduce_scatter_tensor(out, full.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
            return out
        full = full.clone()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        return full[self.rank * per:(self.rank + 1) * per]

    def broadcast(self, tensors, src=0):
        """Make replicated state bit-identical on every rank (parameters after a pre-fit whose GPU reductions are not
        order-deterministic): geometry replication relies on identical SDF signs everywhere."""
        if self.world > 1:
            for t in tensors:
                dist.broadcast(t.data if hasattr(t, "data") else t, src=src, group=self.group)
                torch.autograd.graph.increment_version(t)          # written through .data: caches keyed on the version (packed SDF weights) must see it


def flat_all_reduce_grads(params, shard, buf=None):
    """Sum the gradients of `params` over the ranks of `shard` with ONE all-reduce of a flat fp32 bucket
    (~(3N + N + |hash grid| + |MLPs| + |probe|) floats, ~90 MB at tet-res 256: far below an iteration even on a single
    xGMI link, so no bucketing / overlap).  Parameters without a gradient on this rank contribute zeros.  Returns the
    bucket for reuse."""
    n = sum(p.numel() for p in params)
    if n == 0:
        return buf
    if buf is None or buf.numel() != n or buf.device != params[0].device:
        buf = torch.empty(n, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        k = p.numel()
        if p.grad is None:
            buf[off:off + k].zero_()
        else:
            buf[off:off + k].copy_(p.grad.reshape(-1))
        off += k
    shard.all_reduce_sum(buf)
    off = 0
    for p in params:
        k = p.numel()
        if p.grad is None:
            p.grad = buf[off:off + k].reshape(p.shape).clone()
        else:
            p.grad.copy_(buf[off:off + k].reshape(p.shape))
        off += k
    return buf


def sharded_total_loss(per_view, global_terms, B_local, B_global, world, presharded=None):
    """Loss a rank back-propagates so that the SUM of all ranks' gradients equals the single-GPU gradient of
    mean-over-global-batch(per-view terms) + view-independent terms.  `presharded` terms are already
    divided over the ranks by construction (each rank owns a disjoint part of a sum) and enter with weight 1."""
    total = per_view * (B_local / B_global) + global_terms / world
    return total if presharded is None else total + presharded


def initial_guess_material(geometry, FLAGS):
    dev = geometry.verts.device
    kd_min, kd_max = torch.tensor(FLAGS.kd_min, dtype=torch.float32, device=dev), torch.tensor(FLAGS.kd_max, dtype=torch.float32, device=dev)
    ks_min, ks_max = torch.tensor(FLAGS.ks_min, dtype=torch.float32, device=dev), torch.tensor(FLAGS.ks_max, dtype=torch.float32, device=dev)
    tex = mlptexture.MLPTexture3D(geometry.getAABB(), channels=6, min_max=[torch.cat((kd_min[0:3], ks_min)), torch.cat((kd_max[0:3], ks_max))])
    return {'kd_ks': tex, 'bsdf': FLAGS.bsdf, 'no_perturbed_nrm': FLAGS.no_perturbed_nrm}


class Trainer:
    def __init__(self, FLAGS, tet_grid=None, shard=None, geometry=None):
        self.FLAGS = FLAGS
        self.shard = shard or ViewShard()
        FLAGS.view_shard = self.shard
        # render noise + eikonal samples as functions of (seed, iteration, global view) -- 1-GPU step == N-GPU step
        FLAGS.noise_stream = render.NoiseStream(getattr(FLAGS, 'seed', 0), self.shard.rank, self.shard.world)
        self.glctx = dr.RasterizeGLContext()
        self.lgt = light.create_trainable_env_rnd(FLAGS.probe_res, scale=0.0, bias=0.5)      # train script :656
        self.denoiser = BilateralDenoiser().cuda() if FLAGS.denoiser == 'bilateral' else None
        self.geometry = geometry or GShellTetsGeometry(FLAGS.gshell_grid, FLAGS.mesh_scale, FLAGS, tet_grid=tet_grid)
        self.mat = initial_guess_material(self.geometry, FLAGS)
        self.loss_fn = create_loss(FLAGS)
        lr = FLAGS.learning_rate
        lr_pos = lr[0] if isinstance(lr, (list, tuple)) else lr
        lr_mat = lr[1] if isinstance(lr, (list, tuple)) else lr
        lr_lgt = lr[2] if isinstance(lr, (list, tuple)) and len(lr) > 2 else (lr_pos * 6.0 if not isinstance(lr, (list, tuple)) else lr[0] * 6.0)
        named = list(self.geometry.named_parameters())
        groups = [
            {'params': [p for n, p in named if 'deform' in n], 'lr': lr_pos},
            {'params': [p for n, p in named if 'msdf' in n], 'lr': lr_pos},
            {'params': [p for n, p in named if 'sdf' in n and 'msdf' not in n], 'lr': lr_pos * 1e-2},
            {'params': [p for n, p in named if 'sdf' not in n and 'deform' not in n], 'lr': lr_pos * 1e-2},     # FlexiCubes per-cube weights
        ]
        groups = [g for g in groups if len(g['params'])]
        # torch.optim.Adam (reference train script :372-383) with the step as one HIP launch per optimiser (gshell_amd/optim.py: same
        # arithmetic and state as torch's fused Adam, which needs 5 multi-tensor launches + 5 step-counter launches at 1.6 TB/s)
        Adam = HipAdam if getattr(FLAGS, "hip_adam", True) else (lambda *a, **k: torch.optim.Adam(*a, fused=True, **k))
        self.opt_mesh = Adam(groups, eps=1e-8) if FLAGS.use_sdf_mlp else Adam(self.geometry.parameters(), lr=lr_pos)
        self.mat_params = list(self.mat['kd_ks'].parameters())
        self.opt_mat = Adam(self.mat_params, lr=lr_mat)
        self.opt_light = Adam(self.lgt.parameters(), lr=lr_lgt)
        sched = lambda it: max(0.0, 10 ** (-it * 0.0002))
        self.scheds = [torch.optim.lr_scheduler.LambdaLR(o, lr_lambda=sched) for o in (self.opt_mat, self.opt_mesh, self.opt_light)]
        self.it = 0
        self._flat = None

    def sync_replicas(self):
        """Broadcast every trainable tensor from rank 0 (call once after construction / pre-fitting)."""
        self.shard.broadcast(self.all_params())
        self.lgt.update_pdf()

    def parallelism(self):
        w = self.shard.world
        if w == 1:
            return "single GPU"
        rows = ("SDF-MLP rows + eikonal samples sharded (each rank: ONE-pass fp16-pair forward over its N / G rows, all-gather sdf[N])"
                if getattr(self.FLAGS, "shard_mlp_rows", False) else "geometry replicated")
        return f"view-shard dp{w}, {rows}, one flat RCCL all-reduce of the gradient"

    def all_params(self):
        return [p for g in self.opt_mesh.param_groups for p in g['params']] + self.mat_params + list(self.lgt.parameters())

    def _all_reduce_grads(self):
        self._flat = flat_all_reduce_grads([p for p in self.all_params() if p.requires_grad], self.shard, self._flat)

    def forward_backward(self, target, global_batch=None):
        """zero_grad -> update_pdf -> tick -> backward (-> gradient all-reduce): leaves the GLOBAL-batch gradient in .grad
        on every rank.  `target` holds THIS rank's views; `global_batch` = views over all ranks (default: local x world)."""
        for o in (self.opt_mat, self.opt_mesh, self.opt_light):
            o.zero_grad()
        self.lgt.update_pdf()
        self.FLAGS.noise_stream.set_iteration(self.it, global_batch if self.shard.world > 1 else None)
        img_loss, depth_loss, reg_loss = self.geometry.tick(self.glctx, target, self.lgt, self.mat, self.loss_fn, self.it, denoiser=self.denoiser)
        if self.shard.world > 1:
            B_local = target['mvp'].shape[0]
            B = global_batch or B_local * self.shard.world
            t = self.geometry.last_terms
            total = sharded_total_loss(t['per_view'], t['global'], B_local, B, self.shard.world, t.get('presharded'))
        else:
            total = img_loss + reg_loss
        total.backward()
        if self.shard.world > 1:
            self._all_reduce_grads()
        return img_loss, reg_loss

    def step(self, target, global_batch=None):
        img_loss, reg_loss = self.forward_backward(target, global_batch)
        if self.lgt.base.grad is not None:
            self.lgt.base.grad *= 64                                   # train script :433
        enc = self.mat['kd_ks'].encoder.params
        if enc.grad is not None:
            enc.grad /= 8.0                                            # train script :435
        if self.FLAGS.clip_max_norm > 0.0:                            # train script :440-444 (after the all-reduce: global gradient)
            # ONE joint norm over geometry + material parameters, as the reference's clip_grad_norm_(geometry.parameters() + params)
            torch.nn.utils.clip_grad_norm_(list(self.geometry.parameters()) + list(self.mat_params), self.FLAGS.clip_max_norm)
        self.opt_mat.step(); self.scheds[0].step()
        self.opt_mesh.step(); self.scheds[1].step()
        self.opt_light.step(); self.scheds[2].step()
        with torch.no_grad():
            self.lgt.clamp_(min=1e-4)
            self.geometry.clamp_deform()
        self.it += 1
        return img_loss.detach(), reg_loss.detach()

    # ---- checkpoint / resume (the reference saves geometry / material / light state_dicts, train script :700-720) -------
    def state_dict(self):
        return {
            'it': self.it,
            'geometry': self.geometry.state_dict(),        # keys: sdf, msdf, deform, sdf_net.* (same names as the reference module)
            'material': self.mat['kd_ks'].state_dict(),
            'light': self.lgt.base.detach().clone(),       # [H,W,3] probe texels (the reference stores an .hdr, light.py:118-123)
            'opt': [o.state_dict() for o in (self.opt_mat, self.opt_mesh, self.opt_light)],
            'sched': [s.state_dict() for s in self.scheds],
            # sampler / noise state: without it a resumed run restarts the shadow-ray seed and the noise streams at 0
            'rnd_seed': render.rnd_seed, 'noise_stream': self.FLAGS.noise_stream.state_dict(),
            'rng': {'torch': torch.get_rng_state(), 'cuda': torch.cuda.get_rng_state(self.geometry.verts.device), 'numpy': np.random.get_state()},
        }

    def load_state_dict(self, sd):
        self.geometry.load_state_dict(sd['geometry'])
        self.mat['kd_ks'].load_state_dict(sd['material'])
        self.lgt.base.data.copy_(sd['light'])
        for o, s in zip((self.opt_mat, self.opt_mesh, self.opt_light), sd['opt']):
            o.load_state_dict(s)
        for o, s in zip(self.scheds, sd['sched']):
            o.load_state_dict(s)
        self.it = int(sd['it'])
        if 'rnd_seed' in sd:
            render.rnd_seed = int(sd['rnd_seed'])
            self.FLAGS.noise_stream.load_state_dict(sd['noise_stream'])
            torch.set_rng_state(sd['rng']['torch'].cpu())
            torch.cuda.set_rng_state(sd['rng']['cuda'].cpu(), self.geometry.verts.device)
            np.random.set_state(sd['rng']['numpy'])
        self.lgt.update_pdf()

    def save_checkpoint(self, path):
        if self.shard.rank == 0:
            torch.save(self.state_dict(), path)

    def load_checkpoint(self, path):
        self.load_state_dict(torch.load(path, map_location=self.geometry.verts.device, weights_only=False))


@torch.no_grad()
def validate(trainer, targets, out_dir=None):
    """Validation loop of the reference (train script :227-272): render each target view with the current state,
    MSE / PSNR of the clamped sRGB image against the target, `metrics.txt` with `ID, MSE, PSNR` rows and the averages.
    `targets` is an iterable of single-view target dicts (mvp, campos, resolution, spp, background, img)."""
    from .render import util
    rows = []
    for it, target in enumerate(targets):
        buf = trainer.geometry.render(trainer.glctx, target, trainer.lgt, trainer.mat, denoiser=trainer.denoiser)['buffers']
        opt = util.rgb_to_srgb(buf['shaded'][..., 0:3]).clamp(0.0, 1.0)
        ref = util.rgb_to_srgb(target['img'][..., 0:3]).clamp(0.0, 1.0)
        mse = float(torch.nn.functional.mse_loss(opt, ref))
        rows.append((it, mse, util.mse_to_psnr(max(mse, 1e-12))))
    avg_mse = sum(r[1] for r in rows) / max(len(rows), 1)
    avg_psnr = sum(r[2] for r in rows) / max(len(rows), 1)
    if out_dir is not None and trainer.shard.rank == 0:
        import os
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'metrics.txt'), 'w') as f:
            f.write('ID, MSE, PSNR\n')
            for r in rows:
                f.write("%d, %1.8f, %1.8f\n" % r)
            f.write("AVERAGES: %1.4f, %2.3f\n" % (avg_mse, avg_psnr))
    return avg_psnr
