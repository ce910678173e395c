"""Trainable lat-long environment probe + the sampling tables the Monte-Carlo shader reads
(semantics of the reference's render/light.py:21-59 and :102-105).

`update_pdf` must run before every iteration (train_gshelltet_deepfashion.py:412): pdf = max_c(base) * sin(theta),
normalised; `cols` = per-row CDF over columns; `rows` = CDF over the row marginals (the shader gets rows[:,0])."""
import math

import torch


class EnvironmentLight:
    LIGHT_MIN_RES = 16
    MIN_ROUGHNESS = 0.08
    MAX_ROUGHNESS = 0.5

    def __init__(self, base):
        self.mtx = None
        self.base = base
        self.pdf_scale = (base.shape[0] * base.shape[1]) / (2 * math.pi * math.pi)
        self._sin_theta = None
        self.update_pdf()

    def xfm(self, mtx):
        self.mtx = mtx

    def parameters(self):
        return [self.base]

    def clone(self):
        return EnvironmentLight(self.base.clone().detach())

    def clamp_(self, min=None, max=None):
        self.base.clamp_(min, max)

    @torch.no_grad()
    def update_pdf(self):
        Hl, Wl = self.base.shape[0], self.base.shape[1]
        if self.base.is_cuda and Hl <= 1024:
            # two launches (csrc/pixelops.hip: k_light_rows / k_light_norm) instead of 15 ATen launches before every iteration
            from .. import _lib
            base = self.base.detach().contiguous().float()
            self._pdf = torch.empty((Hl, Wl), dtype=torch.float32, device=base.device)
            self.rows, self.cols = torch.empty_like(self._pdf), torch.empty_like(self._pdf)
            mass = torch.empty(Hl, dtype=torch.float32, device=base.device)
            with torch.cuda.device(base.device):
                _lib.check(_lib.lib().gs_light_tables(_lib.ptr(base, torch.float32, "base"), _lib.c_int64(Hl), _lib.c_int64(Wl), _lib.ptr(self._pdf),
                                                      _lib.ptr(self.rows), _lib.ptr(self.cols), _lib.ptr(mass), _lib.stream()), "gs_light_tables")
            return
        # CPU tensors (host-side tests, the reference script's set-up on a GPU-less box): the same tables as torch ops
        if self._sin_theta is None or self._sin_theta.device != self.base.device:
            v = (torch.arange(Hl, dtype=torch.float32, device=self.base.device) + 0.5) / Hl      # texel-centre latitude
            self._sin_theta = torch.sin(v * math.pi)[:, None]
        w = self.base.amax(dim=-1) * self._sin_theta
        self._pdf = w / w.sum()
        cols = self._pdf.cumsum(dim=1)
        row_mass = cols[:, -1:]
        rows = row_mass.expand(Hl, Wl).cumsum(dim=0)
        self.cols = cols / torch.where(row_mass > 0, row_mass, torch.ones_like(row_mass))
        total = rows[-1:, :]
        self.rows = rows / torch.where(total > 0, total, torch.ones_like(total))


def create_trainable_env_rnd(base_res, scale=0.5, bias=0.25, device='cuda'):
    base = torch.rand(base_res, base_res, 3, dtype=torch.float32, device=device) * scale + bias
    return EnvironmentLight(base.requires_grad_(True))
