"""Cross-bilateral denoiser applied to the demodulated diffuse / specular radiance (reference
denoiser/denoiser.py:21-35).  Input layout [..., 8] = (rgb, normal, depth, |d depth|); sigma ramps with the
shadow influence, 0 -> 2 px over the first 1000 iterations (geometry/gshell_tets_geometry.py:264-265)."""
import math

import torch

from ..render import optixutils as ou
from ..render import util


class BilateralDenoiser(torch.nn.Module):
    def __init__(self, influence=1.0):
        super().__init__()
        self.set_influence(influence)

    def set_influence(self, factor):
        self.sigma = max(2.0 * factor, 1e-4)
        self.variance = self.sigma * self.sigma
        self.N = 2 * math.ceil(2.5 * self.sigma) + 1        # filter radius used by the kernel

    def filter_raw(self, rgb, nrm_unit, zdz, mask=None):
        """(sum_t w c_t, max(sum_t w, 1e-4)) [...,4] of the filter -- the division is done by the caller (render.shade fuses it
        into the buffer assembly).  `nrm_unit` must already be normalised.  `mask` > 0 marks the pixels whose value the caller uses."""
        return ou.bilateral_denoiser_raw(rgb, nrm_unit, zdz, self.sigma, mask)

    def filter_raw_pair(self, rgb_a, rgb_b, nrm_unit, zdz, mask=None):
        """filter_raw of two images with the same guides (diffuse and specular radiance): the weights are computed once"""
        return ou.bilateral_denoiser_raw_pair(rgb_a, rgb_b, nrm_unit, zdz, self.sigma, mask)

    def forward(self, input):
        rgb, nrm, zdz = input[..., :3], input[..., 3:6], input[..., 6:8]
        return ou.bilateral_denoiser(rgb, util.safe_normalize(nrm), zdz, self.sigma)   # bent normals are not unit length
