"""Small tensor helpers used by the render path (semantics of the reference's render/util.py:19-31, :61-65,
:97-101, :195-212, :242-248; written for device-agnostic use -- the reference hard-codes device='cuda')."""
import math

import torch
import torch.nn.functional as F


def dot(a, b):
    return (a * b).sum(dim=-1, keepdim=True)


def length(v, eps=1e-20):
    # clamp under the root: d sqrt(0) would be NaN in the backward pass
    return dot(v, v).clamp_min(eps).sqrt()


def safe_normalize(v, eps=1e-20):
    return v / length(v, eps)


def reflect(v, n):
    return 2 * dot(v, n) * n - v


def to_hvec(v, w):
    return F.pad(v, (0, 1), value=w)


def rgb_to_srgb(img):
    """Linear -> sRGB on the first three channels; a 4th (alpha) channel passes through."""
    assert img.shape[-1] in (3, 4)
    rgb = img[..., :3]
    srgb = torch.where(rgb <= 0.0031308, rgb * 12.92, rgb.clamp_min(0.0031308).pow(1.0 / 2.4) * 1.055 - 0.055)
    return srgb if img.shape[-1] == 3 else torch.cat((srgb, img[..., 3:4]), dim=-1)


_PIXEL_GRIDS = {}


def pixel_grid(width, height, center_x=0.5, center_y=0.5, device="cuda"):
    """[height, width, 2] of (x, y) pixel-centre coordinates in [0,1].  A constant of the frame size: built once per (size, device) and
    returned READ-ONLY (seven launches per render otherwise); callers that change it clone it."""
    key = (int(width), int(height), float(center_x), float(center_y), str(device))
    g = _PIXEL_GRIDS.get(key)
    if g is None:
        xs = (torch.arange(width, dtype=torch.float32, device=device) + center_x) / width
        ys = (torch.arange(height, dtype=torch.float32, device=device) + center_y) / height
        g = _PIXEL_GRIDS[key] = torch.stack((xs[None, :].expand(height, width), ys[:, None].expand(height, width)), dim=-1)
    return g


def scale_img_nhwc(x, size, mag='bilinear', min='area'):
    h, w = x.shape[1:3]
    shrink, grow = (h >= size[0] and w >= size[1]), (h < size[0] and w < size[1])
    assert shrink or grow, "cannot magnify one axis and minify the other"
    y = x.permute(0, 3, 1, 2)
    if h > size[0] and w > size[1]:
        y = F.interpolate(y, size, mode=min)
    elif mag in ('bilinear', 'bicubic'):
        y = F.interpolate(y, size, mode=mag, align_corners=True)
    else:
        y = F.interpolate(y, size, mode=mag)
    return y.permute(0, 2, 3, 1).contiguous()


def avg_pool_nhwc(x, size):
    return F.avg_pool2d(x.permute(0, 3, 1, 2), size).permute(0, 2, 3, 1).contiguous()


def perspective(fovy=0.7854, aspect=1.0, n=0.1, f=1000.0, device=None):
    """gluPerspective with the reference's flipped y row (render/util.py:242-248)."""
    t = math.tan(fovy / 2)
    m = torch.zeros(4, 4, dtype=torch.float32)
    m[0, 0], m[1, 1] = 1 / (t * aspect), -1 / t
    m[2, 2], m[2, 3], m[3, 2] = -(f + n) / (f - n), -(2 * f * n) / (f - n), -1.0
    return m.to(device) if device is not None else m


def mse_to_psnr(mse):
    return -10.0 * math.log10(mse)


def time_to_text(seconds):
    """'1.50 h' / '2.00 m' / '3.00 s' (reference render/util.py:511-517; the training loops print the remaining time with it)"""
    for unit, span in (("h", 3600.0), ("m", 60.0)):
        if seconds > span:
            return "%.2f %s" % (seconds / span, unit)
    return "%.2f s" % seconds
