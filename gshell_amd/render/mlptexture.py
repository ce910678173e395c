"""MLPTexture3D with the reference's surface (render/mlptexture.py:18-106): multiresolution hash-grid
encoding (HIP, gshell_amd/csrc/hashgrid.hip) + a 32-wide bias-free ReLU MLP + sigmoid range mapping."""
import numpy as np
import torch

from .. import _lib
from .._lib import c_float, c_int, c_int64, check, ptr, stream


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, mask, cfg):
        L = _lib.lib()
        x_c = x.detach().contiguous().float()
        p_c = params.detach().contiguous().float()
        m_c = None if mask is None else mask.detach().reshape(-1).contiguous().float()
        N = x_c.shape[0]
        out = torch.empty((N, cfg[0] * cfg[1]), dtype=torch.float32, device=x_c.device)
        with torch.cuda.device(x_c.device):
            check(L.gs_hashgrid_fwd(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]), ptr(x_c, torch.float32, "x"),
                                    ptr(m_c), c_int64(N), ptr(p_c, torch.float32, "params"), ptr(out), stream()), "gs_hashgrid_fwd")
        ctx.save_for_backward(x_c, p_c, m_c)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, g_out):
        x_c, p_c, m_c = ctx.saved_tensors
        cfg = ctx.cfg
        N = x_c.shape[0]
        g = g_out.contiguous().float()
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_params = torch.zeros_like(p_c) if need_p else None
        g_xl = torch.empty((cfg[0], N, 3), dtype=torch.float32, device=g.device) if need_x else None
        with torch.cuda.device(g.device):
            check(_lib.lib().gs_hashgrid_bwd(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]), ptr(x_c), ptr(m_c), c_int64(N),
                                             ptr(p_c), ptr(g), ptr(g_params), ptr(g_xl), stream()), "gs_hashgrid_bwd")
        return (g_xl.sum(0) if need_x else None), g_params, None, None


class _ScaleGrad(torch.autograd.Function):
    """identity forward, gradient * s backward (the reference does this with module backward hooks,
    render/mlptexture.py:31, :74)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


class _TexMlpFn(torch.autograd.Function):
    """sigmoid(W3 relu(W2 relu(W1 x))) * (hi - lo) + lo over the rows with mask > 0, one fused HIP kernel each way
    (gshell_amd/csrc/texmlp.hip)."""

    @staticmethod
    def forward(ctx, x, mask, w1, w2, w3, lo, hi):
        x_c = x.detach().contiguous().float()
        m_c = None if mask is None else mask.detach().reshape(-1).contiguous().float()
        ws = [t.detach().contiguous().float() for t in (w1, w2, w3, lo, hi)]
        N, C = x_c.shape[0], ws[2].shape[0]
        out = torch.empty((N, C), dtype=torch.float32, device=x_c.device)
        with torch.cuda.device(x_c.device):
            check(_lib.lib().gs_texmlp_fwd(ptr(x_c, torch.float32, "x"), ptr(m_c), c_int64(N), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), c_int(C),
                                           ptr(ws[3]), ptr(ws[4]), ptr(out), stream()), "gs_texmlp_fwd")
        ctx.save_for_backward(x_c, m_c, *ws)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x_c, m_c, w1, w2, w3, lo, hi = ctx.saved_tensors
        g = g_out.contiguous().float()
        N, C = x_c.shape[0], w3.shape[0]
        g_x = torch.empty_like(x_c) if ctx.needs_input_grad[0] else None
        g_w1, g_w2, g_w3 = torch.zeros_like(w1), torch.zeros_like(w2), torch.zeros_like(w3)
        with torch.cuda.device(g.device):
            check(_lib.lib().gs_texmlp_bwd(ptr(x_c), ptr(m_c), c_int64(N), ptr(w1), ptr(w2), ptr(w3), c_int(C), ptr(lo), ptr(hi), ptr(g),
                                           ptr(g_x), ptr(g_w1), ptr(g_w2), ptr(g_w3), stream()), "gs_texmlp_bwd")
        return g_x, None, g_w1, g_w2, g_w3, None, None


BINNED_TABLE_GRAD = True   # _FieldFn backward: hashed levels' table gradient through per-bin record arrays (gs_hashgrid_encode_bwd_binned), no atomics
BIN_COVERAGE = 0.2         # without a hint: bin capacity for this fraction of the rows having mask > 0; fuller frames spill to the atomic path
ACTIVE_ROWS_HINT = [None]  # rows with mask > 0 of the coming call, when the caller knows them on the host (render.shade: the covered pixels x 2
                           # coordinate sets -- the shader's pixel list is synchronised anyway): the capacity then follows the frame
_bin_scratch = {}


def _bins(cfg, N, device):
    """(counters [+ spill word], records, capacity) of the binned table gradient, cached per device.  The capacity only grows (in steps of
    1.5x, so that alternating frame sizes do not reallocate): 1.25 x the uniform share of the hinted active rows."""
    nb = int(_lib.lib().gs_hashgrid_bin_count(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4])))
    if nb <= 0:
        return None
    per_level = max(1, (1 << cfg[2]) // int(_lib.lib().gs_hashgrid_bin_entries()))      # bins of a full-size (hashed) level
    # the hint belongs to ONE coming call: consumed here (a later backward that does not come from shade() -- another frame size, a bake, a test --
    # must not size its bins from a stale frame)
    hint, ACTIVE_ROWS_HINT[0] = ACTIVE_ROWS_HINT[0], None
    active = hint if hint is not None else BIN_COVERAGE * N
    want = int(1.25 * min(active, N) * 8 / per_level) + 1024
    key = (str(device), nb)
    cur = _bin_scratch.get(key)
    if cur is None or cur[2] < want:
        cap = want if cur is None else max(want, int(1.5 * cur[2]))
        spilled = 0 if cur is None else int(cur[0][nb])
        _bin_scratch.pop(key, None)
        cur = None                       # the old record array is released BEFORE the larger one is allocated (peak = new, not old + new)
        counters = torch.zeros(nb + 1, dtype=torch.int32, device=device)
        counters[nb] = spilled
        _bin_scratch[key] = (counters, torch.empty(nb * cap * 3, dtype=torch.int32, device=device), cap)
    return _bin_scratch[key]


def bin_spill_count():
    """records of the binned table gradient that did not fit their bin and took the atomic path, summed over the process (one host sync)"""
    return sum(int(v[0][-1]) for v in _bin_scratch.values())


def free_bin_scratch():
    """release the cached record arrays (trainer teardown; ~0.5 GB at 4 x 512^2)"""
    _bin_scratch.clear()


COMPACT_ROWS = True      # _FieldFn: texture MLP over a device-compacted list of the masked rows (False: masked waves over all rows)


class _FieldFn(torch.autograd.Function):
    """MLPTexture3D.sample as two kernels each way: AABB normalisation + clamp + hash-grid encoding
    (gs_hashgrid_encode_*), texture MLP + range mapping (gs_texmlp_*_level_major), with the [L, N, 2] LEVEL-MAJOR
    feature tensor between them (hashgrid.hip explains why) and the reference's two gradient-scaling hooks
    (mlptexture.py:31 x128 into the MLP input, :74 /128 out of the encoder) folded in: their product is exactly 1
    on the table gradient and 1/128 on the position gradient... applied as the reference applies them.
    pos [N,3] world positions, mask [N] or None, img = (H, W) when the rows are whole images (tiles the backward)."""

    @staticmethod
    def forward(ctx, pos, params, mask, w1, w2, w3, lo, hi, aabb, cfg, mlp_scale, enc_scale, img, parts=1):
        L = _lib.lib()
        f = lambda t: None if t is None else t.detach().contiguous().float()
        pos_c, p_c, m_c = f(pos), f(params), (None if mask is None else f(mask.reshape(-1)))
        ws = [f(t) for t in (w1, w2, w3, lo, hi)]
        ab = f(aabb)
        N, C = pos_c.shape[0], ws[2].shape[0]
        feat = torch.empty((cfg[0], N, cfg[1]), dtype=torch.float32, device=pos_c.device)
        rows = count = None
        with torch.cuda.device(pos_c.device):
            if m_c is not None and N > 0 and COMPACT_ROWS:
                # encoder and texture MLP walk a compact list of the rows with mask > 0 (compacted on the device, no host sync): image
                # rows in scan order put a 64-row chunk on every crossing of the silhouette, two thirds of its lanes idle
                rows = torch.empty(N, dtype=torch.int32, device=pos_c.device)
                count = torch.empty(2, dtype=torch.int64, device=pos_c.device)
                scratch = torch.empty((int(L.gs_compact_rows_scratch_bytes(c_int64(N))) + 7) // 8, dtype=torch.int64, device=pos_c.device)
                check(L.gs_compact_rows(ptr(m_c), c_int64(N), c_int64(N), ptr(scratch), ptr(rows), ptr(None), ptr(count), stream()), "gs_compact_rows")
                check(L.gs_hashgrid_encode_fwd_rows(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]),
                                                    ptr(pos_c, torch.float32, "pos"), ptr(ab), ptr(rows), ptr(count), c_int64(N), c_int64(N),
                                                    ptr(p_c, torch.float32, "params"), ptr(feat), stream()), "gs_hashgrid_encode_fwd_rows")
            else:
                check(L.gs_hashgrid_encode_fwd(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]),
                                               ptr(pos_c, torch.float32, "pos"), ptr(ab), ptr(m_c), c_int64(N), ptr(p_c, torch.float32, "params"),
                                               ptr(feat), stream()), "gs_hashgrid_encode_fwd")
            if rows is not None:
                out = (0.5 * (ws[4] - ws[3]) + ws[3]).expand(N, C).contiguous()         # rows outside the list: the all-zero feature row, same expression as the kernel
                check(L.gs_texmlp_fwd_rows(ptr(feat), ptr(rows), ptr(count), c_int64(N), c_int64(N), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), c_int(C),
                                           ptr(ws[3]), ptr(ws[4]), ptr(out), stream()), "gs_texmlp_fwd_rows")
            else:
                out = torch.empty((N, C), dtype=torch.float32, device=pos_c.device)
                check(L.gs_texmlp_fwd_level_major(ptr(feat), ptr(m_c), c_int64(N), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), c_int(C), ptr(ws[3]), ptr(ws[4]),
                                                  ptr(out), stream()), "gs_texmlp_fwd_level_major")
        ctx.rows = (rows, count)
        ctx.save_for_backward(pos_c, p_c, m_c, feat, ab, *ws)
        ctx.cfg, ctx.scales, ctx.img, ctx.parts = cfg, (float(mlp_scale), float(enc_scale)), img, parts
        if parts == 1:
            return out
        # the `parts` coordinate sets as separate outputs (row blocks of the one buffer): slicing a single output instead costs a zero fill,
        # a copy and an add of the whole [parts * n, C] gradient per part in backward
        n = N // parts
        return tuple(out[j * n:(j + 1) * n] for j in range(parts))

    @staticmethod
    def backward(ctx, *g_outs):
        pos_c, p_c, m_c, feat, ab, w1, w2, w3, lo, hi = ctx.saved_tensors
        cfg, (mlp_scale, enc_scale), img = ctx.cfg, ctx.scales, ctx.img
        L = _lib.lib()
        if ctx.parts == 1:
            g = g_outs[0].contiguous().float()
        else:
            n = pos_c.shape[0] // ctx.parts
            g = torch.cat([torch.zeros((n, w3.shape[0]), dtype=torch.float32, device=pos_c.device) if t is None else t.float() for t in g_outs], 0)
        N, C = pos_c.shape[0], w3.shape[0]
        need_pos, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_feat = torch.empty_like(feat) if (need_pos or need_p) else None
        g_w = torch.zeros(w1.numel() + w2.numel() + w3.numel(), dtype=torch.float32, device=g.device)
        g_w1, g_w2, g_w3 = g_w[:w1.numel()].view_as(w1), g_w[w1.numel():w1.numel() + w2.numel()].view_as(w2), g_w[w1.numel() + w2.numel():].view_as(w3)
        g_params = torch.zeros_like(p_c) if need_p else None
        g_pos = torch.empty_like(pos_c) if need_pos else None
        H, W = img if img is not None else (0, 0)
        with torch.cuda.device(g.device):
            rows, count = ctx.rows
            if rows is not None:
                check(L.gs_texmlp_bwd_rows(ptr(feat), ptr(rows), ptr(count), c_int64(N), c_int64(N), ptr(w1), ptr(w2), ptr(w3), c_int(C), ptr(lo), ptr(hi),
                                           ptr(g), ptr(g_feat), ptr(g_w1), ptr(g_w2), ptr(g_w3), stream()), "gs_texmlp_bwd_rows")
            else:
                check(L.gs_texmlp_bwd_level_major(ptr(feat), ptr(m_c), c_int64(N), ptr(w1), ptr(w2), ptr(w3), c_int(C), ptr(lo), ptr(hi), ptr(g),
                                                  ptr(g_feat), ptr(g_w1), ptr(g_w2), ptr(g_w3), stream()), "gs_texmlp_bwd_level_major")
            if g_feat is not None:
                # d/d feat of the MLP carries the x128 hook; the table gradient takes it as is, the position gradient takes x128 / 128
                bins = _bins(cfg, N, g.device) if (BINNED_TABLE_GRAD and need_p) else None
                if bins is not None:
                    check(L.gs_hashgrid_encode_bwd_binned(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]), ptr(pos_c), ptr(ab),
                                                          ptr(m_c), c_int64(N), ptr(p_c), ptr(g_feat), ptr(g_params), ptr(g_pos),
                                                          c_float(mlp_scale * enc_scale), c_float(mlp_scale), c_int64(W), c_int64(H), ptr(bins[0]),
                                                          _lib.c_void_p(bins[0].data_ptr() + 4 * (bins[0].numel() - 1)),      # spill word = the last one
                                                          ptr(bins[1]), c_int64(bins[2]), stream()), "gs_hashgrid_encode_bwd_binned")
                else:
                    check(L.gs_hashgrid_encode_bwd(c_int(cfg[0]), c_int(cfg[1]), c_int(cfg[2]), c_int(cfg[3]), c_float(cfg[4]), ptr(pos_c), ptr(ab),
                                                   ptr(m_c), c_int64(N), ptr(p_c), ptr(g_feat), ptr(g_params), ptr(g_pos),
                                                   c_float(mlp_scale * enc_scale), c_float(mlp_scale), c_int64(W), c_int64(H), stream()),
                          "gs_hashgrid_encode_bwd")
        return g_pos, g_params, None, g_w1, g_w2, g_w3, None, None, None, None, None, None, None, None


class HashGridEncoding(torch.nn.Module):
    """Stand-in for `tcnn.Encoding(3, {"otype": "HashGrid", ...})`: same config keys, `.params`,
    `.n_output_dims`; parameters are fp32 and initialised U(-1e-4, 1e-4) like tiny-cuda-nn."""

    def __init__(self, n_input_dims, encoding_config, seed=1337):
        super().__init__()
        if n_input_dims != 3 or encoding_config.get("otype") != "HashGrid":
            raise NotImplementedError("only the 3-D HashGrid encoding used by MLPTexture3D is implemented")
        c = encoding_config
        self.cfg = (int(c["n_levels"]), int(c["n_features_per_level"]), int(c["log2_hashmap_size"]), int(c["base_resolution"]),
                    float(c["per_level_scale"]))
        self.n_output_dims = self.cfg[0] * self.cfg[1]
        n = _lib.lib().gs_hashgrid_num_params(c_int(self.cfg[0]), c_int(self.cfg[1]), c_int(self.cfg[2]), c_int(self.cfg[3]), c_float(self.cfg[4]))
        if n < 0:
            raise _lib.GShellHipError(_lib.lib().gs_last_error().decode())
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter((torch.rand(n, generator=g) * 2e-4 - 1e-4).cuda())

    def forward(self, x, mask=None):
        return _HashGridFn.apply(x, self.params, mask, self.cfg)


class _MLP(torch.nn.Module):
    """32-wide bias-free ReLU MLP (render/mlptexture.py:18-44)."""

    def __init__(self, cfg, loss_scale=1.0):
        super().__init__()
        self.loss_scale = loss_scale
        net = (torch.nn.Linear(cfg['n_input_dims'], cfg['n_neurons'], bias=False), torch.nn.ReLU())
        for _ in range(cfg['n_hidden_layers'] - 1):
            net = net + (torch.nn.Linear(cfg['n_neurons'], cfg['n_neurons'], bias=False), torch.nn.ReLU())
        net = net + (torch.nn.Linear(cfg['n_neurons'], cfg['n_output_dims'], bias=False),)
        self.net = torch.nn.Sequential(*net).cuda()
        self.net.apply(self._init_weights)

    def forward(self, x):
        # reference: full-backward hook scaling the gradient w.r.t. the MLP input by loss_scale (:31)
        return self.net(_ScaleGrad.apply(x.to(torch.float32), self.loss_scale))

    def fusable(self, x):
        lin = [m for m in self.net if isinstance(m, torch.nn.Linear)]
        return (x.is_cuda and len(lin) == 3 and tuple(lin[0].weight.shape) == (32, 32) and tuple(lin[1].weight.shape) == (32, 32)
                and lin[2].weight.shape[1] == 32 and lin[2].weight.shape[0] <= 8)

    def forward_mapped(self, x, mask, lo, hi):
        """sigmoid(net(x)) * (hi - lo) + lo; rows with mask <= 0 take the value of an all-zero feature row."""
        if not self.fusable(x):
            return torch.sigmoid(self.forward(x)) * (hi - lo)[None, :] + lo[None, :]
        lin = [m for m in self.net if isinstance(m, torch.nn.Linear)]
        return _TexMlpFn.apply(_ScaleGrad.apply(x.to(torch.float32), self.loss_scale), mask, lin[0].weight, lin[1].weight, lin[2].weight, lo, hi)

    @staticmethod
    def _init_weights(m):
        if type(m) == torch.nn.Linear:
            torch.nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')


class MLPTexture3D(torch.nn.Module):
    def __init__(self, AABB, channels=3, internal_dims=32, hidden=2, min_max=None, use_float16=False):
        super().__init__()
        if use_float16:
            raise NotImplementedError("use_float16=True is not supported (fp32 hash grid + fp32 texture MLP)")
        self.channels, self.internal_dims, self.AABB, self.min_max, self.use_float16 = channels, internal_dims, AABB, min_max, use_float16
        desired_resolution, base_grid_resolution, num_levels = 4096, 16, 16
        per_level_scale = np.exp(np.log(desired_resolution / base_grid_resolution) / (num_levels - 1))
        enc_cfg = {"otype": "HashGrid", "n_levels": num_levels, "n_features_per_level": 2, "log2_hashmap_size": 19,
                   "base_resolution": base_grid_resolution, "per_level_scale": per_level_scale}
        gradient_scaling = 128.0
        self.gradient_scaling = gradient_scaling
        self.encoder = HashGridEncoding(3, enc_cfg)
        mlp_cfg = {"n_input_dims": self.encoder.n_output_dims, "n_output_dims": self.channels, "n_hidden_layers": hidden,
                   "n_neurons": self.internal_dims}
        self.net = _MLP(mlp_cfg, gradient_scaling)

    def sample(self, texc, mask=None):
        """texc [...,3] world positions -> [..., channels].  `mask` (optional, [...]) skips rows whose value cannot
        reach any output (background pixels); the reference evaluates them and then discards them."""
        if self._fusable(texc)[0]:
            return self.sample_many([texc], mask)[0]
        _texc = (texc.view(-1, 3) - self.AABB[0][None, ...]) / (self.AABB[1][None, ...] - self.AABB[0][None, ...])
        _texc = torch.clamp(_texc, min=0, max=1)
        # reference: encoder backward hook divides the gradient w.r.t. the encoder input by 128 (:74)
        p_enc = self.encoder(_ScaleGrad.apply(_texc.contiguous(), 1.0 / self.gradient_scaling), mask)
        out = self.net.forward_mapped(p_enc, mask, self.min_max[0], self.min_max[1])
        return out.view(*texc.shape[:-1], self.channels)

    def _aabb_tensor(self):
        ab = getattr(self, "_aabb_dev", None)
        if ab is None or ab[0] is not self.AABB:
            t = self.AABB if torch.is_tensor(self.AABB) else torch.stack((self.AABB[0], self.AABB[1]))
            ab = self._aabb_dev = (self.AABB, t.detach().float().contiguous())
        return ab[1]

    def _fusable(self, texc):
        lin = [m for m in self.net.net if isinstance(m, torch.nn.Linear)]
        return (texc.is_cuda and self.encoder.cfg[1] == 2 and self.encoder.cfg[0] == 16 and self.net.fusable(texc) and
                getattr(self, "fused_field", True)), lin

    def sample_many(self, texcs, mask=None):
        """sample() of several coordinate sets that share one mask (render.py:68,70 samples the jittered and the plain
        g-buffer position) as ONE launch sequence over the concatenated rows: one table-gradient buffer, one pass of
        the backward kernels.  -> list of [..., channels]"""
        ok, lin = self._fusable(texcs[0])
        if not ok:
            return [self.sample(t, mask) for t in texcs]
        shp = texcs[0].shape
        k = len(texcs)
        pos = torch.cat([t.reshape(-1, 3) for t in texcs], 0) if k > 1 else texcs[0].reshape(-1, 3)
        m = None if mask is None else (mask.reshape(-1).float().repeat(k) if k > 1 else mask.reshape(-1))
        img = (int(shp[-3]), int(shp[-2])) if texcs[0].dim() >= 3 else None
        out = _FieldFn.apply(pos, self.encoder.params, m, lin[0].weight, lin[1].weight, lin[2].weight, self.min_max[0], self.min_max[1],
                             self._aabb_tensor(),
                             self.encoder.cfg, self.net.loss_scale, 1.0 / self.gradient_scaling, img, k)
        return [o.view(*shp[:-1], self.channels) for o in ((out,) if k == 1 else out)]

    def clamp_(self):
        pass

    def cleanup(self):
        pass
