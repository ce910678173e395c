"""`render.optixutils` surface of the reference (render/optixutils/ops.py:128-147) on MI355X:
software BVH + Monte-Carlo environment shading with shadow rays + bilateral denoiser, all HIP
(gshell_amd/csrc/{bvh,envshade}.hip) behind the C ABI.  No OptiX, no NVRTC, nothing compiled at import."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import c_float, c_int, c_int64, c_void_p, check, ptr, stream

c_uint32 = ctypes.c_uint32
PERM_ROWS = 32768          # rows of the stratification permutation table (ops.py:89)
_BSDF_IDS = ['pbr', 'diffuse', 'white']   # ordering must match the kernel (ops.py:142)


class OptiXContext:
    """Holds the shadow-ray acceleration structure of the current mesh (name kept for drop-in use;
    there is no OptiX here).  One context per device."""

    def __init__(self):
        self._h = c_void_p(0)
        check(_lib.lib().gs_bvh_create(ctypes.byref(self._h)), "gs_bvh_create")
        self.num_tris = 0

    @property
    def handle(self):
        return self._h

    def info(self):
        T, d, l, b = c_int64(), c_int64(), c_int64(), c_int64()
        check(_lib.lib().gs_bvh_info(self._h, ctypes.byref(T), ctypes.byref(d), ctypes.byref(l), ctypes.byref(b)))
        return dict(T=T.value, depth=d.value, leaf_size=l.value, bytes=b.value)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().gs_bvh_destroy(self._h)
                self._h = c_void_p(0)
        except Exception:
            pass


def optix_build_bvh(optix_ctx, verts, tris, rebuild):
    """verts [V,3] f32, tris [T,3] i32; empty meshes are legal (ops.py:133-139).  `rebuild` is accepted
    for signature compatibility; the structure is always rebuilt (the reference passes rebuild=1)."""
    verts = verts.detach().reshape(-1, 3).contiguous().float()
    tris = tris.reshape(-1, 3).contiguous()
    if tris.dtype != torch.int32:
        tris = tris.int()
    with torch.cuda.device(verts.device):
        check(_lib.lib().gs_bvh_build(optix_ctx.handle, ptr(verts, torch.float32, "verts"), c_int64(verts.shape[0]), ptr(tris, torch.int32, "tris"),
                                      c_int64(tris.shape[0]), stream()), "gs_bvh_build")
    optix_ctx.num_tris = int(tris.shape[0])


def any_hit(optix_ctx, origins, dirs):
    """Stand-alone shadow query: uint8 [n], 1 = occluded."""
    o, d = origins.detach().reshape(-1, 3).contiguous().float(), dirs.detach().reshape(-1, 3).contiguous().float()
    hit = torch.empty((o.shape[0],), dtype=torch.uint8, device=o.device)
    with torch.cuda.device(o.device):
        check(_lib.lib().gs_bvh_any_hit(optix_ctx.handle, ptr(o), ptr(d), c_int64(o.shape[0]), ptr(hit), stream()), "gs_bvh_any_hit")
    return hit


_random_perm = {}


def random_perm(n_samples_x, device):
    """[32768, n^2] int32 table of random permutations that decorrelates the light / BSDF strata
    (ops.py:86-89).  Seeded, so every rank of a view-sharded job holds the same table."""
    key = (n_samples_x, str(device))
    if key not in _random_perm:
        g = torch.Generator(device="cpu").manual_seed(0x6553 + n_samples_x)
        _random_perm[key] = torch.argsort(torch.rand(PERM_ROWS, n_samples_x * n_samples_x, generator=g), dim=-1).int().to(device)
    return _random_perm[key]


def set_random_perm(n_samples_x, table):
    _random_perm[(n_samples_x, str(table.device))] = table.int().contiguous()


class CoveredPixels:
    """The covered-pixel list of a rasterised frame, requested AHEAD of its use: `rast` [B,H,W,4] -> ascending linear indices of the pixels with a
    triangle (channel 3 > 0), compacted on the device (gs_compact_rows_strided, no channel copy); the COUNT travels to a pinned host word by an
    asynchronous copy and is read where the shader needs it (`resolve`).  The reference pays a synchronisation for the same quantity at the same
    place (its mask-dependent launches); here the request is issued right after rasterisation, so that by the time the shader asks, the count has
    long arrived and the host never drains the queue (the GPU used to idle ~100 us behind that read-back: profiles/r04_gpu_gaps.txt)."""
    _pinned = {}
    RING = 8

    def __init__(self, rast):
        L = _lib.lib()
        B, H, W, C = rast.shape
        assert C == 4 and rast.is_contiguous() and rast.dtype == torch.float32
        dev = rast.device
        n = B * H * W
        self.key = (rast.data_ptr(), rast._version, (B, H, W))
        self.rows = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        self.count_dev = torch.empty(2, dtype=torch.int64, device=dev)
        scratch = torch.empty(int(L.gs_compact_rows_scratch_bytes(c_int64(n))) // 4 + 4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(L.gs_compact_rows_strided(c_void_p(rast.data_ptr() + 12), c_int64(4), c_int64(n), c_int64(n), ptr(scratch), ptr(self.rows), c_void_p(0),
                                            ptr(self.count_dev), stream()), "gs_compact_rows_strided")
            # a ring of pinned count words per device: a request that is still unread when the next one is issued keeps its own word
            ring = CoveredPixels._pinned.get(str(dev))
            if ring is None:
                ring = CoveredPixels._pinned[str(dev)] = [torch.zeros(CoveredPixels.RING, 2, dtype=torch.int64).pin_memory(), 0]
            self.count_host = ring[0][ring[1] % CoveredPixels.RING]
            ring[1] += 1
            self.count_host.copy_(self.count_dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def matches(self, mask, dims):
        """`mask` is channel 3 of the very tensor the request was made for, unmodified since (same storage, same version counter)"""
        return (mask.data_ptr() == self.key[0] + 12 and mask._version == self.key[1] and tuple(dims) == self.key[2] and mask.dim() == 3
                and mask.stride() == (dims[1] * dims[2] * 4, dims[2] * 4, 4))

    def resolve(self):
        self.event.synchronize()
        return self.rows[: int(self.count_host[0])]


PENDING_PIXELS = None           # the CoveredPixels request of the frame being rendered (render.render_mesh issues it right after rasterisation)
last_covered_pixels = None      # covered pixels of the last optix_env_shade call (bench.py: rays per second)
SAVED_SAMPLES = True      # backward from the forward pass's saved ray buffer (gs_env_shade_bwd_saved); False = the one-kernel sampler replay of round 1
                          # (gs_env_shade_bwd: an oracle kernel, present in lib/variants/oracles.so only -- tests compare the two).
                          # The buffer (40 B per ray: ~0.8 GB at 4 x 512^2, n = 8, 15 % coverage) stays alive from forward to backward;
                          # the first backward overwrites it in place, a second one (retain_graph) replays the sampler instead.
SCRATCH_BOUND = 16 << 30   # bytes of per-sample records one env-shade call may hold (5.5 % of the MI355X's 288 GB).  A frame above it -- e.g. 4 views of
                          # the reference's default render size, 1024^2 at n_samples 24 (configs/deepfashion_mc_256.json: 1152 rays per covered pixel
                          # and pass, ~28 GB of records) -- is shaded in chunks of covered pixels through ONE scratch of this size
                          # (gs_env_shade_fwd_bounded: bit-identical outputs) and back-propagated chunk by chunk too (gs_env_shade_bwd_bounded: the
                          # sampler regenerates a chunk's records, no rays): no records kept.  Measured on MI355X at 2 x 1024^2, n = 24 (13.8 GB of
                          # records): 83 ms / iteration with the records kept, 103 ms through a 2 GiB scratch (peak memory 20 GB -> 8 GB).
last_bounded = False       # whether the last forward call took the bounded path (bench.py / tests)


def _padded(n):
    """allocation size for a coverage-dependent buffer of n elements: rounded up to 1/8 of its leading power of two (<= 12.5 % more), so that frames
    whose covered-pixel counts differ by a few per cent ask torch's caching allocator for the SAME block sizes and no first-size hipMalloc lands in a
    later iteration (round 5: one 5-step batch of the close-camera run read 50 ms per step instead of 25 on the driver's box)"""
    n = max(int(n), 1)
    g = 1 << max(n.bit_length() - 4, 0)
    return (n + g - 1) // g * g


class _optix_env_shade_func(torch.autograd.Function):
    @staticmethod
    def _launch_fwd(optix_ctx, pix, t, view, lgt, t_pdf, t_rows, t_cols, perms, BSDF, n, seed, shadow_scale, dims, vis, diff, spec, view_map=(0, 1)):
        L = _lib.lib()
        B, H, W = dims
        n_cov = pix.shape[0]
        need = max(int(L.gs_env_shade_scratch_bytes(c_int64(n_cov), c_int(n))), 8)
        global last_bounded
        last_bounded = SCRATCH_BOUND is not None and need > SCRATCH_BOUND
        if last_bounded:
            # chunks of covered pixels through ONE scratch of SCRATCH_BOUND bytes; the records do not survive the call (-> None: replay backward)
            min_bytes = 64 * 2 * n * n * 40 + 256
            nbytes = max(int(SCRATCH_BOUND), min_bytes)
            scratch = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=pix.device)
            check(L.gs_env_shade_fwd_bounded(optix_ctx.handle, ptr(pix, torch.int32), c_int64(n_cov), ptr(t["ro"]), ptr(t["pos"]), ptr(t["nrm"]), ptr(view),
                                             t["kd_ptr"], t["ks_ptr"], ptr(lgt), ptr(t_pdf), ptr(t_rows), ptr(t_cols), c_int64(lgt.shape[0]), c_int64(lgt.shape[1]),
                                             ptr(perms, torch.int32), c_int64(perms.shape[0]), c_int64(B), c_int64(H), c_int64(W), c_int64(view_map[0]),
                                             c_int64(view_map[1]), c_int(BSDF), c_int(n), c_uint32(seed & 0xFFFFFFFF), c_float(shadow_scale), ptr(scratch),
                                             c_int64(nbytes), ptr(vis), ptr(diff), ptr(spec), stream()), "gs_env_shade_fwd_bounded")
            return None
        scratch = torch.empty(_padded((need + 7) // 8), dtype=torch.int64, device=pix.device)
        check(L.gs_env_shade_fwd(optix_ctx.handle, ptr(pix, torch.int32), c_int64(n_cov), ptr(t["ro"]), ptr(t["pos"]), ptr(t["nrm"]), ptr(view),
                                 t["kd_ptr"], t["ks_ptr"], ptr(lgt), ptr(t_pdf), ptr(t_rows), ptr(t_cols), c_int64(lgt.shape[0]), c_int64(lgt.shape[1]),
                                 ptr(perms, torch.int32), c_int64(perms.shape[0]), c_int64(B), c_int64(H), c_int64(W), c_int64(view_map[0]), c_int64(view_map[1]),
                                 c_int(BSDF), c_int(n), c_uint32(seed & 0xFFFFFFFF), c_float(shadow_scale), ptr(scratch), ptr(vis), ptr(diff), ptr(spec), stream()),
              "gs_env_shade_fwd")
        return scratch

    @staticmethod
    def forward(ctx, optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, BSDF, n_samples_x, rnd_seed,
                shadow_scale, view_map=(0, 1), kd_ks=None):
        L = _lib.lib()
        _rnd_seed = int(np.random.randint(2 ** 31)) if rnd_seed is None else int(rnd_seed)
        B, H, W, _ = gb_pos.shape
        dev = gb_pos.device
        full = (B, H, W, 3)

        def c3(t):
            return t.detach().expand(full).contiguous().float()
        # covered pixels, ascending.  The count is needed on the host (scratch sizes, grids) like in the reference's mask-dependent launches; when the
        # frame's list was requested ahead (CoveredPixels, issued by render_mesh right after rasterisation) it has arrived by now: no queue drain
        global PENDING_PIXELS
        pend, PENDING_PIXELS = PENDING_PIXELS, None
        if pend is not None and pend.matches(mask, (B, H, W)):
            pix = pend.resolve()
        else:
            pix = torch.nonzero(mask.detach().expand(B, H, W).reshape(-1) > 0).reshape(-1).int()
        global last_covered_pixels
        last_covered_pixels = int(pix.shape[0])
        t = dict(ro=None if ro is None else c3(ro), pos=c3(gb_pos), nrm=c3(gb_normal))      # ro None: gb_pos + gb_normal * 0.001 inside the kernel
        if kd_ks is not None:      # kd | ks as the channel halves of one [B,H,W,6] tensor (gs_env_shade_*: ks = kd + 3 -> pixel stride 6)
            if tuple(kd_ks.shape) != (B, H, W, 6):
                raise _lib.GShellHipError(f"kd_ks must be [B,H,W,6], got {tuple(kd_ks.shape)}")
            t["tex"] = kd_ks.detach().contiguous().float()
            t["kd_ptr"], t["ks_ptr"] = ptr(t["tex"]), c_void_p(t["tex"].data_ptr() + 12)
        else:
            t["kd"], t["ks"] = c3(gb_kd), c3(gb_ks)
            t["kd_ptr"], t["ks_ptr"] = ptr(t["kd"]), ptr(t["ks"])
        if tuple(gb_view_pos.shape) not in ((B, 1, 1, 3), (1, 1, 1, 3)):
            raise _lib.GShellHipError(f"gb_view_pos must be [B,1,1,3] (one eye per view), got {tuple(gb_view_pos.shape)}")
        view = gb_view_pos.detach().expand(B, 1, 1, 3).reshape(B, 3).contiguous().float()
        lgt, t_pdf, t_rows, t_cols = (x.detach().contiguous().float() for x in (light, pdf, rows, cols))
        perms = random_perm(n_samples_x, dev)
        diff = torch.empty(full, dtype=torch.float32, device=dev)
        spec = torch.empty(full, dtype=torch.float32, device=dev)
        vis = torch.empty((_padded(int(L.gs_env_shade_vis_words(c_int64(pix.shape[0]), c_int(n_samples_x)))),), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            scratch = _optix_env_shade_func._launch_fwd(optix_ctx, pix, t, view, lgt, t_pdf, t_rows, t_cols, perms, BSDF, n_samples_x, _rnd_seed,
                                                        shadow_scale, (B, H, W), vis, diff, spec, view_map)
        # the ray buffer (direction + MIS weight per sample, 40 B / ray) stays alive for the backward pass, which then needs no
        # RNG replay (SAVED_SAMPLES = False: regenerate the samples, the round-1 path; same gradients)
        ctx.scratch = scratch if (SAVED_SAMPLES and any(ctx.needs_input_grad)) else None
        ctx.args = (optix_ctx, pix, t, view, lgt, t_pdf, t_rows, t_cols, perms, BSDF, n_samples_x, rnd_seed, _rnd_seed, shadow_scale, vis)
        ctx.view_map = view_map
        ctx.shapes = (gb_pos.shape, gb_normal.shape, None if kd_ks is not None else gb_kd.shape, None if kd_ks is not None else gb_ks.shape, light.shape)
        return diff, spec

    @staticmethod
    def backward(ctx, diff_grad, spec_grad):
        optix_ctx, pix, t, view, lgt, t_pdf, t_rows, t_cols, perms, BSDF, n_samples_x, rnd_seed, _fwd_seed, shadow_scale, vis = ctx.args
        B, H, W, _ = t["pos"].shape
        dev = t["pos"].device
        gd, gs = diff_grad.contiguous().float(), spec_grad.contiguous().float()
        g_pos, g_nrm = (torch.empty((B, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2))
        packed = "tex" in t
        if packed:
            g_tex = torch.empty((B, H, W, 6), dtype=torch.float32, device=dev)
            g_kd_ptr, g_ks_ptr = ptr(g_tex), c_void_p(g_tex.data_ptr() + 12)
        else:
            g_kd, g_ks = (torch.empty((B, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2))
            g_kd_ptr, g_ks_ptr = ptr(g_kd), ptr(g_ks)
        g_light = torch.zeros_like(lgt)
        with torch.cuda.device(dev):
            if rnd_seed is None:
                # "decorrelated" mode (ops.py:100): the backward pass draws a fresh seed -> new rays: trace them (no shading outputs)
                _rnd_seed = int(np.random.randint(2 ** 31))
                vis = torch.empty_like(vis)
                scratch = _optix_env_shade_func._launch_fwd(optix_ctx, pix, t, view, lgt, t_pdf, t_rows, t_cols, perms, BSDF, n_samples_x, _rnd_seed,
                                                            shadow_scale, (B, H, W), vis, None, None, ctx.view_map)
                if not SAVED_SAMPLES:
                    scratch = None
            else:
                _rnd_seed = _fwd_seed      # same seed -> same rays -> the cached visibility bits are exact
                scratch = ctx.scratch
            need = max(int(_lib.lib().gs_env_shade_scratch_bytes(c_int64(pix.shape[0]), c_int(n_samples_x))), 8)
            if scratch is not None:
                fn, extra = _lib.lib().gs_env_shade_bwd_saved, (ptr(scratch),)
            elif SAVED_SAMPLES:
                # no records at hand -- the frame was shaded in chunks (gs_env_shade_fwd_bounded), or this is a second backward through a retained
                # graph (the first consumed them in place): the sampler regenerates them (no rays: the visibility bits are cached) into a scratch
                # of at most SCRATCH_BOUND bytes, chunk by chunk
                nbytes = need if (SCRATCH_BOUND is None or need <= SCRATCH_BOUND) else max(int(SCRATCH_BOUND), 64 * 2 * n_samples_x * n_samples_x * 40 + 256)
                scratch = torch.empty(_padded((nbytes + 7) // 8), dtype=torch.int64, device=dev)
                fn, extra = _lib.lib().gs_env_shade_bwd_bounded, (ptr(scratch), c_int64(nbytes))
            else:
                fn, extra = _lib.lib().gs_env_shade_bwd, ()      # sampler replay in one kernel (round 1's path): lib/variants/oracles.so only
            check(fn(optix_ctx.handle, ptr(pix), c_int64(pix.shape[0]), ptr(t["pos"]), ptr(t["nrm"]), ptr(view), t["kd_ptr"],
                                              t["ks_ptr"], ptr(lgt), ptr(t_pdf), ptr(t_rows), ptr(t_cols), c_int64(lgt.shape[0]),
                                              c_int64(lgt.shape[1]), ptr(perms), c_int64(perms.shape[0]), c_int64(B), c_int64(H), c_int64(W),
                                              c_int64(ctx.view_map[0]), c_int64(ctx.view_map[1]), c_int(BSDF), c_int(n_samples_x), c_uint32(_rnd_seed & 0xFFFFFFFF), c_float(shadow_scale), ptr(vis),
                                              *extra, ptr(gd), ptr(gs), ptr(g_pos), ptr(g_nrm), g_kd_ptr, g_ks_ptr, ptr(g_light), stream()),
                  "gs_env_shade_bwd")
        ctx.scratch = None
        s = ctx.shapes

        def red(g, shape):
            return g if tuple(shape) == tuple(g.shape) else g.sum_to_size(shape)
        if packed:
            return (None, None, None, red(g_pos, s[0]), red(g_nrm, s[1]), None, None, None, g_light, None, None, None, None, None, None, None, None, g_tex)
        return (None, None, None, red(g_pos, s[0]), red(g_nrm, s[1]), None, red(g_kd, s[2]), red(g_ks, s[3]), g_light, None, None, None, None, None,
                None, None, None, None)


def optix_env_shade(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, BSDF='pbr', n_samples_x=8,
                    rnd_seed=None, shadow_scale=1.0, view_offset=0, view_stride=1, kd_ks=None):
    """-> (diffuse [B,H,W,3], specular [B,H,W,3]) demodulated radiance (ops.py:141-143).
    `view_offset`, `view_stride` (not in the reference, which is single-GPU): local view b is view b*stride + offset of the
    global batch -- the sampler hashes the GLOBAL pixel index so that view-sharded ranks draw the single-GPU samples.
    `kd_ks` (not in the reference): the [B,H,W,6] tensor whose channel halves gb_kd / gb_ks are (what MLPTexture3D.sample returns); the
    kernels then read it in place and the gradient comes back as one tensor (gb_kd / gb_ks are ignored).  `ro` None: gb_pos + 0.001 gb_normal."""
    iBSDF = _BSDF_IDS.index(BSDF)
    return _optix_env_shade_func.apply(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, None if kd_ks is not None else gb_kd,
                                       None if kd_ks is not None else gb_ks, light, pdf, rows, cols, iBSDF,
                                       n_samples_x, rnd_seed, shadow_scale, (int(view_offset), int(view_stride)), kd_ks)


def optix_env_shade_samples(optix_ctx, mask, ro, gb_pos, gb_normal, gb_view_pos, gb_kd, gb_ks, light, pdf, rows, cols, BSDF='pbr', n_samples_x=8,
                            rnd_seed=0, shadow_scale=1.0, view_offset=0, view_stride=1):
    """Diagnostics: the forward pass's per-sample records, for comparing sample by sample with the reference's raygen loop
    (kernel.cu:488-529).  -> pix [n_cov] (linear index of the covered pixels), dirs [n_cov, 2, S, 3] (0: light samples, 1: BSDF
    samples), k [n_cov, 2, S] (MIS weight x sample weight = 1 / max(pdf_light + pdf_bsdf, 1e-4) / n^2), live [n_cov, 2, S] (False:
    the unshadowed contribution is exactly zero and the ray was not traced), visible [n_cov, 2, S] (the cached shadow-ray bit)."""
    L = _lib.lib()
    B, H, W, _ = gb_pos.shape
    dev = gb_pos.device
    full = (B, H, W, 3)

    def c3(t):
        return t.detach().expand(full).contiguous().float()
    pix = torch.nonzero(mask.detach().expand(B, H, W).reshape(-1) > 0).reshape(-1).int()
    t = dict(ro=c3(ro), pos=c3(gb_pos), nrm=c3(gb_normal), kd=c3(gb_kd), ks=c3(gb_ks))
    t["kd_ptr"], t["ks_ptr"] = ptr(t["kd"]), ptr(t["ks"])
    view = gb_view_pos.detach().expand(B, 1, 1, 3).reshape(B, 3).contiguous().float()
    lgt, t_pdf, t_rows, t_cols = (x.detach().contiguous().float() for x in (light, pdf, rows, cols))
    perms = random_perm(n_samples_x, dev)
    S = n_samples_x * n_samples_x
    n_cov = int(pix.shape[0])
    vis = torch.zeros((int(L.gs_env_shade_vis_words(c_int64(n_cov), c_int(n_samples_x))),), dtype=torch.int64, device=dev)
    global SCRATCH_BOUND
    bound, SCRATCH_BOUND = SCRATCH_BOUND, None          # the records of EVERY pixel are what this call returns
    try:
        with torch.cuda.device(dev):
            scratch = _optix_env_shade_func._launch_fwd(optix_ctx, pix, t, view, lgt, t_pdf, t_rows, t_cols, perms, _BSDF_IDS.index(BSDF), n_samples_x,
                                                        int(rnd_seed), shadow_scale, (B, H, W), vis, None, None, (int(view_offset), int(view_stride)))
    finally:
        SCRATCH_BOUND = bound
    dk = scratch.view(torch.float32)[: n_cov * 2 * S * 4].reshape(n_cov, 2, S, 4)
    bits = (vis[:, None] >> torch.arange(64, device=dev)[None, :]) & 1
    visible = bits.reshape(-1)[: n_cov * 2 * S].reshape(n_cov, 2, S).bool()
    w = dk[..., 3]
    live = ~torch.signbit(w)
    return pix, dk[..., :3].clone(), w.abs(), live, visible


class _bilateral_denoiser_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, col, nrm, zdz, sigma, mask=None):
        col_c, nrm_c, zdz_c = (x.detach().contiguous().float() for x in (col, nrm, zdz))
        m_c = None if mask is None else mask.detach().reshape(col_c.shape[:3]).contiguous().float()
        B, H, W, _ = col_c.shape
        out = torch.empty((B, H, W, 4), dtype=torch.float32, device=col_c.device)
        with torch.cuda.device(col_c.device):
            check(_lib.lib().gs_bilateral_fwd_masked(ptr(col_c, torch.float32, "col"), ptr(nrm_c), ptr(zdz_c), ptr(m_c), c_int64(B), c_int64(H), c_int64(W),
                                                     c_float(sigma), ptr(out), stream()), "gs_bilateral_fwd_masked")
        ctx.save_for_backward(nrm_c, zdz_c, m_c)
        ctx.sigma = sigma
        return out

    @staticmethod
    def backward(ctx, out_grad):
        nrm_c, zdz_c, m_c = ctx.saved_tensors
        B, H, W, _ = nrm_c.shape
        g = out_grad.contiguous().float()
        g_col = torch.empty((B, H, W, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(_lib.lib().gs_bilateral_bwd_masked(ptr(nrm_c), ptr(zdz_c), ptr(m_c), c_int64(B), c_int64(H), c_int64(W), c_float(ctx.sigma), ptr(g),
                                                     ptr(g_col), stream()), "gs_bilateral_bwd_masked")
        return g_col, None, None, None, None


class _bilateral_denoiser_pair_func(torch.autograd.Function):
    """two colour images, one set of guides, one kernel each way (gs_bilateral_*_masked2): the filter weights are shared"""

    @staticmethod
    def forward(ctx, col_a, col_b, nrm, zdz, sigma, mask=None):
        ca, cb, nrm_c, zdz_c = (x.detach().contiguous().float() for x in (col_a, col_b, nrm, zdz))
        m_c = None if mask is None else mask.detach().reshape(ca.shape[:3]).contiguous().float()
        B, H, W, _ = ca.shape
        out_a = torch.empty((B, H, W, 4), dtype=torch.float32, device=ca.device)
        out_b = torch.empty_like(out_a)
        with torch.cuda.device(ca.device):
            check(_lib.lib().gs_bilateral_fwd_masked2(ptr(ca, torch.float32, "col_a"), ptr(cb, torch.float32, "col_b"), ptr(nrm_c), ptr(zdz_c), ptr(m_c),
                                                      c_int64(B), c_int64(H), c_int64(W), c_float(sigma), ptr(out_a), ptr(out_b), stream()),
                  "gs_bilateral_fwd_masked2")
        ctx.save_for_backward(nrm_c, zdz_c, m_c)
        ctx.sigma = sigma
        return out_a, out_b

    @staticmethod
    def backward(ctx, ga, gb):
        nrm_c, zdz_c, m_c = ctx.saved_tensors
        B, H, W, _ = nrm_c.shape
        ga, gb = ga.contiguous().float(), gb.contiguous().float()
        g_a = torch.empty((B, H, W, 3), dtype=torch.float32, device=ga.device)
        g_b = torch.empty_like(g_a)
        with torch.cuda.device(ga.device):
            check(_lib.lib().gs_bilateral_bwd_masked2(ptr(nrm_c), ptr(zdz_c), ptr(m_c), c_int64(B), c_int64(H), c_int64(W), c_float(ctx.sigma), ptr(ga),
                                                      ptr(gb), ptr(g_a), ptr(g_b), stream()), "gs_bilateral_bwd_masked2")
        return g_a, g_b, None, None, None, None


def bilateral_denoiser_raw_pair(col_a, col_b, nrm, zdz, sigma, mask=None):
    """bilateral_denoiser_raw of two images that share their guides (and the filter weights): one launch each way"""
    return _bilateral_denoiser_pair_func.apply(col_a, col_b, nrm, zdz, sigma, mask)


def bilateral_denoiser_raw(col, nrm, zdz, sigma, mask=None):
    """[B,H,W,4] = (sum w c, max(sum w, 1e-4)): the kernel's own output (denoising.cu:66-70), before the reference's division.
    `mask` [B,H,W(,1)]: only pixels with mask > 0 are consumed by the caller (covered pixels); the others are written as
    (0,0,0,1e-4) without being filtered."""
    return _bilateral_denoiser_func.apply(col, nrm, zdz, sigma, mask)


def bilateral_denoiser(col, nrm, zdz, sigma):
    col_w = _bilateral_denoiser_func.apply(col, nrm, zdz, sigma)
    return col_w[..., 0:3] / col_w[..., 3:4]
