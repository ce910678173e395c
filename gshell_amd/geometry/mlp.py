"""SDF network of G-Shell (reference geometry/mlp.py:7-40 + geometry/embedding.py:4-39):
positional encoding (x, sin(2^k x), cos(2^k x))_{k<n_freq} -> Linear+Softplus(beta=100) stack with an optional
skip-concatenation of the encoding -> Linear(d_hidden, d_out).  Module / parameter names match the reference so that
`state_dict()` round-trips ("sdf_net.net.<i>.weight").

This file is the plain-torch (rocBLAS) formulation; the fused MFMA kernel path is gshell_amd/csrc/mlp.hip."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .._lib import c_int, c_int64, check, ptr, stream


class Embedding(nn.Module):
    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs, self.in_channels = N_freqs, in_channels
        self.out_channels = in_channels * (2 * N_freqs + 1)
        if logscale:
            self.freq_bands = [2.0 ** k for k in range(N_freqs)]
        else:
            self.freq_bands = torch.linspace(1, 2 ** (N_freqs - 1), N_freqs).tolist()

        self.register_buffer("_freqs", torch.tensor(self.freq_bands, dtype=torch.float32).view(-1, 1), persistent=False)

    def forward(self, x):
        # (x, sin(f0 x), cos(f0 x), sin(f1 x), ...) as 5 launches instead of 3 per frequency: same products, same order
        fx = x.unsqueeze(-2) * self._freqs.to(x.dtype)                       # [..., F, C]
        sc = torch.stack((torch.sin(fx), torch.cos(fx)), dim=-2)             # [..., F, 2, C]
        return torch.cat((x, sc.reshape(*x.shape[:-1], -1)), -1)


def _splitk_tn(a, b, split):
    """a^T @ b for a [R,m], b [R,n] with R >> m, n: `split` slabs as one batched GEMM + a reduction."""
    R = a.shape[0]
    R0 = (R // split) * split
    out = torch.bmm(a[:R0].reshape(split, R0 // split, -1).transpose(1, 2), b[:R0].reshape(split, R0 // split, -1)).sum(0)
    if R0 < R:
        out = out + a[R0:].t() @ b[R0:]
    return out


class _SplitKMatmulFn(torch.autograd.Function):
    """y = a @ w whose gradient w.r.t. w (again a^T @ gy with K = rows) is split-K.  Used for the input-gradient product
    inside _SplitKLinearFn.backward so that the DOUBLE backward of the eikonal term does not fall back to a single
    256 x 256-output GEMM over 50 k rows (measured 1.25 ms as one hipBLASLt call on 32 workgroups)."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return a @ w

    @staticmethod
    def backward(ctx, gy):
        a, w = ctx.saved_tensors
        g_a = gy @ w.t() if ctx.needs_input_grad[0] else None
        g_w = _splitk_tn(a, gy, _SplitKLinearFn.SPLIT) if ctx.needs_input_grad[1] else None
        return g_a, g_w


class _SplitKLinearFn(torch.autograd.Function):
    """F.linear whose weight gradient is a split-K batched GEMM.  dW = g^T x has a 256 x 256 output and K = rows
    (10^5..10^6): as ONE GEMM hipBLASLt covers the output with 32 workgroups (MT32x64) and leaves 7/8 of the chip idle
    -- measured 1.2-1.6 ms per layer stack.  Splitting the rows into SPLIT slabs turns it into a batch of SPLIT GEMMs
    plus one small reduction.  backward() is built from differentiable torch ops, so double backward (eikonal term) works."""
    SPLIT = 16
    MIN_ROWS = 16384

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g_x = _SplitKMatmulFn.apply(g, weight) if ctx.needs_input_grad[0] else None
        g_w = _splitk_tn(g, x, _SplitKLinearFn.SPLIT) if ctx.needs_input_grad[1] else None
        g_b = g.sum(0) if ctx.needs_input_grad[2] else None
        return g_x, g_w, g_b


class _SoftplusBwdFn(torch.autograd.Function):
    """g * softplus'(x) as one kernel, with a one-kernel backward of its own (the eikonal term differentiates it again)."""

    @staticmethod
    def forward(ctx, x, g, beta):
        x_c, g_c = x.detach().contiguous(), g.detach().contiguous()
        out = torch.empty_like(x_c)
        with torch.cuda.device(x_c.device):
            _lib.check(_lib.lib().gs_softplus_bwd(_lib.ptr(x_c, torch.float32, "x"), _lib.ptr(g_c, torch.float32, "g"), _lib.c_int64(x_c.numel()),
                                                  _lib.c_float(beta), _lib.ptr(out), _lib.stream()), "gs_softplus_bwd")
        ctx.save_for_backward(x_c, g_c)
        ctx.beta = beta
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        x_c, g_c = ctx.saved_tensors
        gg_c = gg.contiguous()
        d_g = torch.empty_like(x_c) if ctx.needs_input_grad[1] else None
        d_x = torch.empty_like(x_c) if ctx.needs_input_grad[0] else None
        if d_g is not None or d_x is not None:
            with torch.cuda.device(x_c.device):
                _lib.check(_lib.lib().gs_softplus_bwd_bwd(_lib.ptr(x_c), _lib.ptr(g_c), _lib.ptr(gg_c, torch.float32, "gg"), _lib.c_int64(x_c.numel()),
                                                          _lib.c_float(ctx.beta), _lib.ptr(d_g), _lib.ptr(d_x), _lib.stream()), "gs_softplus_bwd_bwd")
        return d_x, d_g, None


class _SoftplusFn(torch.autograd.Function):
    """nn.Softplus(beta, threshold=20) on the HIP elementwise kernels (value / gradient / gradient of the gradient)."""

    @staticmethod
    def forward(ctx, x, beta):
        x_c = x.detach().contiguous()
        y = torch.empty_like(x_c)
        with torch.cuda.device(x_c.device):
            _lib.check(_lib.lib().gs_softplus_fwd(_lib.ptr(x_c, torch.float32, "x"), _lib.c_int64(x_c.numel()), _lib.c_float(beta), _lib.ptr(y),
                                                  _lib.stream()), "gs_softplus_fwd")
        ctx.save_for_backward(x)
        ctx.beta = beta
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return _SoftplusBwdFn.apply(x, g, ctx.beta), None


def _softplus(module, h):
    if h.is_cuda and h.dtype == torch.float32 and module.threshold == 20 and h.numel() >= (1 << 16):
        return _SoftplusFn.apply(h, float(module.beta))
    return module(h)


def _linear(module, h):
    if h.dim() == 2 and h.shape[0] >= _SplitKLinearFn.MIN_ROWS and h.is_cuda:
        return _SplitKLinearFn.apply(h, module.weight, module.bias)
    return module(h)


class MLP(nn.Module):
    def __init__(self, n_freq=6, d_hidden=128, d_out=1, n_hidden=3, skip_in=[], use_float16=False):
        super().__init__()
        self.emb = Embedding(3, n_freq)
        layers = [nn.Linear(self.emb.out_channels, d_hidden), nn.Softplus(beta=100)]
        self.skip_count, self.skip_in = [], skip_in
        for i in range(n_hidden):
            wide = i in skip_in
            if wide:
                self.skip_count.append(len(layers))
            layers += [nn.Linear(d_hidden + (self.emb.out_channels if wide else 0), d_hidden), nn.Softplus(beta=100)]
        layers.append(nn.Linear(d_hidden, d_out))
        self.net = nn.ModuleList(layers)
        if use_float16:     # reference geometry/mlp.py:36-38 autocasts; this path computes fp32 only -- never ignore the flag silently
            raise NotImplementedError("use_float16=True is not supported by the fp32 MFMA path (the G-Shell scripts set it False)")
        self.use_float16 = use_float16

    def forward(self, x):
        emb = self.emb(x)
        h = emb
        for i, module in enumerate(self.net):
            hin = torch.cat([h, emb], dim=-1) if i in self.skip_count else h
            h = _linear(module, hin) if isinstance(module, nn.Linear) else (_softplus(module, hin) if isinstance(module, nn.Softplus) else module(hin))
        return h


# Every time an SDF-network pass leaves the hand-written kernels (a network shape they do not cover, a CPU tensor of the gloo tests,
# or the exact-fp32 mode after an fp16-range overflow) it is counted here and announced once: bench.py refuses to report a number
# when the count is non-zero, so a figure can never come from the torch path unnoticed.
FALLBACKS = {}


def _note_fallback(what, why):
    n = FALLBACKS.get(what, 0)
    FALLBACKS[what] = n + 1
    if n == 0:
        import warnings
        warnings.warn(f"gshell_amd SDF network: {what} runs as plain torch ops, not as the HIP chain kernels ({why})")


def _shape_fusable(net, x):
    lin = [m for m in net.net if isinstance(m, nn.Linear)]
    return (x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3 and lin[-1].out_features == 1
            and all(m.out_features == 256 for m in lin[:-1]) and len(net.skip_count) <= 1 and net.emb.N_freqs <= 6
            and all(f == 2.0 ** k for k, f in enumerate(net.emb.freq_bands)))


def _fusable(net, x, what=None):
    """True: the fused kernels cover this call.  False on a GPU tensor is recorded (FALLBACKS) when `what` names the pass."""
    ok = x.is_cuda and _shape_fusable(net, x) and net.__dict__.get("_gs_precision") != "torch"
    if not ok and x.is_cuda and what:
        _note_fallback(what, "the exact-fp32 mode after an fp16-range overflow" if net.__dict__.get("_gs_precision") == "torch"
                       else "hidden width != 256, more than 6 frequencies or more than one skip connection")
    return ok


def pack_weights(net):
    """Flat fp32 buffer in the layout of gs_sdf_mlp_fwd (include/gshell_hip.h): k-major weights, zero-padded embedding rows."""
    lin = [m for m in net.net if isinstance(m, nn.Linear)]
    E = net.emb.out_channels
    Epad = (E + 7) // 8 * 8
    skip = -1
    if net.skip_count:
        skip = [i for i, m in enumerate(net.net) if isinstance(m, nn.Linear)].index(net.skip_count[0])
    parts = []
    for i, m in enumerate(lin[:-1]):
        wt = m.weight.detach().t()                                   # [K, 256]
        if i == 0 or i == skip:
            wt = torch.cat([wt, wt.new_zeros(Epad - E, wt.shape[1])], dim=0)
        parts += [wt.reshape(-1), m.bias.detach()]
    parts += [lin[-1].weight.detach().reshape(-1), lin[-1].bias.detach().reshape(-1)]
    return torch.cat(parts).contiguous().float(), len(lin) - 2, skip


# Arithmetic of the fused full-grid forward pass:
#   "h2"   (default) csrc/mlp_h2.hip -- fp16-pair operands on the f16 matrix path, three MFMAs per product, fp32 accumulate;
#          operands represented to 2^-22, measured max |err| vs float64 within 2x of the fp32 kernel's own
#   "fp32" csrc/mlp.hip -- v_mfma_f32_32x32x2_f32, bitwise a k-ordered fmaf chain, 4x slower; kept as the oracle of "h2"
SDF_MLP_PRECISION = "h2"
# weight gradients: False = bf16-pair operands on the bf16 matrix path (2^-16 per product, unbiased, summed over >= 10^5 rows);
# True = the exact-fp32 MFMA version (5x slower), kept as its oracle
SDF_MLP_WGRAD_FP32 = False


# Full-grid forward as TWO passes when the caller hands over the grid's topology (GShellTetsGeometry.getMesh): a one-product fp16 pass
# over every row (a third of the matrix work; error ~1e-4), then the three-product "h2" arithmetic on the rows that can matter --
# |sdf| < SDF_TWO_PASS_TAU, or an end point of an edge that changes sign or has such an end point.  With the one-product error below
# tau everywhere, every SIGN and the VALUE at both end points of every crossing edge equal the one-pass h2 result bit for bit; that
# is all the reference consumes (gshell_tets.py:250, :277-290; gshell_tets_geometry.py:33-39).  Off-surface values of the returned
# tensor carry the one-product error.  The second pass measures the error on ~1e5 rows per call; check_forward_status() compares it
# with tau (safety factor 4) and re-runs the one-pass kernel when it is not met.
SDF_TWO_PASS = True
SDF_TWO_PASS_TAU = 2e-3
SDF_TWO_PASS_SAFETY = 4.0
# AUDIT of the rows the second pass would NOT touch: the proof needs "first-pass error < |sdf|" at every
# UNREFINED row, and the refined rows are by construction the near-surface ones.  Every call therefore also re-evaluates a rotating
# sample of ALL rows -- every k-th row, k = N / SDF_TWO_PASS_AUDIT_ROWS, another residue class each call, so that every row of the grid
# is audited once per k calls -- and the kernel records, beside max |new - old|, the largest FRACTION OF ITS SIGN MARGIN any re-evaluated
# row used up: |new - old| / max(tau, |new|).  check_forward_status demands fraction * safety <= 1 (for near-surface rows that is the old
# rule max |dev| * safety <= tau).  Cost: ~16 k rows beside the ~90 k of the bench grid, +0.04 ms.  0 = no audit.
SDF_TWO_PASS_AUDIT_ROWS = 16384


def _layer_structure(net):
    lin = [m for m in net.net if isinstance(m, nn.Linear)]
    skip = -1
    if net.skip_count:
        skip = [i for i, m in enumerate(net.net) if isinstance(m, nn.Linear)].index(net.skip_count[0])
    return lin, len(lin) - 2, skip


def pack_weights_h2(net, status=None, out=None):
    """Device buffer in the layout of gs_sdf_mlp_fwd_h2: ONE launch that reads the module's parameters in place.
    status: optional int32 [2] device tensor (word 0 is raised for a weight beyond the fp16 range).
    out: pack into this int64 buffer instead of a fresh / cached one (tests: a pre-filled buffer shows which words the packer writes -- the section of
    the kernel variant that is not selected and the staging pad behind the image stay untouched)."""
    import ctypes
    lin, n_hidden, skip = _layer_structure(net)
    # the packed image of the SAME parameter values is reused (an iteration packs for the grid pass, the eikonal term and the row-sparse
    # backward: two of three launches and their host time -- right after the extraction's sync, where the GPU waits for the host).  Key: storage
    # and version counter of every parameter (torch's in-place updates and HipAdam.step both bump the version).  A call that asks for the
    # fp16-range check (`status`) always packs.
    # INVARIANT for every writer of the parameters: a write must bump the version counter (any torch in-place op does; a raw-pointer kernel or a
    # write through `.data` -- HipAdam.step, ViewShard.broadcast -- calls torch.autograd.graph.increment_version) or call invalidate_packed(net).
    # The cache lives OUTSIDE the module (weak table): copy.deepcopy(net) / state_dict round trips never carry a stale image along.
    key = tuple((p.data_ptr(), p._version) for m in lin for p in (m.weight, m.bias)) + (int(_lib._real_lib_fn().gs_sdf_mlp_h1_impl(c_int(-1))),)          # (the raw library: not an op to be timed)
    cached = _PACKED.get(net)
    if status is None and out is None and cached is not None and cached[0] == key:
        return cached[1], n_hidden, skip
    L = _lib.lib()
    nf = net.emb.N_freqs
    dev = lin[0].weight.device
    nbytes = int(L.gs_sdf_mlp_h2_packed_bytes(c_int(nf), c_int(n_hidden), c_int(skip)))
    packed = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=dev) if out is None else out
    if packed.numel() * 8 < nbytes or packed.dtype != torch.int64:
        raise _lib.GShellHipError("pack_weights_h2: `out` must be an int64 buffer of gs_sdf_mlp_h2_packed_bytes bytes")
    ws = [m.weight.detach() for m in lin]
    bs = [m.bias.detach() for m in lin]
    for t in ws + bs:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.GShellHipError("SDF network parameters must be contiguous fp32 tensors in HBM")
    PtrArr = ctypes.c_void_p * len(ws)
    with torch.cuda.device(dev):
        check(L.gs_sdf_mlp_h2_pack(PtrArr(*[t.data_ptr() for t in ws]), PtrArr(*[t.data_ptr() for t in bs]), c_int(nf), c_int(n_hidden), c_int(skip),
                                   ptr(packed), ptr(status), stream()), "gs_sdf_mlp_h2_pack")
    if out is None:
        _PACKED[net] = (key, packed)
    return packed, n_hidden, skip


import weakref      # noqa: E402
_PACKED = weakref.WeakKeyDictionary()      # net -> (key, packed image)


def invalidate_packed(net):
    """Drop the cached packed image of `net`: for a writer of its parameters that does not bump their version counters."""
    _PACKED.pop(net, None)


class ForwardStatus:
    """Device status words of one fused forward call + their asynchronous copy in pinned host memory."""

    def __init__(self, device, tau=None):
        self.dev = torch.zeros(3, dtype=torch.int32, device=device)        # csrc/mlp_h2.hip ST_NONFINITE, ST_MAXDEV, ST_MAXREL
        self.host = torch.empty(3, dtype=torch.int32, pin_memory=True)
        self.event = None
        self.tau = tau
        self.n_rows = None          # device int64 [2]: rows the second pass recomputed

    def fetch(self):
        self.host.copy_(self.dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def result(self, wait=True):
        """-> (non-finite flag, max |three-product - one-product| on the re-evaluated rows, the largest fraction of its sign margin
        max(tau, |sdf|) such a row's first-pass error used up) or None if not ready and wait=False"""
        if self.event is None:
            self.fetch()
        if not self.event.query():
            if not wait:
                return None
            self.event.synchronize()
        return bool(self.host[0].item()), float(self.host[1:2].view(torch.float32).item()), float(self.host[2:3].view(torch.float32).item())


def check_forward_status(net, wait=True):
    """Inspect the status of the LAST fused forward pass over `net` (h2 arithmetic only).  -> None (fine / nothing to check) or a
    string naming what the caller must do: "fp32" = the fp16-pair arithmetic overflowed (activation >= 65 504 or a weight beyond
    the fp16 range): evaluate with precision="fp32"; "one_pass" = the one-product pass was measured too inaccurate for the
    two-pass scheme at this tau.  GShellTetsGeometry.getMesh calls this right after the extraction's own stream sync (no
    extra stall) and re-runs; every other caller of fused_forward gets the check synchronously."""
    st = net.__dict__.get("_gs_fwd_status")
    if st is None:
        return None
    r = st.result(wait)
    if r is None:
        return None
    net.__dict__["_gs_fwd_status"] = None
    nonfinite, maxdev, maxrel = r
    if nonfinite:
        return "fp32"
    if st.tau is not None:
        net.__dict__["_gs_two_pass_maxdev"] = maxdev
        net.__dict__["_gs_two_pass_margin_used"] = maxrel        # over the refined rows AND the audit sample of unrefined rows
        if maxrel * SDF_TWO_PASS_SAFETY > 1.0:
            return "one_pass"
    return None


class EdgeList:
    """The static edges over which a consumer of the SDF looks for sign changes, for the two-pass forward when there is no TetTopology
    (G-FlexiCubes: the unique cube edges).  edges [E,2] int32 on the device, N = number of grid vertices."""

    def __init__(self, edges, N):
        self.edges = edges.detach().int().contiguous()
        self.N = int(N)


def fused_forward(net, x, precision=None, occ_bits_ptr=None, refine_topo=None, defer_status=False):
    """sdf = net(x) through the fused MFMA kernel (no autograd).  `occ_bits_ptr`: device address of a [ceil(N/64)] uint64
    array that receives the sign bits (h2 only; the extraction's occupancy bits).  `refine_topo` (a TetTopology
    whose vertices are the rows of x): evaluate in two passes (see SDF_TWO_PASS).  The h2 paths raise GShellHipError when the
    fp16-pair arithmetic overflowed, unless `defer_status` (then the caller must call check_forward_status)."""
    precision = precision or net.__dict__.get("_gs_precision") or SDF_MLP_PRECISION
    if precision == "torch":
        precision = "fp32"
    L = _lib.lib()
    xc = x.detach().contiguous()
    out = torch.empty((xc.shape[0],), dtype=torch.float32, device=xc.device)
    if precision == "h2":
        N = xc.shape[0]
        two_pass = refine_topo is not None and SDF_TWO_PASS and refine_topo.N == N and N > 0 and not net.__dict__.get("_gs_one_pass", False)
        st = ForwardStatus(xc.device, tau=float(net.__dict__.get("_gs_two_pass_tau", SDF_TWO_PASS_TAU)) if two_pass else None)
        packed, n_hidden, skip = pack_weights_h2(net, st.dev)
        nf = net.emb.N_freqs
        occ = _lib.c_void_p(occ_bits_ptr or 0)
        with torch.cuda.device(xc.device):
            if not two_pass:
                check(L.gs_sdf_mlp_fwd_h2(ptr(xc, torch.float32, "x"), c_int64(N), ptr(packed), c_int(nf), c_int(n_hidden), c_int(skip),
                                          ptr(out), occ, ptr(st.dev), stream()), "gs_sdf_mlp_fwd_h2")
            else:
                check(L.gs_sdf_mlp_fwd_h1(ptr(xc, torch.float32, "x"), c_int64(N), ptr(packed), c_int(nf), c_int(n_hidden), c_int(skip),
                                          ptr(out), occ, ptr(st.dev), stream()), "gs_sdf_mlp_fwd_h1")
                flags = torch.zeros(N, dtype=torch.float32, device=xc.device)
                if isinstance(refine_topo, EdgeList):
                    check(L.gs_flag_refine_rows_edges(ptr(refine_topo.edges, torch.int32, "edges"), c_int64(refine_topo.edges.shape[0]), ptr(out),
                                                      _lib.c_float(st.tau), ptr(flags), stream()), "gs_flag_refine_rows_edges")
                else:
                    check(L.gs_mtets_flag_refine_rows(refine_topo.handle, ptr(out), _lib.c_float(st.tau), ptr(flags), stream()), "gs_mtets_flag_refine_rows")
                if SDF_TWO_PASS_AUDIT_ROWS > 0:
                    # the audit sample: one residue class of the row index, rotating with the calls (plain strided fill: one launch)
                    k = max(1, N // SDF_TWO_PASS_AUDIT_ROWS)
                    phase = net.__dict__.get("_gs_audit_phase", 0)
                    net.__dict__["_gs_audit_phase"] = phase + 1
                    flags[(phase * 7919) % k::k] = 1.0
                rows = torch.empty(N, dtype=torch.int32, device=xc.device)
                st.n_rows = torch.empty(2, dtype=torch.int64, device=xc.device)
                scratch = torch.empty(int(L.gs_compact_rows_scratch_bytes(c_int64(N))) // 4 + 4, dtype=torch.int32, device=xc.device)
                check(L.gs_compact_rows(ptr(flags), c_int64(N), c_int64(N), ptr(scratch), ptr(rows), _lib.c_void_p(0), ptr(st.n_rows), stream()), "gs_compact_rows")
                check(L.gs_sdf_mlp_h2_refine_rows(ptr(xc), ptr(rows), c_int64(N), ptr(st.n_rows), ptr(packed), c_int(nf), c_int(n_hidden), c_int(skip),
                                                  ptr(out), occ, ptr(st.dev), _lib.c_float(st.tau), stream()), "gs_sdf_mlp_h2_refine_rows")
        st.fetch()
        net.__dict__["_gs_fwd_status"] = st
        if not defer_status:
            todo = check_forward_status(net)
            if todo == "fp32":
                raise _lib.GShellHipError("SDF network: a value beyond the fp16 range (activation >= 65 504 or |weight| > 60 000) overflowed the "
                                          "fp16-pair arithmetic; evaluate with precision='fp32' (gs_sdf_mlp_fwd)")
            if todo == "one_pass":
                raise _lib.GShellHipError(f"SDF network, two-pass forward: the one-product pass used up {net.__dict__.get('_gs_two_pass_margin_used'):.3f} of a row's "
                                          f"sign margin max(tau, |sdf|) (max deviation {net.__dict__.get('_gs_two_pass_maxdev')}, tau {st.tau}); more than 1 / "
                                          f"{SDF_TWO_PASS_SAFETY}: raise SDF_TWO_PASS_TAU or evaluate in one pass")
        return out[:, None]
    if precision != "fp32":
        raise ValueError(f"unknown SDF-MLP precision {precision!r} (h2 | fp32)")
    packed, n_hidden, skip = pack_weights(net)
    assert packed.numel() == L.gs_sdf_mlp_packed_floats(c_int(net.emb.N_freqs), c_int(n_hidden), c_int(skip))
    with torch.cuda.device(xc.device):
        check(L.gs_sdf_mlp_fwd(ptr(xc, torch.float32, "x"), c_int64(xc.shape[0]), ptr(packed), c_int(net.emb.N_freqs), c_int(n_hidden), c_int(skip),
                               ptr(out), stream()), "gs_sdf_mlp_fwd")
    return out[:, None]


class _ParamGate(torch.autograd.Function):
    """All parameters of the network behind ONE autograd edge.  The chain kernels write every parameter gradient of a pass into
    one flat buffer; with the 16 parameters as 16 inputs of each pass the engine sums the grid pass and the eikonal pass with
    16 `add` launches and zero-fills 16 buffers per pass.  The gate's output is an uninitialised flat TOKEN (its values are
    never read: the kernels take the parameters' own device pointers); a pass returns its flat gradient for the token, the
    engine adds the passes with one launch, and the gate hands each parameter its view of the sum."""

    @staticmethod
    def forward(ctx, *params):
        ctx.shapes = [p.shape for p in params]
        return torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)

    @staticmethod
    def backward(ctx, g):
        out, off = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp)) if len(shp) else 1
            out.append(g[off:off + n].view(shp))
            off += n
        return tuple(out)


def param_gate(net, reuse=False):
    """a token for this forward pass; `reuse` = take the one the grid pass of the same iteration left on the network"""
    if reuse:
        g = net.__dict__.pop("_gs_gate", None)
        if g is not None and (g.requires_grad or not torch.is_grad_enabled()):      # never a token made under no_grad for a pass that needs gradients
            return g
    g = _ParamGate.apply(*list(net.parameters()))
    if not reuse and g.requires_grad:
        net.__dict__["_gs_gate"] = g
    return g


def split_param_grads(net, flat):
    """flat gradient (parameter order) -> list of views shaped like net.parameters()"""
    out, off = [], 0
    for p in net.parameters():
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return out


def _flat_grads(params, grads):
    """list of per-parameter gradients -> one flat tensor (the torch fallback paths)"""
    return torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1).float() for p, g in zip(params, grads)])


class _RowSparseBackward(torch.autograd.Function):
    """y = net(x) over ALL rows, backward only over the rows whose upstream gradient is non-zero.

    The SDF of every grid vertex is needed in the forward pass (its sign decides the topology), but d loss / d sdf is
    non-zero only at end points of sign-crossing edges (extraction backward + the sign regulariser touch nothing else):
    ~10 % of the rows at tet-res 256.  Rows with zero upstream gradient contribute exactly zero to every weight and input
    gradient, so restricting the backward GEMMs to the active rows is exact -- and the forward pass then needs to keep NO
    activations (the reference's autograd saves 14 x [N,256] fp32 = 30 GB at res 256): the active rows are
    recomputed in the backward pass."""

    @staticmethod
    def forward(ctx, x, net, sign_sink, gate):
        with torch.no_grad():
            presign = sign_sink is not None and SDF_MLP_PRECISION == "h2" and _fusable(net, x) and sign_sink.N == x.shape[0]
            bits = sign_sink.occ_bits_ptr() if (presign and not isinstance(sign_sink, EdgeList)) else None      # an EdgeList only selects the refined rows
            y = (fused_forward(net, x, occ_bits_ptr=bits, refine_topo=sign_sink if presign else None,
                               defer_status=presign) if _fusable(net, x, "forward") else net(x))
        ctx.net = net
        ctx.presigned = presign
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (x,) = ctx.saved_tensors
        g_x, g_flat = row_sparse_backward(ctx.net, x, g_y, ctx.needs_input_grad[0])
        return g_x, None, None, g_flat


def row_sparse_backward_torch(net, x, g_y, need_x):
    """Plain-torch formulation (autograd over the recomputed active rows): CPU path of the gloo tests, and the oracle of the
    HIP chain kernels in tests/."""
    params = [p for p in net.parameters()]
    rows = torch.nonzero(g_y.reshape(x.shape[0], -1).abs().sum(dim=1) != 0).reshape(-1)      # one host sync
    g_x = torch.zeros_like(x) if need_x else None
    if rows.numel() == 0:
        return g_x, _flat_grads(params, [None] * len(params))
    x_a = x[rows].detach().requires_grad_(need_x)
    with torch.enable_grad():
        y_a = net(x_a)
        grads = torch.autograd.grad(y_a, ([x_a] if need_x else []) + params, g_y[rows], allow_unused=True)
    if need_x:
        g_x[rows] = grads[0]
        grads = grads[1:]
    return g_x, _flat_grads(params, grads)


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


LAST_CHAIN_ROWS = {}      # mode (1 = grid rows with gradient, 2 = eikonal samples) -> rows of the last saved pass (bench.py: flops per launch)


class _SavedChain:
    """Saved planes of one pass of the h2 chain kernels over n (virtual) rows: A [layers, Rpad, 256], EMB [Rpad, 48]."""

    def __init__(self, net, mode, x, rows, n, n_dev=None):
        """n = number of (virtual) rows, or their CAPACITY when `n_dev` (device int64 tensor, [0] = the count) is given."""
        L = _lib.lib()
        self.net, self.mode, self.n, self.n_dev = net, mode, int(n), n_dev
        LAST_CHAIN_ROWS[mode] = int(n)
        self.lin, self.n_hidden, self.skip = _layer_structure(net)
        self.nf = net.emb.N_freqs
        self.packed, _, _ = pack_weights_h2(net)
        dev = x.device
        self.Rpad = int(L.gs_sdf_mlp_h2_rows_padded(c_int(mode), c_int64(self.n)))
        nl = self.n_hidden + 1
        self.A = torch.empty((nl, self.Rpad, 256), dtype=torch.float32, device=dev)
        self.EMB = torch.empty((self.Rpad, 48), dtype=torch.float32, device=dev)
        self.out = torch.empty((self.Rpad,), dtype=torch.float32, device=dev) if mode == 2 else None
        self.rows = rows
        with torch.cuda.device(dev):
            check(L.gs_sdf_mlp_h2_save_fwd(c_int(mode), ptr(x, torch.float32, "x"), ptr(rows, torch.int32, "rows"), c_int64(self.n), ptr(self.n_dev), ptr(self.packed),
                                           c_int(self.nf), c_int(self.n_hidden), c_int(self.skip), ptr(self.A), ptr(self.EMB), ptr(self.out), stream()),
                  "gs_sdf_mlp_h2_save_fwd")

    def backward(self, g_out, g_x=None):
        """g_out [Rpad] per virtual row -> the gradients of net.parameters() as ONE flat tensor (parameter order); g_x [N,3] is
        filled in place (mode 1)."""
        L = _lib.lib()
        dev = g_out.device
        D = torch.empty_like(self.A)
        params = list(self.net.parameters())
        # one zero-filled buffer, one view per parameter (16 fills -> 1)
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        grads, off = {}, 0
        for p in params:
            grads[id(p)] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        dW = [grads[id(m.weight)] for m in self.lin]
        db = [grads[id(m.bias)] for m in self.lin]
        with torch.cuda.device(dev):
            check(L.gs_sdf_mlp_h2_bwd(c_int(self.mode), ptr(g_out, torch.float32, "g_out"), ptr(self.rows, torch.int32, "rows"), c_int64(self.n), ptr(self.n_dev),
                                      ptr(self.packed),
                                      c_int(self.nf), c_int(self.n_hidden), c_int(self.skip), ptr(self.A), ptr(self.EMB), ptr(D), ptr(g_x), stream()),
                  "gs_sdf_mlp_h2_bwd")
            check(L.gs_sdf_mlp_h2_wgrad(c_int(self.mode), ptr(g_out), c_int64(self.n), ptr(self.n_dev), c_int(self.nf), c_int(self.n_hidden), c_int(self.skip),
                                        ptr(self.A),
                                        ptr(self.EMB), ptr(D), _ptr_array(dW), _ptr_array(db), c_int(1 if SDF_MLP_WGRAD_FP32 else 0), stream()),
                  "gs_sdf_mlp_h2_wgrad")
        if self.mode == 1:      # output bias: sum of the upstream gradient over the (value) rows
            db[-1].copy_(g_out.sum().reshape(db[-1].shape))
        return flat


def row_sparse_backward(net, x, g_y, need_x):
    """(d loss / d x [N,3] or None, flat gradient of net.parameters()) from the upstream gradient g_y [N,1],
    touching only the rows where g_y != 0: they are recomputed by the h2 chain kernels (csrc/mlp_h2.hip: forward with saved
    planes -> backward chain -> weight gradients on the matrix cores); the full-grid forward pass keeps no activations."""
    if not (_fusable(net, x, "row-sparse backward") and g_y.is_cuda):
        return row_sparse_backward_torch(net, x, g_y, need_x)
    g = g_y.detach().reshape(-1).contiguous().float()
    g_x = torch.zeros_like(x) if need_x else None
    bound = getattr(net, "_gs_rows_bound", None)
    if bound is not None:
        # no host sync: the rows with gradient are compacted on the device and the planes are sized by the caller's bound
        # (GShellTetsGeometry.getMesh: 2 x crossing edges -- d loss / d sdf is non-zero only at their end points)
        _check_rows_overflow(net)
        L = _lib.lib()
        N = g.numel()
        cap = max(int(min(N, bound)), 1)
        rows = torch.empty(cap, dtype=torch.int32, device=x.device)
        count = torch.empty(2, dtype=torch.int64, device=x.device)
        scratch = torch.empty(int(L.gs_compact_rows_scratch_bytes(c_int64(N))) // 4 + 4, dtype=torch.int32, device=x.device)
        g_rows = torch.zeros((int(L.gs_sdf_mlp_h2_rows_padded(c_int(1), c_int64(cap))),), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(L.gs_compact_rows(ptr(g), c_int64(N), c_int64(cap), ptr(scratch), ptr(rows), ptr(g_rows), ptr(count), stream()), "gs_compact_rows")
        _watch_rows_overflow(net, count)
        saved = _SavedChain(net, 1, x.detach().contiguous(), rows, cap, n_dev=count)
        return g_x, saved.backward(g_rows, g_x)
    # rows with gradient: compacted by the library's own three launches (ascending indices + their values), ONE host sync for the count that
    # sizes the saved planes (torch.nonzero + gather + scatter were ~11 launches)
    L = _lib.lib()
    N = g.numel()
    rows_all = torch.empty(N, dtype=torch.int32, device=x.device)
    g_all = torch.zeros(N + 256, dtype=torch.float32, device=x.device)          # zero tail: the chain kernels read whole 128-row tiles
    count = torch.empty(2, dtype=torch.int64, device=x.device)
    scratch = torch.empty(int(L.gs_compact_rows_scratch_bytes(c_int64(N))) // 4 + 4, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        check(L.gs_compact_rows(ptr(g), c_int64(N), c_int64(N), ptr(scratch), ptr(rows_all), ptr(g_all), ptr(count), stream()), "gs_compact_rows")
    n = int(count[0])                                               # the host sync
    if n == 0:
        return g_x, torch.zeros(sum(p.numel() for p in net.parameters()), dtype=torch.float32, device=x.device)
    saved = _SavedChain(net, 1, x.detach().contiguous(), rows_all[:n], n)
    assert saved.Rpad <= g_all.numel()
    return g_x, saved.backward(g_all[:saved.Rpad], g_x)


def _watch_rows_overflow(net, count):
    """The bound on the rows with gradient is checked OFF the hot path: the (count, overflow) pair is copied to pinned host memory
    asynchronously and inspected at the next call, by which time the copy has long completed (no stall)."""
    host = getattr(net, "_gs_rows_host", None)
    if host is None:                     # pinned once: a hipHostMalloc per step costs more than the sync it replaces
        host = net._gs_rows_host = torch.empty(2, dtype=torch.int64, pin_memory=True)
    if getattr(net, "_gs_rows_watch", None) is not None:
        return                           # the previous copy has not been inspected yet: keep watching that one
    host.copy_(count, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    net._gs_rows_watch = (host, ev)


def _check_rows_overflow(net):
    w = getattr(net, "_gs_rows_watch", None)
    if w is not None and w[1].query():
        net._gs_rows_watch = None
        if int(w[0][1]) != 0:
            raise _lib.GShellHipError(f"row-sparse SDF backward: more than the {getattr(net, '_gs_rows_bound', '?')} rows promised by net._gs_rows_bound carried "
                                      "gradient in the previous step (a loss touches sdf values off the crossing edges?) -- unset the bound")


class _EikonalFn(torch.autograd.Function):
    """sum_i (|grad_x f(x_i)| - 1)^2 with gradients w.r.t. the network parameters (reference geometry/gshell_tets_geometry.py:
    302-324: autograd.grad(..., create_graph=True) + a second backward).  Here: forward-mode tangents through the chain kernel
    (4 virtual rows per sample), then ONE reverse pass over those rows -- no double backward, no hipBLASLt."""

    @staticmethod
    def forward(ctx, pts, net, gate):
        n = pts.shape[0]
        saved = _SavedChain(net, 2, pts.detach().contiguous().float(), None, n)
        loss = torch.empty(1, dtype=torch.float32, device=pts.device)
        g_unit = torch.empty_like(saved.out)
        with torch.cuda.device(pts.device):
            check(_lib.lib().gs_sdf_eikonal_loss(ptr(saved.out), c_int64(n), c_int64(saved.Rpad), ptr(loss), ptr(g_unit), stream()), "gs_sdf_eikonal_loss")
        ctx.saved, ctx.g_unit = saved, g_unit
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        saved, g_unit = ctx.saved, ctx.g_unit
        g_flat = saved.backward(g_unit * g)            # virtual-row order 64 t + 16 c + i
        ctx.saved = ctx.g_unit = None
        return None, None, g_flat


# "reverse": reverse over reverse, the reference's own formulation (value pass, reverse chain, tangent pass, reverse chain with the second-order
# source: 6 row passes per sample); "tangent": forward-mode tangent rows (4 virtual rows per sample through 3 passes: 12).  Same gradient.
EIKONAL_FORMULATION = "reverse"


class _EikonalRRFn(torch.autograd.Function):
    """sum_i (|grad_x f(x_i)| - 1)^2 by reverse over reverse (csrc/mlp_h2.hip MODE_RR): forward = gs_sdf_eikonal_rr_fwd (value pass + reverse
    chain + loss), backward = gs_sdf_eikonal_rr_bwd (tangent pass + reverse chain with source + ONE weight-gradient launch over 2 n rows)."""

    @staticmethod
    def forward(ctx, pts, net, gate):
        L = _lib.lib()
        n = pts.shape[0]
        x = pts.detach().contiguous().float()
        dev = x.device
        _, n_hidden, skip = _layer_structure(net)
        packed, _, _ = pack_weights_h2(net)
        Rpad = int(L.gs_sdf_eikonal_rr_rows_padded(c_int64(n)))
        LAST_CHAIN_ROWS[4] = n
        nl = n_hidden + 1
        # two allocations (the planes; everything else): the call sits right behind the extraction's sync, where host time is GPU idle time
        planes = torch.empty((2, nl, 2 * Rpad, 256), dtype=torch.float32, device=dev)
        A, Dp = planes[0], planes[1]
        rest = torch.empty(2 * Rpad * 48 + 2 * Rpad + 6 * n + 4, dtype=torch.float32, device=dev)
        o = 0
        EMB = rest[o:o + 2 * Rpad * 48].view(2 * Rpad, 48); o += 2 * Rpad * 48
        g_all = rest[o:o + 2 * Rpad]; o += 2 * Rpad
        grad_f = rest[o:o + 3 * n].view(n, 3); o += 3 * n
        gbar = rest[o:o + 3 * n].view(n, 3); o += 3 * n
        loss = rest[o:o + 1]
        with torch.cuda.device(dev):
            check(L.gs_sdf_eikonal_rr_fwd(ptr(x, torch.float32, "pts"), c_int64(n), ptr(packed), c_int(net.emb.N_freqs), c_int(n_hidden), c_int(skip), ptr(A), ptr(EMB),
                                          ptr(Dp), ptr(g_all), ptr(grad_f), ptr(gbar), ptr(loss), stream()), "gs_sdf_eikonal_rr_fwd")
        ctx.state = (net, n, packed, n_hidden, skip, A, EMB, Dp, g_all, gbar)
        ctx.grad_f = grad_f          # grad_x f of the samples (tests)
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        net, n, packed, n_hidden, skip, A, EMB, Dp, g_all, gbar = ctx.state
        ctx.state = None
        L = _lib.lib()
        dev = A.device
        lin, _, _ = _layer_structure(net)
        params = list(net.parameters())
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        grads = dict(zip((id(p) for p in params), split_param_grads(net, flat)))
        dW = [grads[id(m.weight)] for m in lin]
        db = [grads[id(m.bias)] for m in lin]
        g_up = g.detach().reshape(1).float().contiguous()
        with torch.cuda.device(dev):
            check(L.gs_sdf_eikonal_rr_bwd(c_int64(n), ptr(packed), c_int(net.emb.N_freqs), c_int(n_hidden), c_int(skip), ptr(A), ptr(EMB), ptr(Dp), ptr(g_all),
                                          ptr(gbar), ptr(g_up), _ptr_array(dW), _ptr_array(db), c_int(1 if SDF_MLP_WGRAD_FP32 else 0), stream()),
                  "gs_sdf_eikonal_rr_bwd")
        return None, None, flat


def eikonal_sq_sum(net, pts):
    """sum_i (|d net / d x (pts_i)| - 1)^2, differentiable w.r.t. the parameters of `net` (the points are constants, as in the
    reference, which detaches them)."""
    if _fusable(net, pts, "eikonal term"):
        fn = _EikonalRRFn if EIKONAL_FORMULATION == "reverse" else _EikonalFn
        return fn.apply(pts, net, param_gate(net, reuse=True))
    v = pts.detach().requires_grad_(True)
    grad = torch.autograd.grad(net(v).sum(), v, create_graph=True)[0]
    return (grad.pow(2).sum(dim=-1).sqrt() - 1).pow(2).sum()


class _RowShardedForward(torch.autograd.Function):
    """View-sharded jobs ("shard MLP rows across ranks + all-gather sdf [N]"): rank r evaluates the grid rows
    [r * per, (r + 1) * per) through the fused kernel and all-gathers the N signed distances (8.8 MB at tet-res 256) -- the
    16 ms full-grid forward stops being replicated.  Backward: the upstream gradients of all ranks are reduce-scattered
    (sum) onto the owning rank, which runs the row-sparse backward on ITS rows; the parameter / input gradients are partial
    sums that the trainer's flat all-reduce completes.  Every rank sees bit-identical sdf values (gathered, not recomputed),
    so the replicated extraction still yields identical meshes."""

    @staticmethod
    def forward(ctx, x, net, shard, gate):
        N, world, rank = x.shape[0], shard.world, shard.rank
        per = (N + world - 1) // world
        lo, hi = min(rank * per, N), min((rank + 1) * per, N)
        with torch.no_grad():
            x_loc = x[lo:hi].contiguous()
            y_loc = (fused_forward(net, x_loc) if _fusable(net, x_loc, "row-sharded forward") else net(x_loc)).reshape(-1)
            pad = torch.zeros(per, dtype=y_loc.dtype, device=y_loc.device)
            pad[:hi - lo] = y_loc
            y_full = shard.all_gather_rows(pad)
        ctx.net, ctx.shard, ctx.range = net, shard, (lo, hi, per)
        ctx.save_for_backward(x)
        return y_full[:N, None].contiguous()

    @staticmethod
    def backward(ctx, g_y):
        (x,) = ctx.saved_tensors
        lo, hi, per = ctx.range
        N, world = x.shape[0], ctx.shard.world
        g_pad = torch.zeros(per * world, dtype=g_y.dtype, device=g_y.device)
        g_pad[:N] = g_y.reshape(-1)
        g_loc = ctx.shard.reduce_scatter_sum(g_pad)[:hi - lo]
        need_x = ctx.needs_input_grad[0]
        g_x_loc, g_flat = row_sparse_backward(ctx.net, x[lo:hi].contiguous(), g_loc[:, None], need_x)
        g_x = None
        if need_x:
            g_x = torch.zeros_like(x)
            g_x[lo:hi] = g_x_loc
        return g_x, None, None, g_flat


def forward_row_sharded(net, x, shard):
    """net(x) with rows sharded over the ranks of `shard` (see _RowShardedForward)."""
    return _RowShardedForward.apply(x, net, shard, param_gate(net))


def forward_row_sparse_backward(net, x, sign_sink=None):
    """net(x) with the row-sparse backward described above (first-order gradients only).  `sign_sink` (a TetTopology): the
    kernel's epilogue also writes the sign bits of the result straight into the extractor's occupancy array; the returned
    tensor is tagged (`_gs_presigned`) so that GShell_Tets skips its own sign pass.  `sign_sink` (an EdgeList): two-pass evaluation
    over that edge set, no sign bits.  Either way the caller must call check_forward_status(net) after its next host sync."""
    y = _RowSparseBackward.apply(x, net, sign_sink, param_gate(net))
    if sign_sink is not None and not isinstance(sign_sink, EdgeList) and SDF_MLP_PRECISION == "h2" and _fusable(net, x) and sign_sink.N == x.shape[0]:
        sign_sink.sign_epoch += 1
        y._gs_presigned = (sign_sink, sign_sink.sign_epoch, y._version)
    return y
