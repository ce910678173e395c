"""torch.optim.Adam with its step as ONE HIP launch over all parameter tensors (csrc/adam.hip: gs_adam_step).

Same constructor, parameter groups, schedulers and state_dict as torch.optim.Adam(fused=True) -- the state is the fused optimiser's
(`step` a float32 device scalar per parameter, `exp_avg`, `exp_avg_sq`), so checkpoints move freely between the two -- and the same
arithmetic (ATen/native/cuda/fused_adam_utils.cuh).  The reference trains with torch.optim.Adam (train_gshelltet_deepfashion.py:372-383)."""
import ctypes

import torch

from . import _lib
from ._lib import c_int, check, stream

CONTRACT = 1          # rounding of the moment updates that reproduces ATen's fused kernel bit for bit on gfx950 (tests/test_adam_gpu.py)


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **kw):
        for k, bad in (("weight_decay", 0), ("amsgrad", False), ("maximize", False), ("capturable", False), ("differentiable", False)):
            if kw.pop(k, bad) != bad:
                raise NotImplementedError(f"HipAdam: {k} is not used by the reference's optimisers and is not implemented")
        kw.pop("fused", None)
        kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_key = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise _lib.GShellHipError("HipAdam: dense fp32 parameters in HBM only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["_host_step"] = None
                if st.get("_host_step") is None:          # fresh state or one loaded from a checkpoint: read the counter once
                    st["_host_step"] = float(st["step"])
                    # a checkpoint written by torch's default (non-fused) Adam keeps `step` on the CPU: the kernel writes the counter through
                    # this pointer, so it has to be a float32 scalar on the parameter's device
                    if st["step"].device != p.device or st["step"].dtype != torch.float32:
                        st["step"] = st["step"].detach().to(device=p.device, dtype=torch.float32)
                    for name in ("exp_avg", "exp_avg_sq"):
                        if st[name].device != p.device or st[name].dtype != torch.float32:
                            st[name] = st[name].detach().to(device=p.device, dtype=torch.float32).contiguous()
                st["_host_step"] += 1.0
                if not (p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    raise _lib.GShellHipError("HipAdam: parameters and their moments must be contiguous")
                key = (p.device, tuple(group["betas"]), group["eps"], st["_host_step"])          # a loaded state may carry betas as a list
                by_key.setdefault(key, []).append((p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st, float(group["lr"])))
        L = _lib.lib()
        for (dev, betas, eps, step_value), items in by_key.items():
            n = len(items)
            vp = ctypes.c_void_p * n
            keep = [g for _, g, _, _ in items]          # contiguous copies must outlive the launch call
            with torch.cuda.device(dev):
                check(L.gs_adam_step(c_int(n), vp(*[p.data_ptr() for p, _, _, _ in items]), vp(*[g.data_ptr() for g in keep]),
                                     vp(*[st["exp_avg"].data_ptr() for _, _, st, _ in items]), vp(*[st["exp_avg_sq"].data_ptr() for _, _, st, _ in items]),
                                     vp(*[st["step"].data_ptr() for _, _, st, _ in items]), (ctypes.c_int64 * n)(*[p.numel() for p, _, _, _ in items]),
                                     (ctypes.c_double * n)(*[lr for _, _, _, lr in items]), ctypes.c_double(betas[0]), ctypes.c_double(betas[1]),
                                     ctypes.c_double(eps), ctypes.c_double(step_value), c_int(CONTRACT), stream()), "gs_adam_step")
            # the kernel writes the parameters through raw pointers: tell autograd (saved-tensor checks, and every cache keyed on a parameter's
            # version such as the packed SDF-network weights, see them as modified in place)
            torch.autograd.graph.increment_version([p for p, _, _, _ in items])
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd["state"] = {k: {n: v for n, v in st.items() if n != "_host_step"} for k, st in sd["state"].items()}      # torch's keys only
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            st["_host_step"] = None
