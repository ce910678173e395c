"""Which source lines issue ATen operators in one training iteration: a TorchDispatchMode logs every operator that reaches the dispatcher
together with the innermost gshell_amd frame of the python stack (forward), or -- for operators issued by the autograd engine -- the
custom Function / torch node whose backward is running (engine threads have no python stack of their own; custom Functions' backward
bodies do).  Operators that launch nothing (views, empty, as_strided ...) are listed separately.  GPU box:
    python tools/dispatch_sites.py [tets|flexicubes] [res]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from gshell_amd import workload

GEOM = sys.argv[1] if len(sys.argv) > 1 else "tets"
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NO_LAUNCH = ("view", "empty", "as_strided", "reshape", "expand", "select", "slice", "unsqueeze", "squeeze", "permute", "transpose", "detach", "alias", "t.default",
             "_unsafe_view", "unbind", "split", "is_", "size", "stride", "numel", "storage_offset", "_local_scalar_dense", "lift_fresh", "record_stream", "set_",
             "_to_copy" if False else "\0", "narrow", "unfold", "result_type", "_reshape_alias", "resize_", "sym_", "dim", "item")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.ops = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(p) for p in NO_LAUNCH):
            site = "<no gshell_amd frame>"
            for fr in reversed(traceback.extract_stack(limit=40)):
                fn = fr.filename
                if "gshell_amd/" in fn and not fn.endswith("_lib.py"):
                    site = f"{fn.split('gshell_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            self.sites[site] += 1
            self.ops[site][name] += 1
        return func(*args, **(kwargs or {}))


tr = workload.build(res=RES, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200, geometry=GEOM)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
tr.it = 1000
for _ in range(3):
    tr.step(tg)
torch.cuda.synchronize()
log = Log()
with log:
    tr.step(tg)
torch.cuda.synchronize()
print(f"{sum(log.sites.values())} dispatched operators that may launch, by site:")
for site, n in log.sites.most_common():
    print(f"{n:4d}  {site}")
    print("        " + ", ".join(f"{k} x{v}" for k, v in log.ops[site].most_common()))
