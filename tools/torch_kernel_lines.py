"""Which SOURCE LINES issue the torch / rocPRIM / memcpy kernels of a training iteration (everything that is not a hand-written k_*
kernel): forward operators by the innermost gshell_amd frame of their python stack, backward operators by the stack of the forward
operator they differentiate (autograd sequence number).  GPU box:  python tools/torch_kernel_lines.py [tets|flexicubes] [res]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from gshell_amd import workload

ITERS = 3
GEOM = sys.argv[1] if len(sys.argv) > 1 else "tets"
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tr = workload.build(res=RES, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200, geometry=GEOM)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
tr.it = 1000
for _ in range(3):
    tr.step(tg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(ITERS):
        tr.step(tg)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]


def site(e):
    p = e
    while p is not None:
        for fr in (p.stack or []):
            if "gshell_amd" in fr and "_lib.py" not in fr:
                return fr.split("gshell_amd/")[-1]
        p = p.cpu_parent
    return None


seq_site = {}
for e in evs:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and not e.name.startswith("autograd::engine"):
        s = site(e)
        if s and e.sequence_nr not in seq_site:
            seq_site[e.sequence_nr] = s + "  [" + e.name + "]"


def bwd_site(e):
    p = e
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function"):
            return "bwd of " + seq_site.get(p.sequence_nr, "<?> ") + " :: " + p.name.split(": ")[-1]
        p = p.cpu_parent
    return None


agg = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
for e in evs:
    for k in getattr(e, "kernels", []):
        if "anonymous namespace)::k_" in k.name:
            continue
        s = bwd_site(e) or site(e) or "<no gshell_amd frame> " + e.name
        a = agg[s]
        a[0] += k.duration
        a[1] += 1
        a[2][e.name + " -> " + k.name.split("<")[0].split("::")[-1][:40]] += 1
tot = sum(a[0] for a in agg.values())
print(f"torch-issued kernels: {tot / ITERS / 1e3:.3f} ms per iteration in {sum(a[1] for a in agg.values()) / ITERS:.0f} launches")
for s, (t, n, ops) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / ITERS / 1e3:7.3f} ms {n / ITERS:5.1f} x  {s}")
    for o, c in ops.most_common(4):
        print(f"                     {c / ITERS:5.1f} x {o}")
