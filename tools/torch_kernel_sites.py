"""Which Python lines of this package issue the torch (non-gs_*) device kernels of a training iteration,
ranked by device time.  Run on the GPU box:  python tools/torch_kernel_sites.py > gpurun_out/sites.txt"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from gshell_amd import workload

ITERS = 3
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
for _ in range(3):
    tr.step(tg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(ITERS):
        tr.step(tg)
    torch.cuda.synchronize()

by_site = collections.defaultdict(lambda: [0.0, 0])
by_site_bwd = collections.defaultdict(lambda: [0.0, 0])
total = 0.0
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", 0.0) or 0.0
    if dt <= 0 or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    site = None
    for fr in ev.stack:
        if "gshell_amd" in fr and "_lib.py" not in fr:
            site = fr.split("gshell_amd/")[-1]
            break
    key = site or ("<autograd> " + ev.name)
    tgt = by_site if site else by_site_bwd
    tgt[key][0] += dt
    tgt[key][1] += 1
    total += dt
print(f"torch-issued device time per iteration: {total / ITERS / 1e3:.2f} ms")
for title, d in (("forward sites", by_site), ("autograd-engine ops (no python frame)", by_site_bwd)):
    print("\n== " + title)
    for k, (t, n) in sorted(d.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"{t / ITERS / 1e3:8.3f} ms {n // ITERS:5d} ops  {k}")
