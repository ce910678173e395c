"""The autograd graph of one training iteration: every node type with the source line (innermost gshell_amd frame) that created it and the
shape it differentiates -- to find where the torch-issued backward launches (SliceBackward zero-fill + copy, gradient accumulation adds,
cat / split copies) come from.  Forward stacks via anomaly mode.  GPU box:  python tools/autograd_sites.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import workload

tr = workload.build(res=int(sys.argv[1]) if len(sys.argv) > 1 else 256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=50)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
tr.it = 1000
tr.step(tg)
with torch.autograd.detect_anomaly(check_nan=False):
    tr.lgt.update_pdf()
    tr.FLAGS.noise_stream.set_iteration(tr.it, None)
    img_loss, depth_loss, reg_loss = tr.geometry.tick(tr.glctx, tg, tr.lgt, tr.mat, tr.loss_fn, tr.it, denoiser=tr.denoiser)
    total = img_loss + reg_loss


def site(node):
    tb = node.metadata.get("traceback_", None)
    if not tb:
        return "?"
    best = "?"
    for line in tb:
        if "gshell_amd/" in line and "_lib.py" not in line:
            head = line.strip().split("\n")
            loc = head[0].split("gshell_amd/")[-1].replace('", line ', ":").replace(", in ", " ")
            best = loc + "  | " + (head[1].strip()[:110] if len(head) > 1 else "")
    return best


seen, stack = set(), [total.grad_fn]
count = collections.Counter()
users = collections.Counter()
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    count[(type(n).__name__, site(n))] += 1
    for nxt, _ in n.next_functions:
        if nxt is not None:
            users[nxt] += 1          # > 1 users: the engine sums the incoming gradients with (users - 1) add launches
            stack.append(nxt)
print(f"{len(seen)} nodes")
print("\n== nodes whose output gradient is the sum of several users (each extra user = one add launch in backward):")
for n, u in sorted(users.items(), key=lambda kv: -kv[1]):
    if u > 1 and type(n).__name__ != "AccumulateGrad":
        print(f"  {u} users  {type(n).__name__:32s} {site(n)}")
print("\n== leaves with several users:")
for n, u in sorted(users.items(), key=lambda kv: -kv[1]):
    if u > 1 and type(n).__name__ == "AccumulateGrad":
        print(f"  {u} users  leaf {tuple(n.variable.shape)}")
print("\n== all nodes by type and site:")
for (t, s), c in sorted(count.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    if not t.startswith("_") and t not in ("AccumulateGrad",):
        print(f"  {c:3d} x {t:28s} {s}")
