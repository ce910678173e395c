"""Lists the host synchronisation points of one training iteration (torch.cuda.set_sync_debug_mode('warn')).  GPU box."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import workload

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tr = workload.build(res=res, n_samples=4, batch=2, train_res=(128, 128), fit_steps=100)
tg = workload.make_targets(tr, [0, 1], (128, 128))
tr.it = 1000
for _ in range(2):
    tr.step(tg)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    tr.step(tg)
torch.cuda.set_sync_debug_mode("default")
import traceback
print(f"{len(w)} torch-visible synchronisations in one iteration (C-ABI calls that sync internally -- gs_mtets_count -- are not seen by torch):")
for x in w:
    print("  ", str(x.message)[:100], "@", x.filename.split("/")[-1], x.lineno)
