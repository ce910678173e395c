"""Host-side profile of the training iteration (cProfile) + per-phase wall times with device syncs."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import workload

tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=400)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
for _ in range(3):
    tr.step(tg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(tg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
pr.disable()
print(f"ms/step {dt * 1e3:.1f}")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
