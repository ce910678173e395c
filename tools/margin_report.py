"""Largest measured / bar ratio per (chain, tensor) over the runs tools/margin_loop.sh kept (gpurun_out/margins/run_*.txt).  CPU.
usage: python tools/margin_report.py [dir]"""
import glob
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "margins")
pat = re.compile(r"gradient (\S+): relative L2 vs float64 ([0-9.e+-]+); float32 oracle ([0-9.e+-]+); bar ([0-9.e+-]+)")
worst, fails, n = {}, 0, 0
for f in sorted(glob.glob(os.path.join(d, "run_*.txt"))):
    n += 1
    chain = "?"
    txt = open(f).read()
    fails += int(re.search(r"rc=[1-9]", txt) is not None)
    for line in txt.splitlines():
        m = re.search(r"chain (\S+): SDF network", line)
        if m:
            chain = m.group(1)
        m = pat.search(line)
        if m:
            name, err, e32, bar = m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4))
            k = (chain, name)
            r = err / bar if bar > 0 else float("inf")
            if k not in worst or r > worst[k][0]:
                worst[k] = (r, err, bar, e32)
print(f"{n} runs, {fails} with a failing test")
for (chain, name), (r, err, bar, e32) in sorted(worst.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {r:5.2f} of its bar: {chain:16s} {name:28s} worst relative L2 {err:.2e}, bar {bar:.2e}, float32 oracle {e32:.2e}")
