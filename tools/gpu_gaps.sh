#!/bin/bash
# Where the GPU idles inside a training iteration: rocprofv3 kernel trace of the default bench -> gaps between consecutive kernels, attributed
# to the kernel before and after each.  GPU box.  usage: tools/gpu_gaps.sh [bench args]
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=/tmp/gaps; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python "$root/bench.py" --no-cpu-baseline --early-steps 0 --extra-steps 0 --steps 1 --warmup 0 --state-file $out/state.pt "$@" > $out/setup.log 2>&1 </dev/null
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python "$root/bench.py" --no-cpu-baseline --early-steps 0 --extra-steps 0 --steps 12 --warmup 4 --state-file $out/state.pt "$@" > $out/run.log 2>&1 </dev/null
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# the timed region: from the LAST 12 launches of k_h1_fwd on
starts = [i for i, r in enumerate(rows) if "k_h1_fwd" in r[2]]
i0 = starts[-12]
rows = rows[i0:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
its = 12
print(f"span {span / its / 1e6:.3f} ms / iteration, kernels busy {busy / its / 1e6:.3f} ms, idle {(span - busy) / its / 1e6:.3f} ms")
import re
short = lambda s: (re.findall(r"k_[a-z0-9_]+(?:<[^>]*>)?", s) or [re.sub(r"^void ", "", s).split("<")[0][-40:]])[0]
gaps = collections.Counter(); cnt = collections.Counter()
end = rows[0][1]
for (s, e, n), (ps, pe, pn) in zip(rows[1:], rows[:-1]):
    g = s - max(end, pe)
    end = max(end, e)
    if g > 0:
        key = (short(pn), short(n))
        gaps[key] += g; cnt[key] += 1
for (a, b), g in gaps.most_common(25):
    print(f"{g / its / 1e3:8.1f} us/it  {cnt[(a, b)] / its:5.1f} x   {a}  ->  {b}")
PY
