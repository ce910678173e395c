#!/bin/bash
# average duration of the kernels whose name matches <regex> over tools/shade_time.py's iterations (rocprofv3 --kernel-trace --stats).  GPU box.
# usage: [GSHELL_HIP_LIB=...] tools/kernel_times.sh <regex>
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=$(mktemp -d /tmp/kt.XXXX)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python "$root/tools/${GS_KT_SCRIPT:-shade_time.py}" ${GS_KT_ARGS:-} > "$out/run.log" 2>&1 </dev/null
python - "$out" "$1" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"])
        print(f'  {m.group(1) if m else r["Name"][:50]:34s} x{r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
rm -rf "$out"
