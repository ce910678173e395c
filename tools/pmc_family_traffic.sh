#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) of a kernel FAMILY of the training iteration: for every kernel whose
# name contains one of the given substrings, the counters of its LAST dispatch; the family total is their sum.  GPU box.
# --extra-steps 0: without it the LAST dispatch of every kernel belongs to bench.py's close-camera side run (41 % coverage), not to the
# headline camera the algorithmic bytes are quoted for.
#   usage: tools/pmc_family_traffic.sh <tag> <substr1> [<substr2> ...]       -> gpurun_out/pmc_<tag>/traffic.json
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_$tag"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python "$root/bench.py" --no-cpu-baseline --early-steps 0 --extra-steps 0 --steps 1 --warmup 0 --state-file "$out/state.pt" > "$out/setup.log" 2>&1 </dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d "$out/$c" -o r --output-format csv -- python "$root/bench.py" --no-cpu-baseline --early-steps 0 --extra-steps 0 --steps 2 --warmup 1 --state-file "$out/state.pt" > "$out/$c.log" 2>&1 </dev/null
  f=$(find "$out/$c" -name "*counter_collection.csv" | head -1)
  python - "$f" "$out/$c.json" "$@" <<'PY'
import csv, sys, json, re
f, o, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
last = {}
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if any(s in name for s in subs):
        m = re.search(r"(k_\w+)(<[^>]*>)?", name)
        key = m.group(1) + (m.group(2) or "")
        d = int(r["Dispatch_Id"])
        if key not in last or d > last[key][0]:
            last[key] = (d, 0.0)
        if d == last[key][0]:
            last[key] = (d, last[key][1] + float(r["Counter_Value"]))
json.dump({k: v[1] for k, v in last.items()}, open(o, "w"), indent=1)
print(json.load(open(o)))
PY
  find "$out/$c" -name "*.csv" -delete
done
rm -f "$out/state.pt"
python - "$out" <<'PY'
import json, os, sys
out = sys.argv[1]
fe, wr = json.load(open(os.path.join(out, "FETCH_SIZE.json"))), json.load(open(os.path.join(out, "WRITE_SIZE.json")))
d = {k: {"FETCH_SIZE_bytes_raw": fe.get(k, 0) * 1024.0, "WRITE_SIZE_bytes": wr.get(k, 0) * 1024.0} for k in set(fe) | set(wr)}
json.dump(d, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
PY
