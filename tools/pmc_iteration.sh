#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) of EVERY kernel of two training iterations -> profiles/r02_pmc_traffic_iteration.json
# (GPU box; ~1 minute).  Set-up runs unprofiled and hands its state over, as in rocprof_iteration.sh.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_iter"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python "$root/bench.py" --no-cpu-baseline --early-steps 0 --steps 1 --warmup 0 --state-file "$out/state.pt" > "$out/setup.log" 2>&1 </dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d "$out/$c" -o r --output-format csv -- python "$root/bench.py" --no-cpu-baseline --early-steps 0 --steps 2 --warmup 1 --state-file "$out/state.pt" > "$out/$c.log" 2>&1 </dev/null
done
rm -f "$out/state.pt"
python "$root/tools/pmc_iteration_json.py" "$out" > "$root/gpurun_out/r02_pmc_traffic_iteration.json"
find "$out" -name "*.csv" -size +2M -delete
head -c 1500 "$root/gpurun_out/r02_pmc_traffic_iteration.json"
