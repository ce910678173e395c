#!/bin/bash
# Round-3 evidence in one GPU call -> gpurun_out/r03/ (copy what is to be judged into profiles/).  GPU box.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03"; mkdir -p "$out"; cd "$root"
python bench.py 2>/dev/null | tail -1 > "$out/bench_default.log"
python bench.py --op-times --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_op_times.log"
python bench.py --geometry flexicubes --res 80 --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_flexicubes_res80.log"
tools/rocprof_iteration.sh r03/rocprof > "$out/rocprof_head.txt" 2>&1
python tools/two_pass.py > "$out/two_pass.json" 2>/dev/null
python tools/two_pass.py --states > "$out/two_pass_states.json" 2>/dev/null
python tools/bvh_stats.py > "$out/bvh_stats.txt" 2>/dev/null
python tools/torch_kernel_regions.py > "$out/torch_kernel_regions.txt" 2>/dev/null
python tools/find_syncs.py > "$out/find_syncs.txt" 2>/dev/null
python tools/render_grad_diag.py 2>/dev/null | grep -v Warning > "$out/render_grad_diag.txt"
python tools/chain_time.py > "$out/chain_time.txt" 2>/dev/null
tools/pmc_script.sh r03_h1 k_h1_fwd tools/h1_only.py 3 > "$out/pmc_h1.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_h1_$c" -o r --output-format csv -- python "$root/tools/h1_only.py" 3 > /dev/null 2>&1 </dev/null
  f=$(find "$out/pmc_h1_$c" -name "*counter_collection.csv" | head -1)
  python - "$f" >> "$out/pmc_h1.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "k_h1_fwd" in r["Kernel_Name"]:
        agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
d = max(agg); print(d, dict(agg[d]))
PY
  rm -rf "$out/pmc_h1_$c"
done
ls -la "$out"
