#!/bin/bash
# Round-4 evidence in one GPU call -> gpurun_out/r04/ (copy what is to be judged into profiles/).  GPU box.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r04"; mkdir -p "$out"; cd "$root"
python bench.py 2>/dev/null | tail -1 > "$out/bench_default.log"
python bench.py --op-times --no-cpu-baseline --extra-steps 0 2>/dev/null | tail -1 > "$out/bench_op_times.log"
tools/rocprof_iteration.sh r04/rocprof --extra-steps 0 > "$out/rocprof_head.txt" 2>&1
cp "$(find "$out/rocprof" -name '*_kernel_stats.csv' | head -1)" "$out/iteration_res256_kernel_stats.csv" 2>/dev/null
python tools/torch_kernel_regions.py > "$out/torch_kernel_regions.txt" 2>/dev/null
python tools/torch_kernel_lines.py > "$out/torch_kernel_ops.txt" 2>/dev/null
python tools/chain_time.py > "$out/chain_time.txt" 2>/dev/null
python tools/bvh_stats.py > "$out/bvh_stats.txt" 2>/dev/null
tools/pmc_script.sh r04_trace k_shade_trace tools/shade_time.py > "$out/pmc_trace.txt" 2>&1
python tools/pmc_sq_json.py r04_trace "k_shade_trace (8-ary Hilbert BVH, 32-bit record offsets), bench frame" rays=$(python - <<PY
import json
d=json.loads(open("$out/bench_default.log").read())
print(d["roofline"].get("shadow_rays") or 18747392)
PY
) useful_per_ray=2400 > "$out/pmc_trace.json" 2>/dev/null
ls -la "$out"
