#!/bin/bash
# Round-3 closing evidence in one GPU call -> gpurun_out/r03f/ (copy what is to be judged into profiles/).  GPU box.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r03f"; mkdir -p "$out"; cd "$root"
python bench.py 2>/dev/null | tail -1 > "$out/bench_default.log"
python bench.py --op-times --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_op_times.log"
python bench.py --geometry flexicubes --res 80 --no-cpu-baseline 2>/dev/null | tail -1 > "$out/bench_flexicubes_res80.log"
tools/rocprof_iteration.sh r03f/rocprof > "$out/rocprof_head.txt" 2>&1
python tools/torch_kernel_regions.py > "$out/torch_kernel_regions.txt" 2>/dev/null
python tools/torch_kernel_lines.py > "$out/torch_kernel_ops.txt" 2>/dev/null
python tools/chain_time.py > "$out/chain_time.txt" 2>/dev/null
tools/pmc_family_traffic.sh r03f_traffic k_shade_samples k_shade_trace k_shade_accumulate k_shade_grad k_encode_bwd k_encode_bin_reduce k_light_reduce k_adam > "$out/pmc_traffic.txt" 2>&1
cp "$root/gpurun_out/pmc_r03f_traffic/traffic.json" "$out/pmc_traffic.json" 2>/dev/null
ls -la "$out"
