#!/bin/bash
# Round-6 evidence in ONE GPU call -> gpurun_out/r06/ ; tools/assemble_r06.py then writes profiles/r06_* from it (stamped with the source hash of the tree: bench.source_hash()).  GPU box.
# Everything is collected on the SAME tree: bench line, per-op HIP-event times, rocprofv3 kernel stats, torch-issued kernels, idle gaps,
# SQ counters of k_shade_trace / k_h1_fwd / the chain kernels, FETCH_SIZE / WRITE_SIZE of the five roofline families (separate --pmc passes,
# --kernel-trace only), and the complete pixel-parity log at 512^2.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r06"; rm -rf "$out"; mkdir -p "$out"; cd "$root"
# the hash of the product sources ON THIS BOX (what the evidence is stamped with: ADVICE r5 -- not the hash of whatever tree assemble_r06.py later runs on)
python -c "import bench; print(bench.source_hash())" > "$out/source_hash.txt"
tools/gpu_suite_r06.sh > "$out/gpu_suite_tail.txt" 2>&1
python bench.py 2>/dev/null | tail -1 > "$out/bench_default.log"
python bench.py --op-times --no-cpu-baseline --extra-steps 0 --early-steps 0 2>/dev/null | tail -1 > "$out/bench_op_times.log"
tools/rocprof_iteration.sh r06/rocprof --extra-steps 0 > "$out/rocprof_head.txt" 2>&1
cp "$(find "$out/rocprof" -name '*_kernel_stats.csv' | head -1)" "$out/iteration_res256_kernel_stats.csv" 2>/dev/null
rm -rf "$out/rocprof"
python tools/torch_kernel_regions.py > "$out/torch_kernel_regions.txt" 2>/dev/null
python tools/torch_kernel_lines.py > "$out/torch_kernel_ops.txt" 2>/dev/null
python tools/chain_time.py > "$out/chain_time.txt" 2>/dev/null
python tools/bvh_stats.py > "$out/bvh_stats.txt" 2>/dev/null
tools/gpu_gaps.sh > "$out/gpu_gaps.txt" 2>&1
tools/pmc_script.sh r06_trace k_shade_trace tools/shade_time.py > "$out/pmc_trace.txt" 2>&1
tools/pmc_script.sh r06_h1 k_h1_fwd tools/h1_only.py 3 > "$out/pmc_h1.txt" 2>&1
tools/pmc_script.sh r06_wg16 k_h2_wgrad16 tools/chain_time.py > "$out/pmc_wg16.txt" 2>&1
tools/pmc_script.sh r06_bwd4 'k_h2_bwd<4>' tools/chain_time.py > "$out/pmc_bwd4.txt" 2>&1
tools/pmc_family_traffic.sh r06_traffic k_shade_samples k_shade_trace k_shade_accumulate k_shade_grad k_light_scan k_light_scatter k_light_reduce k_light_sum \
    k_encode_bwd k_encode_bin_reduce k_h1_fwd k_h2_fwd k_h2_bwd k_h2_wgrad > "$out/pmc_traffic.txt" 2>&1
for t in trace h1 wg16 bwd4 traffic; do mkdir -p "$out/pmc_$t"; cp "$root"/gpurun_out/pmc_r06_$t/*.stdout "$root"/gpurun_out/pmc_r06_$t/*.json "$out/pmc_$t/" 2>/dev/null; done
python -m pytest tests/test_pixel_fullsize_parity_gpu.py tests/test_raster_known_answers_gpu.py -m gpu -q -s -p no:cacheprovider > "$out/pixel_parity_512.txt" 2>&1
python -m pytest tests/test_config0_end_to_end_gpu.py -m gpu -q -s -p no:cacheprovider > "$out/chain_parity.txt" 2>&1
python -m pytest tests/test_ray_stage_fullsize_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s -p no:cacheprovider > "$out/ray_stage_and_fullsize.txt" 2>&1
cp "$root/gpurun_out/ray_stage_fullsize_parity.json" "$out/" 2>/dev/null
git -C "$root" rev-parse HEAD > "$out/commit.txt" 2>/dev/null || true
ls -la "$out"
