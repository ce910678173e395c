#!/bin/bash
# rocprofv3 kernel trace of the default bench workload -> gpurun_out/<tag>/  (GPU box).  usage: tools/rocprof_iteration.sh <tag> [bench args]
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# set-up (torch pre-fit of the SDF network) once, unprofiled; the profiled process loads the saved state, so the kernel
# statistics below contain the training iterations (+ the target renders) only
python "$root/bench.py" --no-cpu-baseline --early-steps 0 --steps 1 --warmup 0 --state-file "$out/state.pt" "$@" > "$out/setup.log" 2>&1 </dev/null
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python "$root/bench.py" --no-cpu-baseline --state-file "$out/state.pt" "$@" > "$out/run.log" 2>&1 </dev/null
rm -f "$out/state.pt"
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" -o -name "*.json" -o -name "*.pftrace" \) -delete   # only the stats summary is kept (traces exceed the merge cap)
find "$out" -name "*_kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -45 {} | cut -c1-160'
tail -1 "$out/run.log" | cut -c1-300
