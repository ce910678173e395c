#!/bin/bash
# chain tests with the eikonal side stream ON, N runs against one library: tools/race_loop.sh <label> <library or ""> [runs]   -> gpurun_out/race/loops.txt   GPU box
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; out="$root/gpurun_out/race"; mkdir -p "$out"
label="$1"; lib="$2"; runs="${3:-20}"; fails=0
for i in $(seq 1 "$runs"); do
  if [ -n "$lib" ]; then export GSHELL_HIP_LIB="$lib"; else unset GSHELL_HIP_LIB; fi
  GSHELL_EIKONAL_SIDE_STREAM=1 timeout 300 python -m pytest tests/test_config0_end_to_end_gpu.py -m gpu -q -x -p no:cacheprovider > "$out/loop_${label}_$i.txt" 2>&1 || { fails=$((fails+1)); continue; }
  rm -f "$out/loop_${label}_$i.txt"
done
echo "side stream ON, library $label: failed runs: $fails of $runs" | tee -a "$out/loops.txt"
