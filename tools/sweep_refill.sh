#!/bin/bash
# Sweep TRACE_REFILL on the GPU box (each variant: rebuild envshade.hip, short bench).  Everything under timeout.
cd "$(dirname "$0")/.."
cp gshell_amd/lib/libgshell_hip.so /tmp/libgshell_hip.orig.so
for v in "$@"; do
  touch gshell_amd/csrc/envshade.hip
  timeout 240 make -C gshell_amd/csrc EXTRA="-DTRACE_REFILL=$v" >/dev/null 2>&1 || { echo "build failed for $v"; continue; }
  echo -n "TRACE_REFILL=$v: "
  timeout 150 python bench.py --no-cpu-baseline --op-times --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['op_ms']['gs_env_shade_fwd'])"
done
cp /tmp/libgshell_hip.orig.so gshell_amd/lib/libgshell_hip.so
