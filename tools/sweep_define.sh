#!/bin/bash
# Rebuild libgshell_hip.so with -D<NAME>=<value> for each value and run a short bench (GPU box).  Everything under timeout.
# usage: tools/sweep_define.sh <source.hip> <NAME> <op-key> v1 v2 ...
cd "$(dirname "$0")/.."
src="$1"; name="$2"; key="$3"; shift 3
cp gshell_amd/lib/libgshell_hip.so /tmp/libgshell_hip.orig.so
for v in "$@"; do
  touch "gshell_amd/csrc/$src"
  timeout 240 make -C gshell_amd/csrc EXTRA="-D$name=$v" >/dev/null 2>&1 </dev/null || { echo "build failed for $v"; continue; }
  echo -n "$name=$v: "
  timeout 150 python bench.py --no-cpu-baseline --op-times --steps 8 2>/dev/null </dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['op_ms'].get('$key'))"
done
cp /tmp/libgshell_hip.orig.so gshell_amd/lib/libgshell_hip.so
