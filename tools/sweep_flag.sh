#!/bin/bash
# A/B a compile-time experiment flag: builds with and without -D<FLAG> and prints ms/iter and one op time (GPU box).
# usage: tools/sweep_flag.sh <source.hip> <FLAG> <op-key>
cd "$(dirname "$0")/.."
src="$1"; flag="$2"; key="$3"
cp gshell_amd/lib/libgshell_hip.so /tmp/libgshell_hip.orig.so
for extra in "" "-D$flag"; do
  touch "gshell_amd/csrc/$src"
  timeout 240 make -C gshell_amd/csrc EXTRA="$extra" >/dev/null 2>&1 </dev/null || { echo "build failed for '$extra'"; continue; }
  echo -n "[$extra]: "
  timeout 150 python bench.py --no-cpu-baseline --op-times --steps 8 2>/dev/null </dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['op_ms'].get('$key'))"
done
cp /tmp/libgshell_hip.orig.so gshell_amd/lib/libgshell_hip.so
