#!/bin/bash
# Build libgshell_hip.so variants with different -D flags and run a command against each (GPU box).
# usage: tools/bench_variants.sh "<cmd>" "-DFOO=1" "-DFOO=2" ...
cmd="$1"; shift
cd "$(dirname "$0")/../gshell_amd/csrc"
cp ../lib/libgshell_hip.so /tmp/libgshell_hip.orig.so
for flags in "$@"; do
  touch envshade.hip bvh.hip
  make EXTRA="$flags" >/dev/null 2>&1 || { echo "build failed for $flags"; continue; }
  echo "== $flags"
  (cd ../.. && eval "$cmd")
done
cp /tmp/libgshell_hip.orig.so ../lib/libgshell_hip.so
