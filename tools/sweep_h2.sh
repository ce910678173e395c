#!/bin/bash
# GPU box: rebuild csrc/mlp_h2.hip with each "-D..." variant given as an argument and time the forward kernel.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$root"
for v in "$@"; do
  touch gshell_amd/csrc/mlp_h2.hip
  make -C gshell_amd/csrc EXTRA="$v" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== $v: $(python tools/mlp_time.py h2 2>&1 | tail -1)"
done
touch gshell_amd/csrc/mlp_h2.hip; make -C gshell_amd/csrc > /dev/null 2>&1
