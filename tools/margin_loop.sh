#!/bin/bash
# How close do the chain tests come to their bars?  N runs of tests/test_config0_end_to_end_gpu.py with its prints kept -> gpurun_out/margins/run_*.txt ; tools/margin_report.py
# then lists, per (chain, tensor), the largest measured / bar ratio over the runs.  Also M runs of the whole GPU suite (flaky tests anywhere?).  GPU box.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$root"; out="$root/gpurun_out/margins"; rm -rf "$out"; mkdir -p "$out"
runs="${1:-30}"; suites="${2:-2}"
for i in $(seq 1 "$runs"); do
  timeout 300 python -m pytest tests/test_config0_end_to_end_gpu.py -m gpu -q -s -p no:cacheprovider > "$out/run_$i.txt" 2>&1; echo "rc=$?" >> "$out/run_$i.txt"
done
for i in $(seq 1 "$suites"); do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$out/suite_$i.txt" 2>&1; echo "rc=$?" >> "$out/suite_$i.txt"; tail -3 "$out/suite_$i.txt"
done
grep -l "rc=[1-9]" "$out"/run_*.txt | wc -l
