#!/bin/bash
# The WHOLE GPU suite + smoke() in one lease, with per-test durations -> gpurun_out/r06/gpu_test_durations.txt (copied to profiles/r06_gpu_test_durations.txt;
# tests/test_bench_evidence_cpu.py fails if the recorded wall time exceeds 600 s).  GPU box.
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
{
  python -c "import bench; print('source_hash', bench.source_hash())"
  timeout 1300 python -m pytest tests -m gpu -q --durations=0 -p no:cacheprovider
  echo "rc=$?"
  python - <<'PY'
import time, __graft_entry__ as g
t0 = time.time()
g.smoke()
print(f"smoke_rc=0 smoke_seconds={time.time() - t0:.1f}")
PY
} > "$out/gpu_test_durations.txt" 2>&1
grep -n "passed\|failed\|rc=\|smoke\|source_hash" "$out/gpu_test_durations.txt" | tail -8
