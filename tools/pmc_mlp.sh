#!/bin/bash
# SQ counters of k_sdf_mlp_fwd (one pass, 8 SQ slots).  GPU box.  Output: gpurun_out/pmc_mlp/
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_mlp"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU \
  --kernel-trace -d "$out" -o r --output-format csv -- python "$root/tools/mlp_only.py" 3 > "$out/run.log" 2>&1 </dev/null
f=$(find "$out" -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "sdf_mlp" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, c in list(agg.items())[-2:]:
    print(d, dict(c))
PY
find "$out" -name "*.csv" -size +5M -delete
