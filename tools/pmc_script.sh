#!/bin/bash
# SQ / LDS / L1 counters of one kernel run by a small python script (separate rocprofv3 --pmc passes).  GPU box.
#   usage: tools/pmc_script.sh <tag> <kernel-substring[,substring2,...]> <script.py> [args]     -> gpurun_out/pmc_<tag>/*.stdout
tag="$1"; sub="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_$tag"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {
  p="$1"; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d "$out/$p" -o r --output-format csv -- python "$root/$SCRIPT" $SARGS > "$out/$p.log" 2>&1 </dev/null
  f=$(find "$out/$p" -name "*counter_collection.csv" | head -1)
  python - "$f" "$sub" > "$out/$p.stdout" <<'PY'
import csv, sys, collections
subs = sys.argv[2].split(",")            # several kernels from ONE set of passes: "k_a,k_b" -> one line per kernel, prefixed with its name
for sub in subs:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sys.argv[1])):
        if sub in r["Kernel_Name"]:
            agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    if agg:
        d = max(agg)
        print(*([sub] if len(subs) > 1 else []), d, dict(agg[d]))
PY
  cat "$out/$p.stdout"
  find "$out/$p" -name "*.csv" -delete
}
SCRIPT="$1"; shift; SARGS="$@"
pass sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_VALU SQ_ACTIVE_INST_MISC
