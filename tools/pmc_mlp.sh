#!/bin/bash
# SQ / LDS / L1 counters of the SDF-MLP forward kernel (separate passes of <= 8 counters).  GPU box.  Output: gpurun_out/pmc_mlp/
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_mlp"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {
  tag="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d "$out/$tag" -o r --output-format csv -- python "$root/tools/mlp_only.py" 3 > "$out/$tag.log" 2>&1 </dev/null
  f=$(find "$out/$tag" -name "*counter_collection.csv" | head -1)
  python - "$f" > "$out/$tag.stdout" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if "h2_fwd" in r["Kernel_Name"] or "sdf_mlp" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, c in list(agg.items())[-1:]:
    print(d, dict(c))
PY
  cat "$out/$tag.stdout"
  find "$out/$tag" -name "*.csv" -size +5M -delete
}
pass sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_VALU SQ_ACTIVE_INST_MISC
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum
# HBM traffic of the same kernel: FETCH_SIZE / WRITE_SIZE in their own pass (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts
# 64 B per 128-B request for wide streaming reads -> double it; units are kilobytes)
pass hbm_fetch FETCH_SIZE
pass hbm_write WRITE_SIZE
