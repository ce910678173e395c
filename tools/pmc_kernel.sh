#!/bin/bash
# SQ / L1 / L2 / HBM counters of ONE kernel of the training iteration (separate rocprofv3 --pmc passes of <= 8 counters, last
# dispatch whose name contains <substring>).  GPU box.
#   usage: tools/pmc_kernel.sh <tag> <kernel-name-substring> [bench args]      -> gpurun_out/pmc_<tag>/{*.stdout, summary.json}
tag="$1"; sub="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/pmc_$tag"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python "$root/bench.py" --no-cpu-baseline --early-steps 0 --steps 1 --warmup 0 --state-file "$out/state.pt" "$@" > "$out/setup.log" 2>&1 </dev/null
pass() {
  p="$1"; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d "$out/$p" -o r --output-format csv -- python "$root/bench.py" --no-cpu-baseline --early-steps 0 --steps 2 --warmup 1 --state-file "$out/state.pt" > "$out/$p.log" 2>&1 </dev/null
  f=$(find "$out/$p" -name "*counter_collection.csv" | head -1)
  python - "$f" "$sub" > "$out/$p.stdout" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
if agg:
    d = max(agg)
    print(d, dict(agg[d]))
PY
  cat "$out/$p.stdout"
  find "$out/$p" -name "*.csv" -delete
}
pass sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_ACCESSES_sum
pass hbm_fetch FETCH_SIZE
pass hbm_write WRITE_SIZE
rm -f "$out/state.pt"
python - "$out" "$sub" <<'PY'
import ast, json, os, sys
out, sub = sys.argv[1], sys.argv[2]
c = {}
for p in ("sq1", "sq2", "tcp", "hbm_fetch", "hbm_write"):
    f = os.path.join(out, p + ".stdout")
    if os.path.isfile(f):
        for line in open(f):
            if "{" in line:
                c.update(ast.literal_eval(line[line.index("{"):].strip()))
d = {"kernel_substring": sub, "counters_summed_over_all_SQs": c, "derived": {}}
busy = c.get("SQ_BUSY_CYCLES", 0) / 32.0
if busy:
    d["derived"]["kernel_cycles"] = busy
    if "SQ_ACTIVE_INST_VALU" in c:
        d["derived"]["valu_busy_fraction"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (busy * 1024)
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        d["derived"]["wave_cycles_split"] = {k: c[k2] / w for k, k2 in (("wait_inst_any", "SQ_WAIT_INST_ANY"), ("wait_any", "SQ_WAIT_ANY"), ("active", "SQ_ACTIVE_INST_ANY")) if k2 in c}
        d["derived"]["mean_resident_waves_per_simd"] = w * 4 / (busy * 1024)
if "TCC_HIT_sum" in c:
    d["derived"]["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1)
if "TCP_TCC_READ_REQ_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
    d["derived"]["l1_read_requests_to_l2_per_l1_access"] = c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1)
    d["derived"]["l2_to_l1_bytes_at_128B_per_request"] = c["TCP_TCC_READ_REQ_sum"] * 128
if "FETCH_SIZE" in c:
    d["derived"]["hbm_bytes"] = {"FETCH_SIZE_raw": c["FETCH_SIZE"] * 1024.0, "WRITE_SIZE": c.get("WRITE_SIZE", 0) * 1024.0,
                                 "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide streaming reads (MI355X_MICROARCH.md); scattered 48-64 B record fetches are not doubled"}
d["notes"] = "SQ_BUSY_CYCLES is per shader engine (32); wave counters are quad-cycles; one dispatch (the last of the run)."
json.dump(d, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(d["derived"], indent=1))
PY
