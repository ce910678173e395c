"""gpurun_out/pmc_mlp/*.log (tools/pmc_mlp.sh) -> profiles/r02_pmc_mlp_h2.json + profiles/r02_pmc_traffic.json"""
import ast
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc_mlp")
c = {}
for tag in ("sq1", "sq2", "sq3", "tcp", "hbm_fetch", "hbm_write"):
    p = os.path.join(src, tag + ".stdout")
    if os.path.isfile(p):
        for line in open(p):
            if "{" in line:
                c.update(ast.literal_eval(line[line.index("{"):].strip()))
if not c:
    sys.exit("no counters found (run tools/pmc_mlp.sh > per-pass stdout files first)")
sq_busy = c.get("SQ_BUSY_CYCLES", 0) / 32.0            # per shader engine -> cycles of the kernel
d = {"kernel": "k_h2_fwd<GRID> (gs_sdf_mlp_fwd_h2), tet-res256 grid, 2 282 489 rows", "command": "tools/pmc_mlp.sh (rocprofv3 --pmc, one pass per <= 8 counters, python tools/mlp_only.py 3)",
     "counters_summed_over_all_SQs": c, "derived": {}}
if sq_busy:
    d["derived"]["kernel_cycles"] = sq_busy
    d["derived"]["mfma_pipe_busy_fraction"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (sq_busy * 1024)
    if "SQ_ACTIVE_INST_VALU" in c:
        d["derived"]["valu_busy_fraction"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (sq_busy * 1024)
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        d["derived"]["wave_cycles_split"] = {k: c[k2] / w for k, k2 in (("wait_inst_any", "SQ_WAIT_INST_ANY"), ("wait_any", "SQ_WAIT_ANY"), ("active", "SQ_ACTIVE_INST_ANY")) if k2 in c}
    if "SQ_LDS_IDX_ACTIVE" in c:
        d["derived"]["lds_busy_fraction"] = c["SQ_LDS_IDX_ACTIVE"] / (sq_busy * 256)
        d["derived"]["lds_bank_conflict_fraction_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in c:
        d["derived"]["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0))
d["notes"] = "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_f16, 1024 SIMDs); SQ_BUSY_CYCLES is per shader engine (32); wave counters are quad-cycles."
json.dump(d, open(os.path.join(root, "profiles", "r02_pmc_mlp_h2.json"), "w"), indent=1)
if "FETCH_SIZE" in c:
    fetch, write = c["FETCH_SIZE"] * 1024.0, c.get("WRITE_SIZE", 0) * 1024.0
    t = {"k_h2_fwd": {"bytes": 2 * fetch + write, "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write, "algorithmic_bytes": 16 * 2282489,
                      "note": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md for wide streaming reads); the 3.3 MB weight image is L2 resident"}}
    json.dump(t, open(os.path.join(root, "profiles", "r02_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(d["derived"], indent=1))
