"""gpurun_out/pmc_<tag>/sq*.stdout (tools/pmc_script.sh) -> one JSON record with the derived utilisations.
usage: python tools/pmc_sq_json.py <tag> "<kernel description>" [rays=<n>] [useful_per_ray=<instr>]  > profiles/<file>.json"""
import ast
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, desc = sys.argv[1], sys.argv[2]
extra = dict(kv.split("=", 1) for kv in sys.argv[3:])
src = os.path.join(root, "gpurun_out", "pmc_" + tag)
c = {}
for p in sorted(os.listdir(src)):
    if p.endswith(".stdout"):
        for line in open(os.path.join(src, p)):
            if "{" in line:
                c.update(ast.literal_eval(line[line.index("{"):].strip()))
if not c:
    sys.exit("no counters found")
cyc = c.get("SQ_BUSY_CYCLES", 0) / 32.0                 # per shader engine -> kernel cycles
d = {"kernel": desc, "command": f"tools/pmc_script.sh {tag} (rocprofv3 --pmc, one pass per <= 8 counters, --kernel-trace only)", "counters_summed_over_all_SQs": c, "derived": {}}
dv = d["derived"]
if cyc:
    dv["kernel_cycles"] = cyc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        dv["mfma_pipe_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
    if "SQ_ACTIVE_INST_VALU" in c:
        dv["valu_issue_busy_fraction"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)
    if "SQ_ACTIVE_INST_SCA" in c:
        dv["salu_busy_fraction"] = c["SQ_ACTIVE_INST_SCA"] * 4 / (cyc * 1024)
    if "SQ_LDS_IDX_ACTIVE" in c:
        dv["lds_busy_fraction"] = c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)
        dv["lds_bank_conflict_fraction_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c["SQ_LDS_IDX_ACTIVE"], 1)
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        dv["mean_resident_waves_per_simd"] = w * 4 / (cyc * 1024)
        dv["wave_cycles_split"] = {k: c[k2] / w for k, k2 in (("wait_inst_any", "SQ_WAIT_INST_ANY"), ("wait_any", "SQ_WAIT_ANY"), ("active", "SQ_ACTIVE_INST_ANY")) if k2 in c}
    if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
        dv["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
if "rays" in extra and "SQ_INSTS_VALU" in c:
    rays = float(extra["rays"])
    dv["valu_lane_slots_per_ray"] = c["SQ_INSTS_VALU"] * 64 / rays
    if "useful_per_ray" in extra:
        dv["useful_lane_instructions_per_ray_estimate"] = float(extra["useful_per_ray"])
        dv["useful_lane_fraction"] = float(extra["useful_per_ray"]) / dv["valu_lane_slots_per_ray"]
d["notes"] = "SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES is per shader engine (/ 32); SQ_VALU_MFMA_BUSY_CYCLES in cycles over 1024 SIMDs"
print(json.dumps(d, indent=1))
