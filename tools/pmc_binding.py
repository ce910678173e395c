"""profiles/r04_binding.json: for every kernel family bench.py prints a roofline record for, the resource the rocprofv3 --pmc passes show it is
limited by and the measured utilisation of that resource.  Derived from the committed PMC summaries; re-run after new PMC passes.

    python tools/pmc_binding.py
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(*names):
    for n in names:
        try:
            with open(os.path.join(P, n)) as f:
                return json.load(f), n
        except Exception:
            continue
    return None, None


def main():
    out = {}
    d, src = load("r04_pmc_trace.json", "r03_pmc_trace_8ary.json")
    if d:
        dv = d["derived"]
        useful = dv.get("useful_lane_fraction")
        if useful is None and "valu_lane_slots_per_ray" in dv:
            useful = round(2500.0 / dv["valu_lane_slots_per_ray"], 3)
        out["gs_env_shade_fwd"] = {"resource": "VALU issue (k_shade_trace, the family's dominant kernel)", "valu_issue_busy": round(dv["valu_issue_busy_fraction"], 3),
                                   "useful_valu_lane_fraction": useful, "valu_lane_slots_per_ray": round(dv.get("valu_lane_slots_per_ray", 0), 1), "source": "profiles/" + src}
    d, src = load("r04_pmc_h1.json", "r03_pmc_h1.json")
    if d:
        dv = d["derived"]
        out["gs_sdf_mlp_fwd_h1"] = {"resource": "MFMA pipe + VALU issue back to back", "mfma_pipe_busy": round(dv["mfma_pipe_busy_fraction"], 3),
                                    "valu_issue_busy": round(dv["valu_issue_busy_fraction"], 3), "lds_busy": round(dv["lds_busy_fraction"], 3), "source": "profiles/" + src}
    d, src = load("r04_pmc_sdf_chain.json", "r03_pmc_sdf_chain.json")
    if d:
        for key, fam in (("wg16", "gs_sdf_mlp_h2_wgrad"), ("bwd2", "gs_sdf_mlp_h2_bwd"), ("fwd2", "gs_sdf_mlp_h2_save_fwd")):
            if key in d:
                dv = d[key]["derived"]
                out[fam] = {"resource": "MFMA pipe (under-filled) + HBM planes", "mfma_pipe_busy": round(dv["mfma_pipe_busy_fraction"], 3),
                            "avg_waves_per_simd": round(dv.get("avg_waves_per_simd", 0), 2), "kernel": d[key]["kernel"], "source": "profiles/" + src}
    d, src = load("r04_pmc_hashgrid_shade_kernels.json", "r03_pmc_hashgrid_shade_kernels.json")
    if d:
        for key, fam in (("k_encode_bwd", "gs_hashgrid_encode_bwd_binned"), ("k_shade_grad", "gs_env_shade_bwd_saved")):
            if key in d:
                dv = d[key].get("derived", {})
                out[fam] = {"resource": "latency (waves waiting)" if fam.startswith("gs_hash") else "VALU issue + HBM records",
                            **{k: round(v, 3) for k, v in dv.items() if isinstance(v, (int, float)) and ("busy" in k or "wait" in k)}, "source": "profiles/" + src}
    with open(os.path.join(P, "r04_binding.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
