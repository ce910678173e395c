"""gpurun_out/r06/ (written by tools/collect_r06.sh in ONE GPU call) -> profiles/r06_*: copies of the logs, and the JSON records bench.py reads
(`traffic`, `binding`), every one stamped with the commit the evidence was collected at.   python tools/assemble_r06.py"""
import ast
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
P = os.path.join(ROOT, "profiles")


def sq_record(tag, desc, **extra):
    """tools/pmc_sq_json.py on gpurun_out/pmc_<tag> (its *.stdout files are kept by the collection script under r06/pmc_<short>/ as well)"""
    args = [sys.executable, os.path.join(ROOT, "tools", "pmc_sq_json.py"), tag, desc] + [f"{k}={v}" for k, v in extra.items()]
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode != 0:
        print(f"  (no counters for {tag}: {r.stderr.strip()[:200]})")
        return None
    return json.loads(r.stdout)


def main():
    sys.path.insert(0, ROOT)
    import bench
    # the stamp is the hash the GPU box computed over ITS copy of the sources when it collected (collect_r06.sh); if this tree has moved on since, the
    # evidence is not presented as current: refuse (re-collect) rather than stamp it with a hash it was not measured on
    collected = open(os.path.join(SRC, "source_hash.txt")).read().strip()
    if collected != bench.source_hash() and "--allow-stale" not in sys.argv:
        raise SystemExit(f"gpurun_out/r06 was collected on sources {collected}, this tree hashes {bench.source_hash()}: re-run tools/collect_r06.sh (or --allow-stale)")
    stamp = {"source_hash": collected, "collected_by": "tools/collect_r06.sh (one gpurun call; hash computed on the GPU box over the tree it ran)"}
    for name in ("bench_default.log", "bench_op_times.log", "iteration_res256_kernel_stats.csv", "torch_kernel_regions.txt", "torch_kernel_ops.txt", "chain_time.txt",
                 "bvh_stats.txt", "gpu_gaps.txt", "pixel_parity_512.txt", "chain_parity.txt", "ray_stage_and_fullsize.txt", "gpu_test_durations.txt",
                 "ray_stage_fullsize_parity.json"):
        if os.path.isfile(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(P, "r06_" + name))
    bench = json.loads(open(os.path.join(SRC, "bench_default.log")).read())
    shade = [r for r in [bench["roofline"]] + bench.get("roofline_others", []) if r.get("shadow_rays")]
    rays = (shade[0]["shadow_rays"] if shade else None) or 18747392

    # ---- SQ counters
    rec = sq_record("r06_trace", "k_shade_trace (8-ary Hilbert BVH, chunk per wave), bench frame", rays=rays, useful_per_ray=2400)
    if rec:
        rec.update(stamp)
        json.dump(rec, open(os.path.join(P, "r06_pmc_trace.json"), "w"), indent=1)
    rec = sq_record("r06_h1", "k_h1_fwd (gs_sdf_mlp_fwd_h1), tet-res256 grid, 2 282 489 rows")
    if rec:
        rec.update(stamp)
        json.dump(rec, open(os.path.join(P, "r06_pmc_h1.json"), "w"), indent=1)
    chain = {}
    for key, tag, desc in (("wg16", "r06_wg16", "k_h2_wgrad16 (last dispatch of tools/chain_time.py: the eikonal pass's weight gradients over 2 x 50 000 rows)"),
                           ("bwd4", "r06_bwd4", "k_h2_bwd<RR> (eikonal reverse chain with the second-order source, 50 000 samples)")):
        rec = sq_record(tag, desc)
        if rec:
            dv = rec["derived"]
            if "mean_resident_waves_per_simd" in dv:
                dv["avg_waves_per_simd"] = dv["mean_resident_waves_per_simd"]
            chain[key] = rec
    if chain:
        chain.update(stamp)
        json.dump(chain, open(os.path.join(P, "r06_pmc_sdf_chain.json"), "w"), indent=1)

    # ---- HBM traffic of the roofline families: bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE counts
    # half the bytes of wide streaming reads; gathers are uncalibrated, so for the traversal the truth lies between FETCH + WRITE and this)
    tf = os.path.join(ROOT, "gpurun_out", "pmc_r06_traffic", "traffic.json")
    if os.path.isfile(tf):
        t = json.load(open(tf))

        def fam(keys, note):
            per = {k: t[k] for k in t if any(k == kk or k.startswith(kk + "<") for kk in keys)}
            fe = sum(v["FETCH_SIZE_bytes_raw"] for v in per.values())
            wr = sum(v["WRITE_SIZE_bytes"] for v in per.values())
            return {"bytes": 2 * fe + wr, "FETCH_SIZE_bytes_raw": fe, "WRITE_SIZE_bytes": wr, "per_kernel": per, "note": note}
        npix = 4 * 512 * 512
        out = {
            "gs_env_shade_fwd": fam(["k_shade_samples", "k_shade_trace", "k_shade_accumulate"], "k_shade_samples<false> + k_shade_trace + k_shade_accumulate, last dispatch of each"),
            "gs_env_shade_bwd_saved": fam(["k_shade_grad", "k_light_scan", "k_light_scatter", "k_light_reduce", "k_light_sum"], "k_shade_grad + the light gradient's counting sort"),
            "gs_hashgrid_encode_bwd_binned": fam(["k_encode_bwd", "k_encode_bin_reduce"], "k_encode_bwd + k_encode_bin_reduce"),
            "k_h1_fwd": fam(["k_h1_fwd"], "one-product SDF forward over the 2 282 489 grid rows; the weight image is L2 resident"),
        }
        out["gs_env_shade_fwd"]["algorithmic_bytes"] = npix * 92
        out["gs_env_shade_bwd_saved"]["algorithmic_bytes"] = npix * 140
        out["k_h1_fwd"]["algorithmic_bytes"] = 2282489 * 16 + 2282489 // 8
        for k, v in out.items():
            if v.get("algorithmic_bytes"):
                v["traffic_over_algorithmic"] = round(v["bytes"] / v["algorithmic_bytes"], 2)
        # the chain kernels: one entry per kernel instantiation (last dispatch in the iteration)
        for k in t:
            if k.startswith(("k_h2_fwd", "k_h2_bwd", "k_h2_wgrad")):
                out[k] = {"bytes": 2 * t[k]["FETCH_SIZE_bytes_raw"] + t[k]["WRITE_SIZE_bytes"], **t[k], "note": "last dispatch of this instantiation in the iteration"}
        out["_meta"] = dict(stamp, command="tools/pmc_family_traffic.sh r06_traffic ... (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only)",
                            correction="bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half of a wide streaming read; gathers uncalibrated)")
        json.dump(out, open(os.path.join(P, "r06_pmc_traffic.json"), "w"), indent=1)
        print({k: (round(v["bytes"] / 1e6, 1), v.get("traffic_over_algorithmic")) for k, v in out.items() if k != "_meta"})

    # ---- binding resource per family (what bench.py prints as `binding`)
    b = {}
    try:
        dv = json.load(open(os.path.join(P, "r06_pmc_trace.json")))["derived"]
        b["gs_env_shade_fwd"] = {"resource": "VALU issue (k_shade_trace, the family's dominant kernel)", "valu_issue_busy": round(dv["valu_issue_busy_fraction"], 3),
                                 "useful_valu_lane_fraction": round(dv["useful_lane_fraction"], 3), "valu_lane_slots_per_ray": round(dv["valu_lane_slots_per_ray"], 1),
                                 "source": "profiles/r06_pmc_trace.json"}
    except Exception as e:
        print("  no trace binding:", e)
    try:
        dv = json.load(open(os.path.join(P, "r06_pmc_h1.json")))["derived"]
        b["gs_sdf_mlp_fwd_h1"] = {"resource": "MFMA pipe + VALU issue, largely back to back (tools/micro/mfma_fillers_waves.hip: a SIMD hides about half of the VALU work "
                                              "behind MFMAs even with four waves)", "mfma_pipe_busy": round(dv["mfma_pipe_busy_fraction"], 3),
                                  "valu_issue_busy": round(dv["valu_issue_busy_fraction"], 3), "lds_busy": round(dv["lds_busy_fraction"], 3), "source": "profiles/r06_pmc_h1.json"}
    except Exception as e:
        print("  no h1 binding:", e)
    try:
        d = json.load(open(os.path.join(P, "r06_pmc_sdf_chain.json")))
        for key, fams in (("wg16", ("gs_sdf_mlp_h2_wgrad",)), ("bwd4", ("gs_sdf_mlp_h2_bwd", "gs_sdf_eikonal_rr_bwd"))):
            if key in d:
                dv = d[key]["derived"]
                for f in fams:
                    b[f] = {"resource": "HBM planes (fp32, 7 KB per row and layer set) + an under-filled MFMA pipe", "mfma_pipe_busy": round(dv["mfma_pipe_busy_fraction"], 3),
                            "valu_issue_busy": round(dv.get("valu_issue_busy_fraction", 0), 3), "avg_waves_per_simd": round(dv.get("avg_waves_per_simd", 0), 2),
                            "kernel": d[key]["kernel"], "source": "profiles/r06_pmc_sdf_chain.json"}
    except Exception as e:
        print("  no chain binding:", e)
    try:
        t = json.load(open(os.path.join(P, "r06_pmc_traffic.json")))
        for f in ("gs_env_shade_bwd_saved", "gs_hashgrid_encode_bwd_binned"):
            b[f] = {"resource": "HBM records + VALU (k_shade_grad: 16-byte sample records in, 16-byte light records out)" if f.startswith("gs_env") else
                                "latency: scattered table gathers / record runs", "traffic_bytes": t[f]["bytes"], "traffic_over_algorithmic": t[f].get("traffic_over_algorithmic"),
                    "source": "profiles/r06_pmc_traffic.json"}
    except Exception as e:
        print("  no traffic binding:", e)
    b["_meta"] = stamp
    json.dump(b, open(os.path.join(P, "r06_binding.json"), "w"), indent=1)
    print(json.dumps(b, indent=1)[:1500])


if __name__ == "__main__":
    main()
