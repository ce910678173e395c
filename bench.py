"""Benchmark of the G-Shell training iteration on MI355X (contract: see the round prompt).

    python bench.py [--gpus N --steps K --warmup W]           # N>1: launched by torch.distributed.run, one rank per GPU

One "step" = one full optimisation iteration (train_gshelltet_deepfashion.py:395-478 bracket): zero_grad, lgt.update_pdf,
geometry.tick (SDF MLP over the whole tet grid -> G-MarchingTets extraction -> BVH rebuild -> rasterise -> interpolate ->
hash-grid texture -> Monte-Carlo env shading with shadow rays -> bilateral denoise -> composite + antialias -> losses),
backward, 3 Adam steps, clamps -- on synthetic inputs of BASELINE.json configs[2]: tet-res256, 4 views of 512x512 per GPU,
n_samples = 8 (128 shadow rays / covered pixel / pass).  Views are sharded across ranks; one RCCL all-reduce of the flat
gradient per iteration.

  --gpus N            N ranks on N devices.  If this process was NOT started by torch.distributed.run (no WORLD_SIZE in the
                      environment) and N > 1, bench.py re-launches ITSELF under `python -m torch.distributed.run
                      --nproc-per-node N`; it fails loudly if fewer than N devices exist.  `n_gpus` in the JSON line is the
                      world size the process group reports, never the flag.
  --global-batch G    fixed global batch (strong scaling): G views dealt round-robin over the ranks, e.g.
                      `--gpus 8 --global-batch 8` = BASELINE.json configs[3] (1 view / GPU).  Default: 4 views per GPU (weak).
  --schedule-it I     iteration counter the timed steps start from (default 1000 = steady state of the reference's
                      schedule: shadow_scale 1, denoiser sigma 2 => 23 x 23 bilateral taps; gshell_tets_geometry.py:264-268,
                      denoiser.py:26-29).  The it = 0 figure (sigma ~ 0, 7 x 7 taps) is measured beside it (`early_schedule`).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver (before HIP initialises)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=256, help="tet grid resolution (64/128/256)")
    ap.add_argument("--views", type=int, default=4, help="views per GPU (weak scaling; ignored with --global-batch)")
    ap.add_argument("--global-batch", type=int, default=None, help="fixed global batch dealt over the ranks (strong scaling); --gpus 8 --global-batch 8 = configs[3]")
    ap.add_argument("--schedule-it", type=int, default=1000, help="iteration counter at the start of the warm-up (1000 = steady-state schedule)")
    ap.add_argument("--early-steps", type=int, default=5, help="timed steps of the extra it=0 measurement (0 = skip)")
    ap.add_argument("--extra-steps", type=int, default=5, help="timed steps of each side measurement printed inside the same JSON line: the conservative "
                    "one-pass SDF forward, a close camera (35-40 %% coverage), and at 8 ranks BASELINE configs[3] (global batch 8); 0 = skip")
    ap.add_argument("--camera-radius", type=float, default=2.2, help="orbit radius of the synthetic cameras (2.2: the garment covers ~14 %% of the frame)")
    ap.add_argument("--train-res", type=int, default=512)
    ap.add_argument("--n-samples", type=int, default=8)
    ap.add_argument("--fit-steps", type=int, default=400)
    ap.add_argument("--geometry", choices=["tets", "flexicubes"], default="tets", help="flexicubes + --res 80 = BASELINE.json configs[4]")
    ap.add_argument("--set", action="append", default=[], metavar="FLAG=VALUE", help="override a training flag (python literal), e.g. --set eikonal_side_stream=False")
    ap.add_argument("--state-file", default=None, help="save the fitted set-up state here / load it if the file exists (profiling runs skip the set-up kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-config", action="store_true", help="skip the side record at the reference's own default workload (2 x 1024^2, n_samples 24)")
    ap.add_argument("--variant-ok", action="store_true", help="measure a library built with non-default compile-time variants (GSHELL_HIP_LIB=...; the line then carries "
                    "`build_flags` and is not a headline)")
    ap.add_argument("--op-times", action="store_true", help="also print per-op HIP-event times (adds sync points)")
    return ap.parse_args()


def relaunch_multi_rank(n):
    """`python bench.py --gpus N` typed by hand: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("GSHELL_BENCH_SAME_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node -- refusing to run fewer ranks than asked")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        relaunch_multi_rank(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # GSHELL_BENCH_SAME_DEVICE=1: plumbing test of the multi-rank path on a ONE-GPU box -- every rank uses device 0 and the
    # ranks talk over gloo (RCCL refuses two ranks on one device).  Never a performance number; the JSON line says so.
    same_device = os.environ.get("GSHELL_BENCH_SAME_DEVICE") == "1"
    dev_index = 0 if same_device else local_rank
    if torch.cuda.device_count() <= dev_index:
        raise SystemExit(f"bench.py: rank {rank} has no device {dev_index} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        world = dist.get_world_size()             # what RCCL actually sees
    from gshell_amd import _lib, workload
    from gshell_amd.train import ViewShard
    build_flags = _lib.lib().gs_build_flags().decode()          # fails loudly if the HIP library is missing
    if build_flags and not a.variant_ok:
        # several compile-time switches are timing-only ablations with wrong results: no number without an explicit opt-in
        raise SystemExit(f"bench.py: the loaded library was built with non-default variants ({build_flags}); pass --variant-ok to measure it anyway")
    shard = ViewShard(rank, world)
    if a.global_batch is not None:
        if a.global_batch % world:
            raise SystemExit(f"bench.py: --global-batch {a.global_batch} does not divide over {world} ranks")
        B_global = a.global_batch
    else:
        B_global = a.views * world
    B_local = len(shard.local_views(B_global))
    H = W = a.train_res
    import ast
    overrides = {kv.split('=', 1)[0]: ast.literal_eval(kv.split('=', 1)[1]) for kv in a.set}
    trainer = workload.build(res=a.res, n_samples=a.n_samples, batch=B_global, train_res=(H, W), shard=shard, fit_steps=a.fit_steps, geometry=a.geometry,
                             state_file=a.state_file, **overrides)
    # this rank's views of every global batch: ids [it*B + r, it*B + r + world, ...]
    n_iters = a.warmup + a.steps
    targets = [workload.make_targets(trainer, [(it * B_global + v) % 72 for v in shard.local_views(B_global)], (H, W), radius=a.camera_radius)
               for it in range(min(n_iters, 4))]
    # HIP events on the launch stream around the C-ABI calls: by default only around the candidates for the dominant kernel
    # (~10 event pairs per step); --op-times brackets every entry point (~250 pairs per step, taxes the headline slightly)
    _lib.enable_op_timing(True, only=None if a.op_times else ROOFLINE_CANDIDATES)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(schedule_it, warmup, steps, tgts=None, gb=None):
        """`warmup` untimed + `steps` timed iterations with the schedule counter starting at `schedule_it`; max over ranks."""
        tgts, gb = tgts or targets, gb or B_global
        trainer.it = schedule_it
        for it in range(warmup):
            trainer.step(tgts[it % len(tgts)], global_batch=gb)
        barrier()
        _lib.reset_op_timing()
        t0 = time.perf_counter()
        for it in range(steps):
            trainer.step(tgts[(warmup + it) % len(tgts)], global_batch=gb)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), _lib.op_timing_summary()

    early = None
    if a.early_steps > 0 and a.schedule_it != 0:       # it = 0 schedule beside the headline (sigma ~ 0: 7 x 7 taps, shadow_scale ~ 0)
        dt0, _ = timed(0, 2, a.early_steps)
        early = {"schedule_it": 0, "steps": a.early_steps, "ms_per_step": round(dt0 / a.early_steps * 1e3, 3),
                 "value": round(B_global * H * W * a.early_steps / dt0 / 1e6, 4), "bilateral_radius": 2 * math.ceil(2.5 * trainer.denoiser.sigma) + 1 if trainer.denoiser else None}
    dt, op_times = timed(a.schedule_it, a.warmup, a.steps)
    from gshell_amd.geometry import mlp as _mlp
    from gshell_amd.render import optixutils as _ou
    covered_main = _ou.last_covered_pixels
    rows_main = dict(_mlp.LAST_CHAIN_ROWS)
    two_pass_ran = "gs_sdf_mlp_fwd_h1" in op_times
    margin_used = getattr(trainer.geometry, "sdf_net", None) and trainer.geometry.sdf_net.__dict__.get("_gs_two_pass_margin_used")
    side = {}
    if a.extra_steps > 0:
        # (1) the conservative figure: the SDF network's full-grid forward in ONE pass of the fp16-pair arithmetic (no one-product pass)
        if two_pass_ran:
            _mlp.SDF_TWO_PASS = False
            dt1, _ = timed(a.schedule_it, 3, a.extra_steps)
            _mlp.SDF_TWO_PASS = True
            side["one_pass"] = {"ms_per_step": round(dt1 / a.extra_steps * 1e3, 3), "value": round(B_global * H * W * a.extra_steps / dt1 / 1e6, 4), "steps": a.extra_steps,
                                "what": "same iteration with SDF_TWO_PASS = False: every grid row through the three-product fp16-pair kernel (k_h2_fwd<GRID>)"}
        # (2) coverage sensitivity: S2 / R5 / S3 scale with the covered pixels; the headline camera leaves 86 % of the frame empty
        near = [workload.make_targets(trainer, [(it * B_global + v) % 72 for v in shard.local_views(B_global)], (H, W), radius=1.4) for it in range(2)]
        # new tensor sizes (3 x the covered pixels: GBs of ray records, bin scratch): the headline run's cached blocks are returned first and the new
        # sizes settle in 8 warm-up steps; two timed batches, BOTH printed.  (A first-size hipMalloc inside a batch made this line read 50 ms for one
        # batch on the driver's box in round 5; the coverage-dependent buffers are now allocated at padded sizes, optixutils._padded.)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        dt2, _ = timed(a.schedule_it, 8, a.extra_steps, tgts=near)
        dt2b, _ = timed(a.schedule_it + 8 + a.extra_steps, 0, a.extra_steps, tgts=near)
        dt2_all = [round(dt2 / a.extra_steps * 1e3, 3), round(dt2b / a.extra_steps * 1e3, 3)]
        dt2 = max(dt2, dt2b)          # the slower batch is the headline of this side record (both are printed); round 5 reported the faster one
        cov2 = _ou.last_covered_pixels
        side["coverage_sensitivity"] = {"camera_radius": 1.4, "ms_per_step": round(dt2 / a.extra_steps * 1e3, 3), "value": round(B_global * H * W * a.extra_steps / dt2 / 1e6, 4),
                                        "covered_pixels_per_rank": cov2, "coverage": None if cov2 is None else round(cov2 / (B_local * H * W), 4), "steps": a.extra_steps,
                                        "batches_ms_per_step": dt2_all}
        torch.cuda.empty_cache()
        # (3) BASELINE.json configs[3] (8 GPUs x 1 view = global batch 8) beside the weak-scaling line of a plain `--gpus 8`
        if world == 8 and a.global_batch is None:
            one = [workload.make_targets(trainer, [(it * 8 + v) % 72 for v in shard.local_views(8)], (H, W), radius=a.camera_radius) for it in range(2)]
            dt3, _ = timed(a.schedule_it, 4, a.extra_steps, tgts=one, gb=8)
            side["configs3_global_batch_8"] = {"ms_per_step": round(dt3 / a.extra_steps * 1e3, 3), "value": round(8 * H * W * a.extra_steps / dt3 / 1e6, 4),
                                               "iters_per_sec": round(a.extra_steps / dt3, 4), "scaling": "strong", "global_batch": 8, "views_per_gpu": 1, "steps": a.extra_steps,
                                               "what": f"BASELINE.json configs[3] partitioning: tet-res{a.res}, global batch 8 over 8 GPUs, 1 view of {H}^2 per GPU"}
    if a.extra_steps > 0 and world == 1 and a.geometry == "tets" and a.res == 256 and not a.no_reference_config:
        # (4) the reference's OWN published workload (configs/deepfashion_mc_256.json:7-8,17,19): batch 2 x 1024^2, n_samples 24 (1152 shadow rays per
        # covered pixel and pass, kernel.cu:490-529), grid 256 -- on the same trainer / mesh, steady-state schedule
        try:
            side["reference_config"] = reference_config_run(trainer, timed, a, shard)
        except Exception as e:           # pragma: no cover
            side["reference_config"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if _mlp.FALLBACKS:
        raise SystemExit(f"bench.py: the SDF network left the HIP kernels during the run ({_mlp.FALLBACKS}); no number is reported for a torch path")
    if rank == 0:
        g = trainer.geometry
        N, Ftets = g.verts.shape[0], g.indices.shape[0]
        V_aug, T = g.last_mesh_sizes            # mesh of the last timed step (a getMesh here would be a one-rank collective)
        ms = dt / a.steps * 1e3
        mpix = B_global * H * W * a.steps / dt / 1e6
        out = {
            "metric": (f"train iters/sec + rendered Mpixels/sec, tet-res{a.res} @{H}², batch={B_global}" if a.geometry == "tets" else
                       f"train iters/sec + rendered Mpixels/sec, G-FlexiCubes res{a.res} @{H}², batch={B_global}"),
            "value": round(mpix, 4), "unit": "Mpixels/s", "iters_per_sec": round(a.steps / dt, 4),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "strong" if a.global_batch is not None else "weak",
            "vs_baseline": None,
            # what the run actually computed in (from the entry points that were timed), not what the geometry's name suggests
            "dtype": ("f32 (SDF-network GEMMs: fp16-pair operands = 2^-22, fp32 accumulate" +
                      ("; its full-grid forward first as a ONE-product fp16 pass, the rows whose value can matter + an audit sample of all rows re-evaluated "
                       f"with the pair arithmetic; sign margin used by the first pass {margin_used}" if two_pass_ran else "; one pass over every row") + ")")
                     if ("gs_sdf_mlp_fwd_h1" in op_times or "gs_sdf_mlp_fwd_h2" in op_times) else "f32",
            "data": "synthetic",
            **({"build_flags": build_flags} if build_flags else {}),
            "config": {"workload": f"{'G-FlexiCubes res' if a.geometry == 'flexicubes' else 'tet-res'}{a.res} ({'voxel grid' if a.geometry == 'flexicubes' else 'BCC'} {N} verts / {Ftets} cells), {B_local} views/GPU x {H}x{W}, n_samples={a.n_samples} "
                                   f"({2 * a.n_samples ** 2} shadow rays/px/pass), full train iteration fwd+bwd+3xAdam",
                       "global_batch": B_global, "views_per_gpu": B_local, "schedule_it": a.schedule_it,
                       "bilateral_radius": (2 * math.ceil(2.5 * trainer.denoiser.sigma) + 1) if trainer.denoiser else None,
                       "shadow_scale": min((trainer.it - 1) / 1000, 1.0),
                       "mesh": {"V_aug": V_aug, "T": T}, "parallelism": trainer.parallelism()},
        }
        if early is not None:
            out["early_schedule"] = early
        out.update(side)
        out["config"]["covered_pixels_per_rank"] = covered_main
        out["config"]["coverage"] = None if covered_main is None else round(covered_main / (B_local * H * W), 4)
        out["config"]["camera_radius"] = a.camera_radius
        if same_device and world > 1:
            out["data"] = "synthetic; PLUMBING TEST: all ranks share one device over gloo (GSHELL_BENCH_SAME_DEVICE=1) -- not a performance number"
        # rows THIS rank pushes through the SDF-network kernel (the grid rows are sharded over the ranks of a multi-GPU job)
        N_mlp = -(-N // world) if (world > 1 and getattr(trainer.FLAGS, 'shard_mlp_rows', False)) else N
        _ou.last_covered_pixels = covered_main          # the side measurements above ran other frames: the records describe the timed region
        _mlp.LAST_CHAIN_ROWS.clear()
        _mlp.LAST_CHAIN_ROWS.update(rows_main)
        roofs = rooflines(op_times, N_mlp, Ftets, V_aug, T, B_local, H, W, a.n_samples, trainer)
        if roofs:
            out["roofline"] = roofs[0]                      # the dominant hand-written kernel family by HIP-event time
            out["roofline_others"] = roofs[1:]
        if a.op_times:
            out["hbm_kernels"] = hbm_kernels(op_times, N, Ftets, V_aug, T, B_local, H, W)
        if a.op_times:
            out["op_ms"] = {k: round(v["ms"], 4) for k, v in sorted(op_times.items(), key=lambda kv: -kv[1]["ms"] * kv[1]["n"])}
            out["op_calls_per_step"] = {k: v["n"] / a.steps for k, v in op_times.items()}
        if not a.no_cpu_baseline and world == 1:      # reported at N=1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(a.res if a.geometry == "tets" else 256)
            try:
                out["cpu_baseline"].setdefault("stages", {})["env_shade_reference_kernel"] = cpu_reference_env_shade(trainer, a, H, W, op_times, B_local)
            except Exception as e:           # pragma: no cover
                out["cpu_baseline"].setdefault("stages", {})["env_shade_reference_kernel"] = {"error": f"{type(e).__name__}: {e}"}
            if a.geometry == "tets" and a.res == 256:
                out["cpu_baseline"]["headline_cpu_estimate"] = headline_cpu_estimate(out["cpu_baseline"], B_local, H, W)
            ref = gpu_reference_formulation(trainer)
            if ref:
                out["gpu_reference_formulation"] = ref
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def reference_config_run(trainer, timed, a, shard, res=1024, views=2, n=24):
    """One side record: the reference's default workload (configs/deepfashion_mc_256.json: batch 2, train_res 1024 x 1024, n_samples 24) on this
    trainer.  The shader's per-sample records (40 B per ray) are ~14 GB at this size.  Headline fields: through ONE scratch of 2 GiB in chunks of covered
    pixels, forward and backward (gs_env_shade_*_bounded: bit-identical results, tests/test_shade_gpu.py) -- what a frame above optixutils.SCRATCH_BOUND
    (16 GiB by default) takes.  `unbounded`: the same frames with the records kept, which is what the default bound does at this size."""
    from gshell_amd import workload
    from gshell_amd.render import optixutils as ou
    F = trainer.FLAGS
    old = (F.n_samples, list(F.train_res), F.batch)
    out = {"workload": f"reference configs/deepfashion_mc_256.json: tet-res256, batch {views} x {res}x{res}, n_samples={n} ({2 * n * n} shadow rays/px/pass)", "steps": a.extra_steps}
    try:
        F.n_samples, F.train_res, F.batch = n, [res, res], views
        tg = [workload.make_targets(trainer, [(it * views + v) % 72 for v in shard.local_views(views)], (res, res), radius=a.camera_radius) for it in range(2)]
        for tag, bound in (("bounded", 2 << 30), ("unbounded", None)):
            keep, ou.SCRATCH_BOUND = ou.SCRATCH_BOUND, bound
            try:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats()
                timed(a.schedule_it, 3, 1, tgts=tg, gb=views)                    # first-size allocations settle
                dt, _ = timed(a.schedule_it + 4, 0, a.extra_steps, tgts=tg, gb=views)
                cov = ou.last_covered_pixels
                rec = {"ms_per_step": round(dt / a.extra_steps * 1e3, 3), "value": round(views * res * res * a.extra_steps / dt / 1e6, 4), "unit": "Mpixels/s",
                       "iters_per_sec": round(a.extra_steps / dt, 4), "covered_pixels": cov, "coverage": None if cov is None else round(cov / (views * res * res), 4),
                       "shadow_rays_per_pass": None if cov is None else cov * 2 * n * n, "sample_record_bytes_unbounded": None if cov is None else cov * 2 * n * n * 40,
                       "scratch_bound_bytes": bound, "took_bounded_path": bool(ou.last_bounded),
                       "peak_torch_allocated_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 3), "peak_torch_reserved_GB": round(torch.cuda.max_memory_reserved() / 2 ** 30, 3),
                       "bvh_bytes": int(trainer.geometry.optix_ctx.info().get("bytes", 0))}
                if tag == "bounded":
                    out.update(rec)
                else:
                    out["unbounded"] = rec
            except Exception as e:       # pragma: no cover
                (out if tag == "bounded" else out.setdefault("unbounded", {}))["error"] = f"{type(e).__name__}: {e}"
            finally:
                ou.SCRATCH_BOUND = keep
    finally:
        F.n_samples, F.train_res, F.batch = old
        ou.random_perm(old[0], trainer.geometry.verts.device)
    return out


ROOFLINE_CANDIDATES = {"gs_env_shade_fwd", "gs_env_shade_fwd_bounded", "gs_env_shade_bwd", "gs_sdf_mlp_fwd_h1", "gs_sdf_mlp_fwd_h2", "gs_sdf_mlp_fwd", "gs_sdf_mlp_h2_refine_rows", "gs_mtets_flag_refine_rows",
                       "gs_env_shade_bwd_saved", "gs_hashgrid_encode_bwd", "gs_hashgrid_encode_bwd_binned", "gs_sdf_mlp_h2_wgrad", "gs_sdf_mlp_h2_bwd",
                       "gs_sdf_mlp_h2_save_fwd", "gs_sdf_eikonal_rr_fwd", "gs_sdf_eikonal_rr_bwd", "gs_flexi_vd_bwd", "gs_flexi_vd_fwd"}


EVIDENCE_ROUND = "r06"          # bench.py reads ONLY this round's PMC files (profiles/r06_*, written by tools/collect_r06.sh + tools/assemble_r06.py)


def _evidence(name):
    try:
        with open(os.path.join(ROOT, "profiles", f"{EVIDENCE_ROUND}_{name}")) as f:
            return json.load(f)
    except Exception:
        return None


def source_hash():
    """sha256 over the product's sources (gshell_amd/**/*.py|hip|hpp, include/*.h): what the PMC evidence is stamped with (tools/assemble_r06.py) and
    compared against here -- a commit id would go stale the moment the evidence itself is committed."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for base in ("gshell_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            dn[:] = [d for d in dn if d not in ("__pycache__", "lib", ".pytest_cache")]
            files += [os.path.join(dp, f) for f in fn if f.endswith((".py", ".hip", ".hpp", ".h")) or f == "Makefile"]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from THIS round's committed PMC passes (profiles/r06_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate runs, bytes = 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950) -- None if not measured."""
    d = _evidence("pmc_traffic.json")
    try:
        return float(d[kernel]["bytes"])
    except Exception:
        return None


def algorithmic_bytes(N, Ftets, V_aug, T, B, H, W):
    """Algorithmic HBM bytes per launch of the HBM-bound kernel families (DESIGN.md "Kernels": inputs read once + outputs
    written once; E = 16 M edges at res 256 is folded into the per-tet / per-vertex figures of the extraction)."""
    npix = B * H * W
    return {
        "gs_env_shade_fwd": npix * 92, "gs_env_shade_bwd": npix * 140,          # 68 in + 24 out; 92 in + 48 out
        "gs_flexi_vd_fwd": 116 * Ftets + 20 * N, "gs_flexi_vd_bwd": 116 * Ftets + 20 * N,      # 32 B/cube + 84 B/cube weights + 20 B/vertex
        "gs_mtets_count": 16 * Ftets + 20 * N, "gs_mtets_fill": 16 * Ftets + 20 * N + 20 * V_aug + 12 * T,
        "gs_bilateral_fwd_masked": npix * (12 + 12 + 8 + 16), "gs_bilateral_bwd_masked": npix * (12 + 8 + 16 + 12),
        "gs_hashgrid_encode_fwd": 2 * npix * (12 + 4 + 128), "gs_hashgrid_encode_bwd": 2 * npix * (12 + 4 + 128 + 12),
        "gs_hashgrid_encode_bwd_binned": 2 * npix * (12 + 4 + 128 + 12),
        "gs_rasterize_fwd": B * (16 * V_aug + 12 * T) + npix * 40, "gs_aa_apply_fwd": npix * 8 * 45, "gs_aa_apply_bwd": npix * 12 * 45,
        "gs_sdf_reg_fwd": 8 * int(1.19 * Ftets) + 4 * N, "gs_texmlp_fwd_level_major": 2 * npix * 152, "gs_texmlp_bwd_level_major": 2 * npix * (152 + 128 + 24),
        "gs_frame_sums_fwd": npix * 4 * 49, "gs_frame_sums_bwd": npix * 8 * 49,
        "gs_interpolate_fwd": npix * (16 + 4 * 7), "gs_auto_normals_fwd": 36 * T + 24 * V_aug,
        "gs_shade_assemble_fwd": npix * 4 * (4 + 6 + 6 + 3 + 3 + 1 + 3 + 3 + 2 + 4 + 4 + 1 + 3 + 45),
        "gs_shade_assemble_bwd": npix * 4 * (4 + 6 + 6 + 3 + 3 + 1 + 3 + 3 + 2 + 4 + 4 + 1 + 45 + 6 + 6 + 3 + 3 + 3 + 3 + 4 + 4 + 1),
    }


def hbm_kernels(op_times, N, Ftets, V_aug, T, B, H, W):
    """Achieved algorithmic GB/s of every HBM-bound kernel family of the iteration (HIP events around the C-ABI launch),
    as a fraction of the 8 TB/s HBM3E peak.  Ray traversal, hash-grid gathers and the bilateral taps are latency / ALU /
    L2 bound: their fractions are reported, not targeted."""
    out = []
    # covered pixels only (their byte counts depend on the coverage) / ALU- and atomic-bound stages: not HBM rooflines
    masked = ("gs_env_shade_fwd", "gs_env_shade_bwd", "gs_hashgrid_encode_fwd", "gs_hashgrid_encode_bwd", "gs_texmlp_fwd_level_major",
              "gs_texmlp_bwd_level_major", "gs_bilateral_fwd_masked", "gs_bilateral_bwd_masked")
    for name, alg in algorithmic_bytes(N, Ftets, V_aug, T, B, H, W).items():
        if name in masked:
            continue
        rec = op_times.get(name)
        if rec and rec["ms"] > 0:
            gbps = alg / (rec["ms"] * 1e-3) / 1e9
            out.append({"op": name, "ms": round(rec["ms"], 4), "GB/s": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4)})
    return sorted(out, key=lambda r: -r["ms"])


def rooflines(op_times, N, Ftets, V_aug, T, B, H, W, n, trainer):
    """Roofline records of the hand-written kernel families, the dominant one (by HIP-event time per step) first."""
    if not op_times:
        return []
    fams = sorted(op_times.items(), key=lambda kv: -kv[1]["ms"] * kv[1]["n"])
    recs = []
    for name, rec in fams[:5]:
        r = roofline_of(name, rec, op_times, N, Ftets, V_aug, T, B, H, W, n, trainer)
        if r:
            recs.append(r)
    return recs


def binding_metric(key):
    """What this round's rocprofv3 --pmc passes say binds a kernel family: the resource and its measured utilisation (profiles/r06_binding.json,
    written by tools/assemble_r06.py).  The record names the hash of the product sources it was collected on; `stale_sources_now` is set when they have
    changed since -- kernel times in the line are live, the utilisation figures are then those of the named source state."""
    d = _evidence("binding.json")
    if not d or key not in d:
        return None
    b = dict(d[key])
    at = (d.get("_meta") or {}).get("source_hash")
    b["collected_at_source_hash"] = at
    now = source_hash()
    if at and at != now:
        b["stale_sources_now"] = now
    return b


def roofline_of(name, rec, op_times, N, Ftets, V_aug, T, B, H, W, n, trainer):
    """One record per kernel family.  `achieved` = ALGORITHMIC work per launch exactly as SURVEY.md 8(d) defines it (no implementation
    buffers) / the HIP-event average of the launch; `bound` / `peak` = the contract's roofline for that work; `binding` = the resource the
    PMC counters show the family is actually limited by, with its measured utilisation (a ray traversal is VALU-issue bound: its HBM
    fraction is reported because the contract has no third kind, and is not the figure of merit)."""
    from gshell_amd.geometry import mlp as _mlp
    from gshell_amd.render import optixutils as _ou
    npix = B * H * W
    ms = rec["ms"]
    t = ms * 1e-3

    def hbm(kernel, alg_bytes, pmc_key=None, **extra):
        gbps = alg_bytes / t / 1e9
        out = {"kernel": kernel, "bound": "hbm", "achieved": round(gbps, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 5),
               "traffic": pmc_traffic(pmc_key or name) if N == 2282489 else None, "avg_launch_ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes)}
        if out["traffic"]:
            out["traffic_over_algorithmic"] = round(out["traffic"] / alg_bytes, 2)        # wasted re-reads / implementation buffers, per launch
        out.update(extra)
        b = binding_metric(name)
        if b:
            out["binding"] = b
        return out

    def mfma(kernel, flops, peak, pmc_key=None, **extra):
        tf = flops / t / 1e12
        out = {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
               "traffic": pmc_traffic(pmc_key or name) if N == 2282489 else None, "avg_launch_ms": round(ms, 4), "algorithmic_flops": flops}
        out.update(extra)
        b = binding_metric(name)
        if b:
            out["binding"] = b
        return out

    if name == "gs_sdf_mlp_fwd":
        # fp32 MFMA roofline: 2 * (39*256 + 5*256*256 + 295*256 + 256) = 826 880 flop per grid vertex
        return mfma("k_sdf_mlp_fwd (gs_sdf_mlp_fwd)", 826880.0 * N, 157.3, "k_sdf_mlp_fwd",
                    note="fp32-in/fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32); HBM traffic is 16 B/vertex by construction")
    if name in ("gs_sdf_mlp_fwd_h2", "gs_sdf_mlp_fwd_h1"):
        # 826 880 ALGORITHMIC flop per grid vertex (what the network defines) over K padded to 16 (48 + 5 x 256 + 304 input columns x
        # 256 outputs); h2 executes 3 MFMA products per algorithmic product, h1 (first pass of the two-pass forward) one
        prods = 3 if name.endswith("h2") else 1
        flops = 826880.0 * N
        executed = 2.0 * 256 * (48 + 5 * 256 + 304) * prods * N
        out = mfma("k_h2_fwd<GRID> (gs_sdf_mlp_fwd_h2)" if prods == 3 else "k_h1_fwd (gs_sdf_mlp_fwd_h1)", flops, 2500.0, "k_h2_fwd" if prods == 3 else "k_h1_fwd",
                   rows_per_launch=int(N), executed_mfma_flops=executed, executed_TFLOPs=round(executed / t / 1e12, 1),
                   executed_frac_of_f16_peak=round(executed / t / 1e12 / 2500.0, 4),
                   note="v_mfma_f32_32x32x16_f16, fp32 accumulate; " + ("operands = fp16 pairs (2^-22): three products per algorithmic product" if prods == 3 else
                        "ONE fp16 product per algorithmic product over every grid row; the rows whose value can matter are re-evaluated by gs_sdf_mlp_h2_refine_rows"))
        if prods == 1:
            ref = op_times.get("gs_sdf_mlp_h2_refine_rows", {"ms": 0.0})["ms"] + op_times.get("gs_mtets_flag_refine_rows", {"ms": 0.0})["ms"]
            net = trainer.geometry.sdf_net if hasattr(trainer.geometry, "sdf_net") else None
            out["two_pass"] = {"refine_ms": round(ref, 4), "whole_forward_ms": round(ms + ref, 4),
                               "whole_forward_algorithmic_TFLOPs": round(flops / ((ms + ref) * 1e-3) / 1e12, 1),
                               "max_dev_one_product_on_reevaluated_rows": None if net is None else net.__dict__.get("_gs_two_pass_maxdev"),
                               "sign_margin_used_incl_audit_rows": None if net is None else net.__dict__.get("_gs_two_pass_margin_used")}
        return out
    if name in ("gs_sdf_mlp_h2_bwd", "gs_sdf_mlp_h2_wgrad", "gs_sdf_mlp_h2_save_fwd"):
        # dense backward = 2 x forward: 826 880 flop per (virtual) row for the reverse chain G = W^T D and the same again for
        # the weight gradient D^T X; the saved forward = one forward.  Per LAUNCH: the mean over the grid pass (rows with gradient) and the
        # eikonal pass (4 virtual rows per surface sample), which are the two launches of each kernel in an iteration.
        rows = _mlp.LAST_CHAIN_ROWS
        launches = max((1 if rows.get(1, 0) else 0) + (1 if rows.get(2, 0) else 0), 1)      # the eikonal term has its own entry points in its default (reverse) formulation
        per_launch = (rows.get(1, 0) + 4 * rows.get(2, 0)) / launches
        flops = 826880.0 * per_launch
        kern = {"gs_sdf_mlp_h2_bwd": "k_h2_bwd<ROWS|EIK> reverse chain", "gs_sdf_mlp_h2_wgrad": "k_h2_wgrad16 (+ k_h2_wgrad for the output layer)",
                "gs_sdf_mlp_h2_save_fwd": "k_h2_fwd<ROWS|EIK> recompute + saved planes"}[name]
        return mfma(kern, flops, 2500.0, rows_with_gradient=int(rows.get(1, 0)), eikonal_samples=int(rows.get(2, 0)), mean_virtual_rows_per_launch=int(per_launch),
                    hbm_plane_bytes_per_launch=int(per_launch * 7 * 256 * 4),
                    note="fp16-pair (reverse chain, saved forward) / bf16-pair (weight gradient) operands: three products per algorithmic product")
    if name in ("gs_sdf_eikonal_rr_fwd", "gs_sdf_eikonal_rr_bwd"):
        # the eikonal term by reverse over reverse (the reference's autograd formulation, gshell_tets_geometry.py:302-324), per sample:
        #   _fwd = value pass + reverse chain to grad_x f                                   = 2 x 826 880 flop
        #   _bwd = tangent pass + reverse chain with source + two outer products per layer    = 4 x 826 880 flop
        ns = _mlp.LAST_CHAIN_ROWS.get(4, 0)
        fwd = name.endswith("_fwd")
        flops = (2 if fwd else 4) * 826880.0 * ns
        return mfma("gs_sdf_eikonal_rr_fwd (k_h2_fwd<ROWS> + k_h2_bwd<ROWS>)" if fwd else "gs_sdf_eikonal_rr_bwd (k_h2_fwd<RR> + k_h2_bwd<RR> + k_h2_wgrad16)",
                    flops, 2500.0, eikonal_samples=int(ns), hbm_plane_bytes_per_launch=int(ns * 7 * 256 * 4 * (3 if fwd else 11)),
                    note="fp16-pair / bf16-pair operands: three products per algorithmic product; bound by the fp32 planes it writes and reads "
                         "(3 plane passes in _fwd, 11 in _bwd)")
    if name == "gs_env_shade_fwd":
        # 2 n^2 shadow rays per covered pixel through a software BVH: the family is VALU-issue bound (this round's SQ counters of k_shade_trace, `binding`),
        # so its roofline is the vector ALU -- achieved = the fraction of VALU lane-cycles doing useful work = issue-busy x useful-lane fraction, measured
        # by rocprofv3 --pmc on this workload; the live figure of merit is rays / s.  The HBM record SURVEY.md 8(d) defines for the stage (68 B/px in +
        # 24 B/px out) is carried as `hbm`: a fraction of a per cent, because bytes are not what the stage is short of.
        n_cov = _ou.last_covered_pixels
        rays = None if n_cov is None else n_cov * 2 * n * n
        h = hbm("gs_env_shade_fwd (k_shade_samples + k_shade_trace + k_shade_accumulate)", npix * 92.0)
        b = h.pop("binding", None)
        busy, useful = (b or {}).get("valu_issue_busy"), (b or {}).get("useful_valu_lane_fraction")
        frac = None if (busy is None or useful is None) else round(busy * useful, 4)
        out = {"kernel": h["kernel"], "bound": "valu", "achieved": frac, "peak": 1.0, "unit": "fraction of VALU lane-cycles doing useful work (issue-busy x useful lanes, rocprofv3 --pmc)",
               "frac": frac, "valu_issue_busy": busy, "useful_valu_lane_fraction": useful, "avg_launch_ms": h["avg_launch_ms"], "covered_pixels": n_cov, "shadow_rays": rays,
               "rays_per_s": None if not rays else round(rays / t / 1e9, 3), "traffic": h.get("traffic"),
               "hbm": {k: h.get(k) for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes", "traffic", "traffic_over_algorithmic")},
               "note": "software BVH any-hit traversal; rays_per_s = G rays / s over the whole family (live); the VALU fractions are this round's counters"}
        if b:
            out["binding"] = b
        return out
    if name in ("gs_env_shade_bwd_saved", "gs_env_shade_bwd"):
        return hbm("gs_env_shade_bwd_saved (k_shade_grad + light-gradient counting sort)", npix * 140.0, covered_pixels=_ou.last_covered_pixels,
                   note="SURVEY.md 8d: 92 B/px in + 48 B/px out; streams the forward pass's saved 16 B/ray records")
    alg = algorithmic_bytes(N, Ftets, V_aug, T, B, H, W).get(name)
    if alg is None:
        return None
    return hbm(name, alg)


def cpu_baseline(res=256):
    """CPU numbers of the reference formulation on this box's host cores (reported, not the target):
      value    end-to-end forward + backward of the oracle pipeline on BASELINE.json configs[0] -- tet-res64 (BCC 26: 202 800 tets),
               1 view 256 x 256, 1 MC light sample (2 shadow rays / pixel), constant kd / ks -- the reference's own CPU-runnable
               case.  ~15-20 s of CPU work.  The headline config itself (20 M shadow rays x 2.3 10^5 triangles
               per pass, brute force) is infeasible on a CPU.
      stages   the stages whose reference formulation DOES run at the HEADLINE size on a CPU: the SDF network over all grid rows
               (geometry/mlp.py module, torch CPU) and the G-MarchingTets extraction (torch.unique per call + gathers, as
               geometry/gshell_tets.py:266-276 does) on the tet-res256 grid."""
    try:
        from oracle import pipeline_oracle
    except Exception as e:           # pragma: no cover
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    out = pipeline_oracle.timed_sample(cells=26, res=(256, 256), n_samples=1, constant_kd=True)
    out["config"] = "BASELINE.json configs[0]: tet-res64, 1 view 256x256, 1 MC light sample, constant kd"
    try:
        out["stages"] = cpu_stage_times(res)
    except Exception as e:           # pragma: no cover
        out["stages"] = {"error": str(e)}
    return out


def headline_cpu_estimate(cpu, views, H, W):
    """What the per-stage CPU numbers imply for the HEADLINE config: the stages that were timed at configs[2]'s size on this box,
    summed -- a LOWER bound of one CPU iteration (the SDF network's backward, rasterisation, texture, antialiasing, denoiser and losses are not in it) --
    and, beside it, the one complete measurement that exists: the oracle chain of configs[2] (network -> extraction -> render -> tick -> backward, float32)
    as minted on the build container's 8 cores (profiles/r06_mint_chains.log)."""
    st = (cpu or {}).get("stages") or {}
    parts = {}
    if "sdf_mlp_fwd" in st:
        parts["sdf_mlp_fwd"] = st["sdf_mlp_fwd"]["s"]
    if "extraction_fwd" in st:
        parts["extraction_fwd"] = st["extraction_fwd"]["s"]
    es = st.get("env_shade_reference_kernel") or {}
    if "s_fwd_launch" in es:
        parts["env_shade_reference_kernel_fwd_bwd_all_views"] = round((es["s_fwd_launch"] + es["s_bwd_launch"]) * views, 3)
    out = {"config": "BASELINE.json configs[2]: tet-res256, 4 views 512x512, n_samples 8", "stages_summed_s": parts}
    if parts:
        tot = sum(parts.values())
        out.update(s_per_iteration_lower_bound=round(tot, 2), Mpixels_per_s_upper_bound=round(views * H * W / tot / 1e6, 5), cores=st.get("cores"))
    try:
        import re
        log = open(os.path.join(ROOT, "profiles", f"{EVIDENCE_ROUND}_mint_chains.log")).read()
        sec = log[log.index("chain config2"):]
        m = re.search(r"\[torch\.float32\] network Jacobian on \d+ rows: (\d+) s", sec)
        if m:
            out["full_oracle_chain_float32"] = {"s": int(m.group(1)), "Mpixels_per_s": round(4 * 512 * 512 / int(m.group(1)) / 1e6, 5), "cores": 8,
                                                "what": "oracle/make_golden_chain.py config2, float32 run (one whole tick + backward), build container"}
    except Exception:
        pass
    return out


def cpu_stage_times(res):
    import numpy as np
    from gshell_amd import grid
    from gshell_amd.geometry.mlp import MLP
    from oracle import fields, mtets_oracle
    st = {"cores": torch.get_num_threads()}
    verts, tets = grid.grid_for_res(res, device="cpu")
    torch.manual_seed(0)
    net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3])
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(0, verts.shape[0], 1 << 17):
            net(verts[i:i + (1 << 17)])
        st["sdf_mlp_fwd"] = {"s": round(time.perf_counter() - t0, 3), "rows": int(verts.shape[0]), "what": "geometry/mlp.py module, torch CPU fp32"}
    vn = verts.numpy()
    pos = vn + fields.make_deform(vn, 1.0 / {64: 26, 128: 52, 256: 104}[res], 3)
    sdf, msdf = fields.make_sdf(pos, "skirt", 3).astype(np.float32), fields.make_msdf(pos, "wavy", 3).astype(np.float32)
    t0 = time.perf_counter()
    topo = mtets_oracle.build_topology(tets)                    # the reference sorts / uniques the crossing edges on every call
    ex = mtets_oracle.extract(torch.tensor(pos.astype(np.float32)), torch.tensor(sdf), torch.tensor(msdf), tets, topo=topo, with_tangents=False)
    st["extraction_fwd"] = {"s": round(time.perf_counter() - t0, 3), "tets": int(tets.shape[0]), "faces": int(ex["faces_aug"].shape[0]),
                            "what": "oracle/mtets_oracle (edge unique + gathers, the formulation of geometry/gshell_tets.py:245-443), torch CPU"}
    return st


def cpu_reference_env_shade(trainer, a, H, W, op_times, B_local):
    """The dominant kernel family on the host, by the REFERENCE'S OWN CODE (`kind: reference`): render/optixutils/c_src/envsampling/kernel.cu
    compiled for the host (oracle/_ref/ref_envshade.so, built by oracle/Makefile from the reference tree; OpenMP over pixels, optixTrace answered
    by oracle/anyhit_grid.h) on ONE view of the headline workload -- this run's mesh, its 512 x 512 g-buffer made by the product's stages, the same
    probe, n_samples -- forward (`backward = 0`) and gradient launch (`backward = 1`), as optixutils/ops.py:81-108 issues them."""
    import numpy as np
    from oracle import refnative as rn
    if not rn.available("ref_envshade"):
        return {"error": "oracle/_ref/ref_envshade.so not built (needs /root/reference at build time)"}
    from gshell_amd import workload
    from gshell_amd.render import rast as dr, renderutils as ru
    n = a.n_samples
    dev = trainer.geometry.verts.device
    with torch.no_grad():
        m = trainer.geometry.getMesh(trainer.mat)['imesh']
        mvp, campos = workload.views([0], dev, radius=a.camera_radius)
        tri = m.faces_i32().contiguous()
        rast, _ = dr.rasterize(None, ru.xfm_points(m.v_pos[None], mvp), tri, (H, W))
        gb_pos, gb_nrm_s = dr.interpolate_groups([m.v_pos.contiguous(), m.v_nrm.contiguous()], rast, tri)
        gb_geo = dr.face_normals(m.v_pos, tri, rast)
        view = campos[:, None, None, :].contiguous()
        tng = torch.cross(torch.nn.functional.normalize(torch.randn_like(gb_nrm_s), dim=-1), gb_nrm_s, dim=-1)
        gb_nrm = ru.prepare_shading_normal(gb_pos, view, None, gb_nrm_s, tng, gb_geo, two_sided_shading=True, opengl=True).contiguous()
        mask = (rast[..., 3] > 0).float()
        tex = trainer.mat['kd_ks'].sample(gb_pos)
        ro = (gb_pos + gb_nrm * 0.001).contiguous()
        trainer.lgt.update_pdf()
        lg = trainer.lgt
        g = [mask, ro, gb_pos.contiguous(), gb_nrm, view, tex[..., 0:3].contiguous(), tex[..., 3:6].contiguous(), lg.base.detach(), lg._pdf, lg.rows[:, 0].contiguous(), lg.cols]
        g = [t.detach().float().cpu().numpy() for t in g]
        verts, tris = m.v_pos.detach().float().cpu().numpy(), tri.cpu().numpy()
    gen = torch.Generator().manual_seed(21)
    perms = torch.argsort(torch.rand(256, n * n, generator=gen), dim=-1).int().numpy()
    tail = (perms, 0, n, 4242, 1.0, verts, tris)
    rn.set_threads(0)
    rn.set_anyhit_mode(True)
    t0 = time.perf_counter()
    rn.env_shade_fwd(*g, *tail)
    t_f = time.perf_counter() - t0
    dg = torch.rand(2, 1, H, W, 3, generator=gen).numpy()
    t0 = time.perf_counter()
    rn.env_shade_bwd(*g, *tail, dg[0], dg[1])
    t_b = time.perf_counter() - t0
    cov = int((g[0] > 0).sum())
    out = {"kind": "reference", "what": "kernel.cu / bsdf.h compiled for the host (oracle/_ref/ref_envshade.so), OpenMP over pixels, grid-filtered any-hit with the kernels' fp32 predicate",
           "sample": f"1 of the {B_local} views, {H}x{W}, n_samples={n} ({2 * n * n} shadow rays / covered pixel / launch), {int(tris.shape[0])} triangles, {cov} covered pixels",
           "cores": int(os.cpu_count() or 1), "s_fwd_launch": round(t_f, 3), "s_bwd_launch": round(t_b, 3), "covered_Mpixels_per_s_fwd": round(cov / t_f / 1e6, 4)}
    f, b = (op_times or {}).get("gs_env_shade_fwd"), (op_times or {}).get("gs_env_shade_bwd_saved")
    if f and b:
        out["gpu_ms_per_view_fwd_bwd"] = [round(f["ms"] / B_local, 4), round(b["ms"] / B_local, 4)]
    return out


def gpu_reference_formulation(trainer):
    """Same-device baseline: the reference's torch formulation of the extraction (per-call edge unique + gathers,
    oracle/mtets_oracle.py -- pinned bit-exactly to the real geometry/gshell_tets.py) and of the SDF network (geometry/mlp.py
    module through hipBLASLt) on THIS GPU, beside the hand-written kernels' times."""
    from oracle import mtets_oracle
    g = trainer.geometry
    if not hasattr(g, "gshell_tets"):
        return None
    out = {}
    with torch.no_grad():
        v = g.verts + g.max_displacement * g.deform
        sdf = g.sdf_net(v[:1 << 18])            # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sdf = torch.cat([g.sdf_net(v[i:i + (1 << 19)]) for i in range(0, v.shape[0], 1 << 19)])[:, 0]
        torch.cuda.synchronize()
        out["sdf_mlp_fwd_torch_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            topo = mtets_oracle.build_topology(g.indices)
            ex = mtets_oracle.extract(v, sdf, g.msdf.detach(), g.indices, topo=topo, with_tangents=False)
            torch.cuda.synchronize()
            out["extraction_fwd_torch_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        out["faces"] = int(ex["faces_aug"].shape[0])
    return out


if __name__ == "__main__":
    main()
