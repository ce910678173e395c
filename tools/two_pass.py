"""Two-pass SDF forward (one-product pass + three-product refinement, csrc/mlp_h2.hip) on the bench grid: equality of what the
reference consumes with the one-pass h2 kernel, the measured one-product error, the share of refined rows, and timings.  GPU box.

    [GSHELL_HIP_LIB=...] python tools/two_pass.py [res] [tau]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import grid, workload
from gshell_amd.geometry import mlp
from gshell_amd.geometry.gshell_tets import GShell_Tets
from gshell_amd.geometry.mlp import MLP, fused_forward


def build(res, steps=400, dev="cuda"):
    torch.manual_seed(0)
    verts, tets = grid.grid_for_res(res, device=dev)
    verts = ((verts - verts.mean(dim=0)) * 1.4).contiguous()

    class G:
        pass
    g = G()
    g.verts, g.sdf_net = verts, MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).to(dev)
    workload.fit_sdf_net(g, steps=steps)
    topo = GShell_Tets().topology(tets, verts.shape[0])
    return g.sdf_net, verts, topo


def occ_words(topo):
    n = (topo.N + 63) // 64
    out = torch.empty(n, dtype=torch.int64, device=topo.device)
    from gshell_amd import _lib
    _lib.check(_lib.lib().gs_memcpy_d2d(_lib.ptr(out), _lib.c_void_p(topo.occ_bits_ptr()), _lib.c_int64(n * 8), _lib.stream()))
    return out


def compare(net, verts, topo, tau):
    """-> dict of the equalities the two-pass scheme promises (against the one-pass h2 kernel on the same inputs)"""
    mlp.SDF_TWO_PASS_TAU = tau
    with torch.no_grad():
        y1 = fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr())[:, 0].clone()
        occ1 = occ_words(topo).clone()
        y2 = fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr(), refine_topo=topo, defer_status=True)[:, 0]
        st = net.__dict__["_gs_fwd_status"]
        occ2 = occ_words(topo)
        nonfinite, maxdev, margin_used = st.result()
        n_ref = int(st.n_rows[0])
        e = topo.edges().long()
        cross = (y1[e[:, 0]] > 0) != (y1[e[:, 1]] > 0)
        ends = torch.unique(e[cross].reshape(-1))
        bits = ((occ2[:, None] >> torch.arange(64, device=occ2.device)[None]) & 1).reshape(-1)[:topo.N].bool()
    return {
        "rows": int(verts.shape[0]), "tau": tau, "refined_rows": n_ref, "refined_fraction": n_ref / verts.shape[0],
        "crossing_edges": int(cross.sum()), "crossing_edge_end_points": int(ends.numel()),
        "max_abs_dev_one_product_on_refined_rows": maxdev, "nonfinite": nonfinite,
        # fraction of its sign margin max(tau, |sdf|) the first pass's error used up: as MEASURED by the kernel on the refined rows + the audit
        # sample, and the TRUTH over every row the second pass left alone (y2 = first-pass value there, y1 = the one-pass kernel's)
        "margin_used_measured": margin_used,
        "margin_used_all_unrefined_rows": float(((y1 - y2).abs() / torch.clamp(y1.abs(), min=tau)).max()),
        "sign_disagreements": int(((y1 > 0) != (y2 > 0)).sum()), "occupancy_words_equal": bool(torch.equal(occ1, occ2)),
        "sign_bits_match_values": bool(torch.equal(bits, y2 > 0)),
        "end_point_values_bit_identical": bool(torch.equal(y1[ends], y2[ends])),
        "max_abs_diff_all_rows": float((y1 - y2).abs().max()), "rows_bit_identical": int((y1 == y2).sum()),
    }


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def precision_table(res=256, states=(30, 400, 1500)):
    """error of the one-pass h2 kernel and of the one-product pass against float64, and the two-pass equalities, for several
    states of the network (fit steps; the last entry of each row scales the fitted hidden weights by 1.5 -- a sharper network)"""
    rows = []
    for steps in states:
        net, verts, topo = build(res, steps=steps)
        for scale in (1.0, 1.5):
            if scale != 1.0:
                with torch.no_grad():
                    for m in net.net:
                        if isinstance(m, torch.nn.Linear) and m.in_features == 256 and m.out_features == 256:
                            m.weight.mul_(scale ** (1.0 / 5.0))
            with torch.no_grad():
                net64 = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda().double()
                net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
                y64 = torch.cat([net64(verts[i:i + (1 << 18)].double()) for i in range(0, verts.shape[0], 1 << 18)])[:, 0]
                y3 = fused_forward(net, verts, "h2")[:, 0]
                L = __import__("gshell_amd")._lib.lib()
                from gshell_amd import _lib
                packed, n_hidden, skip = mlp.pack_weights_h2(net)
                y1 = torch.empty(verts.shape[0], device=verts.device)
                _lib.check(L.gs_sdf_mlp_fwd_h1(_lib.ptr(verts), _lib.c_int64(verts.shape[0]), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip),
                                               _lib.ptr(y1), _lib.c_void_p(0), _lib.c_void_p(0), _lib.stream()))
            r = compare(net, verts, topo, mlp.SDF_TWO_PASS_TAU)
            rows.append({"fit_steps": steps, "hidden_weight_scale": scale, "max_abs_sdf": float(y64.abs().max()),
                         "h2_one_pass": {"max_abs_err_vs_f64": float((y3.double() - y64).abs().max()), "sign_flips_vs_f64": int(((y3 > 0) != (y64 > 0)).sum())},
                         "h1_one_product": {"max_abs_err_vs_f64": float((y1.double() - y64).abs().max()), "rms_err": float((y1.double() - y64).pow(2).mean().sqrt()),
                                            "sign_flips_vs_f64": int(((y1 > 0) != (y64 > 0)).sum())},
                         "two_pass": {k: r[k] for k in ("refined_fraction", "max_abs_dev_one_product_on_refined_rows", "sign_disagreements", "end_point_values_bit_identical",
                                                        "occupancy_words_equal", "max_abs_diff_all_rows")}})
    return rows


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--states":
        print(json.dumps({"grid": "tet-res256", "tau": mlp.SDF_TWO_PASS_TAU, "safety": mlp.SDF_TWO_PASS_SAFETY, "states": precision_table()}, indent=1))
        return
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tau = float(sys.argv[2]) if len(sys.argv) > 2 else mlp.SDF_TWO_PASS_TAU
    net, verts, topo = build(res)
    out = compare(net, verts, topo, tau)
    from gshell_amd import _lib
    L = _lib.lib()
    with torch.no_grad():
        packed, n_hidden, skip = mlp.pack_weights_h2(net)
        y = torch.empty(verts.shape[0], device=verts.device)

        def h1():
            _lib.check(L.gs_sdf_mlp_fwd_h1(_lib.ptr(verts), _lib.c_int64(verts.shape[0]), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip),
                                           _lib.ptr(y), _lib.c_void_p(0), _lib.c_void_p(0), _lib.stream()))
        out["ms"] = {"one_pass_h2": round(timed(lambda: fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr())), 3),
                     "one_product_kernel_alone": round(timed(h1), 3),
                     "two_pass_total": round(timed(lambda: fused_forward(net, verts, "h2", occ_bits_ptr=topo.occ_bits_ptr(), refine_topo=topo, defer_status=True)), 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
