#!/usr/bin/env python3
"""Benchmark of the hot path: one RGAT layer, forward + backward, on an ogbn-mag-shaped graph.

Metric (BASELINE.json): million edges/s (fwd+bwd) of one RGAT layer, ogbn-mag, feat=64.
A step = layer forward (segment GEMMs, edge softmax, aggregation, self-loop GEMM, bias) plus
``out.backward(grad)`` of the same layer; no optimizer step (the reference folds optimizer.step()
into its backward time, hrt/python/RGNNUtils/RGNNUtils.py:304-311 -- deviation stated here and in
DESIGN.md).  Inputs are synthetic (het_amd/synth.py): no dataset can be downloaded.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs at 2.4 GHz)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--scale", type=float, default=1.0, help="shrink the mag-like graph (1.0 = full ogbn-mag size)")
    p.add_argument("--heads", type=int, default=None, help="default: 4 (RGAT, BASELINE.json configs[2]); 8 for --model hgt (configs[3])")
    p.add_argument("--feat", type=int, default=64)
    p.add_argument("--variant", default="default", choices=["default", "compact", "compact_mulfirst", "mulfirst"],
                   help="reference layer flags: default = per-edge projections (the reference's default flags); "
                        "compact = --compact_as_of_node_flag --compact_direct_indexing_flag")
    p.add_argument("--edge-order", default="src", choices=["src", "random"])
    p.add_argument("--model", default="rgat", choices=["rgat", "rgcn", "hgt"],
                   help="rgat = the BASELINE.json metric; rgcn / hgt time BASELINE.json configs[1] / configs[3] (single GPU)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-variants", action="store_true", help="skip timing the other reference flag combinations")
    p.add_argument("--cpu-scale", type=float, default=0.05, help="graph scale of the CPU-baseline sample")
    a = p.parse_args()
    if a.heads is None:
        a.heads = 8 if a.model == "hgt" else 4
    return a


def layer_flags(variant):
    return dict(compact_as_of_node_flag=variant.startswith("compact"),
                compact_direct_indexing_flag=variant.startswith("compact"),
                multiply_among_weights_first_flag=variant.endswith("mulfirst"))


def gat_bwd_bytes(E, N, H, X):
    """Algorithmic bytes of backward_relational_fused_gat_separate_coo, kind 0:
    per edge reads col,eids (8 B each), el,er,exp (H floats each), feat (X); writes grad_el, grad_er (H),
    grad_feat (X); per node reads sum (H), ret, gradout (X each)."""
    return E * (2 * 8 + 4 * (3 * H + X) + 4 * (2 * H + X)) + N * 4 * (H + 2 * X)


def cpu_baseline(args):
    """The oracle's plain-PyTorch RGAT layer (HET semantics) timed on the host cores, fwd+bwd, on a
    bounded sample (a mag-like graph at --cpu-scale)."""
    from het_amd.graph import HetGraph
    from het_amd.synth import make_mag_like
    from oracle import layers as OL
    g = HetGraph.from_integrated_coo(make_mag_like(scale=args.cpu_scale, edge_order=args.edge_order), full=False)
    s = g.get_separate_coo_original()
    N, R, H, K = g.get_num_nodes(), g.get_num_rels(), args.heads, args.feat
    D = K // H
    torch.manual_seed(0)
    x = (torch.randn(N, K) * 0.1).requires_grad_(True)
    W = (torch.randn(R, H, K, D) * 0.1).requires_grad_(True)
    al = (torch.randn(R, H, D) * 0.1).requires_grad_(True)
    ar = (torch.randn(R, H, D) * 0.1).requires_grad_(True)
    lw = (torch.randn(K, K) * 0.1).requires_grad_(True)
    go = torch.randn(N, K)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        out = OL.rgat_layer(x, W, al, ar, s["rel_ptrs"], s["row_indices"], s["col_indices"], N, 0.2, lw, None)
        torch.autograd.grad(out, [x, W, al, ar, lw], go)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(g.get_num_edges() / med / 1e6, 3), "unit": "million edges/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle/layers.py rgat_layer fwd+bwd (torch CPU fp32, HET cross-relation softmax) on a mag-like "
                      f"graph at scale {args.cpu_scale} ({g.get_num_edges()} edges, {N} nodes), median of 3 after 1 warm-up; "
                      f"os.cpu_count()={os.cpu_count()}"}


def other_variants(args, coo, dev, steps, default_ms, default_value):
    """The same layer under the reference's other flag combinations (identical outputs; its sweep runs them too,
    hrt/utils/_do_all_cases.sh:2-40).  Reported next to the headline, which stays on the default flags."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    res = {"default": {"ms_per_step": round(default_ms, 4), "million_edges_per_s": round(default_value, 2)}}
    g = HetGraph.from_integrated_coo(coo, full=True)
    E, N = coo.num_edges, coo.num_nodes
    flag_names = {"compact": "--compact_as_of_node_flag --compact_direct_indexing_flag",
                  "mulfirst": "--multiply_among_weights_first_flag",
                  "compact_mulfirst": "--compact_as_of_node_flag --compact_direct_indexing_flag --multiply_among_weights_first_flag"}
    for variant in ("compact", "mulfirst", "compact_mulfirst"):
        torch.manual_seed(0)
        layer = HET_RGATLayer(args.feat, args.feat, g.get_num_rels(), args.heads, self_loop=True, dropout=0.0,
                              **layer_flags(variant)).to(dev)
        embed = torch.nn.Parameter(torch.empty(N, args.feat, device=dev))
        torch.nn.init.xavier_uniform_(embed)
        go = torch.randn(N, args.feat, device=dev)

        def step():
            for p in layer.parameters():
                p.grad = None
            embed.grad = None
            layer(g, embed).backward(go)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[variant] = {"ms_per_step": round(dt * 1e3, 4), "million_edges_per_s": round(E / dt / 1e6, 2),
                        "flags": flag_names[variant]}
        del layer, embed, go
    return res


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    # (rehearsals of the multi-rank flow on a one-GPU box: HET_DIST_BACKEND=gloo lets the ranks share the device)
    backend = os.environ.get("HET_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("HET_FORCE_DIST") == "1":
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from het_amd import kernels as HK
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like

    coo = make_mag_like(scale=args.scale, edge_order=args.edge_order)
    E_global, N_global = coo.num_edges, coo.num_nodes
    H, K = args.heads, args.feat
    X = K
    g_rels = coo.num_rels
    torch.manual_seed(0)
    use_dist = world > 1 or os.environ.get("HET_FORCE_DIST") == "1"
    layout_ms = None
    if use_dist:
        from het_amd.dist import DistRGAT
        runner = DistRGAT(coo, K, X, H, dev, **layer_flags(args.variant))
        step = runner.step
        E_local, N_local = runner.num_local_edges, runner.num_local_nodes
    else:
        for f in ("row", "col", "rel", "eids", "node_type_offsets"):
            setattr(coo, f, getattr(coo, f).to(dev))
        torch.cuda.synchronize()
        t_l = time.perf_counter()
        g = HetGraph.from_integrated_coo(coo, full=args.variant.startswith("compact") or args.model == "hgt")
        torch.cuda.synchronize()
        layout_ms = (time.perf_counter() - t_l) * 1e3  # device-side builders (layouts.hip); outside the timed region
        extra = ()
        if args.model == "rgat":
            layer = HET_RGATLayer(K, X, g.get_num_rels(), H, self_loop=True, dropout=0.0, **layer_flags(args.variant)).to(dev)
        elif args.model == "rgcn":
            from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
            layer = HET_EglRelGraphConv_EdgeParallel(K, X, g.get_num_rels(),
                                                     compact_as_of_node_flag=args.variant.startswith("compact"),
                                                     compact_direct_indexing_flag=args.variant.startswith("compact")).to(dev)
            extra = (torch.rand(E_global, 1, device=dev),)  # edge norm, as RGCN.py:530
        else:
            from het_amd.layers import HET_HGTLayerHetero
            layer = HET_HGTLayerHetero(g.get_num_ntypes(), g.get_num_rels(), K, X, num_heads=H, dropout=0.0).to(dev)
        embed = torch.nn.Parameter(torch.empty(N_global, K, device=dev))
        torch.nn.init.xavier_uniform_(embed)
        go = torch.randn(N_global, X, device=dev)
        E_local, N_local = E_global, N_global

        def step():
            for p in layer.parameters():
                p.grad = None
            embed.grad = None
            out = layer(g, embed, *extra)
            out.backward(go)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    bwd_name, mm_name = "het_backward_relational_fused_gat_separate_coo", "het_rgnn_relational_matmul"
    HK.event_timers[bwd_name] = []
    step_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        step_events.append((a, b))
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(a.elapsed_time(b) for a, b in step_events)  # device time of every step (events, this rank)
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    ev = HK.event_timers.pop(bwd_name)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = E_global / (dt / args.steps) / 1e6

    def pmc(kernel, field):
        """Per-launch PMC figure from the committed counter passes of this same command (profiles/r01/, written by
        profiles/tools/collect.sh): rocprofv3 cannot run inside the timed process.  None when no profile matches
        this workload.  `kernel` is a prefix of the profile's key (kernel name + grid size)."""
        path = os.path.join(ROOT, "profiles", "r01", "default_pmc.json")
        if args.scale != 1.0 or args.variant != "default" or world != 1 or args.model != "rgat" or not os.path.exists(path):
            return None
        recs = [(int(k.rsplit("grid=", 1)[1]), v) for k, v in json.load(open(path))["kernels"].items()
                if k.startswith(kernel) and field in v]
        return max(recs, key=lambda r: r[0])[1][field] if recs else None  # the largest launch of that kernel (E rows)

    roofline = None
    if ev and not args.variant.startswith("compact") and args.model == "rgat":
        k_ms = sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev)
        nbytes = gat_bwd_bytes(E_local, N_local, H, X)
        parts = {"a5 backward_relational_fused_gat_separate_coo": nbytes}
        if args.variant in ("default", "mulfirst") and g_rels <= 8:
            # the launch also performs the weight gradient of el = <feat, attn_l> (a2 with D_out = 1: reads feat [E,X] and
            # grad_el [E,H], index lists at 8 B), fused into the same pass over feat (fold_attn_l / grad_fold_attn_l)
            parts["a2 weight gradient of el = <feat, attn_l> (D_out = 1), fused into the same launch"] = E_local * (4 * X + 4 * H + 16)
            nbytes += E_local * (4 * X + 4 * H + 16)
        ach = nbytes / (k_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "HET_gat_backward_grouped (backward_relational_fused_gat_separate_coo, kind 0)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": pmc("HET_gat_backward_grouped", "hbm_bytes_per_launch"),
                    "kernel_ms": round(k_ms, 4), "algorithmic_bytes": nbytes, "algorithmic_bytes_by_op": parts,
                    "frac_a5_bytes_only": round(parts["a5 backward_relational_fused_gat_separate_coo"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if roofline["traffic"]:  # physical rate: PMC bytes of the launch over the same duration (measured copy rate of the box: 4.7-5.1 TB/s)
            roofline["traffic_rate_GBps"] = round(roofline["traffic"] / (k_ms * 1e-3) / 1e9, 1)
    # second view, the MFMA side of the path (north_star: MFMA utilisation of the segment GEMM): the reference-named op
    # rgnn_relational_matmul exactly as the reference calls it for the per-edge projection (kind 0, gather by source,
    # E rows, one input head), launched a few times after the timed region, HIP events on the launch stream.
    # (Inside the layer the same product runs on the distinct (relation, node) rows only, see DESIGN.md.)
    roofline_gemm = None
    if args.model == "rgat" and world == 1 and not use_dist:
        sc = g.get_separate_coo_original()
        d_src = {"separate_coo_rel_ptrs": sc["rel_ptrs"], "separate_coo_node_indices": sc["row_indices"],
                 "separate_coo_eids": sc["eids"]}
        Wp = torch.randn(g.get_num_rels(), H, K, X // H, device=dev) * 0.1
        retp = torch.empty(E_local, H, X // H, device=dev)
        HK.event_timers[mm_name] = []
        with torch.no_grad():
            for _ in range(6):
                HK.K.rgnn_relational_matmul(d_src, 0, Wp, embed.detach(), retp, True)
        torch.cuda.synchronize()
        proj = HK.event_timers.pop(mm_name)[1:]
        g_ms = sum(a.elapsed_time(b) for a, b, _ in proj) / len(proj)
        flops = 2.0 * E_local * K * X
        tf = flops / (g_ms * 1e-3) / 1e12
        roofline_gemm = {"bound": "mfma", "kernel": "HET_seg_gemm_mfma (rgnn_relational_matmul, kind 0, E rows, K=X=%d)" % K,
                         "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "kernel_ms": round(g_ms, 4), "flops": flops,
                         "mfma_busy_frac_pmc": pmc("HET_seg_gemm_mfma<64, 2, false, false>", "mfma_busy_frac"),
                         "traffic": pmc("HET_seg_gemm_mfma<64, 2, false, false>", "hbm_bytes_per_launch")}
        del Wp, retp

    # per-entry-point device time, from a few extra steps after the timed region (HIP events around every C-ABI call)
    per_op = None
    if world == 1 and not use_dist:
        HK.event_timers["*"] = []
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        acc = {}
        for a, b, name in HK.event_timers.pop("*"):
            acc[name] = acc.get(name, 0.0) + a.elapsed_time(b) / 3
        per_op = {k[4:]: round(v, 3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}
        per_op["(sum of C-ABI calls)"] = round(sum(acc.values()), 3)

    if rank == 0:
        out = {
            "metric": f"million edges/s (fwd+bwd) {args.model.upper()} layer, ogbn-mag feat=64",
            "value": round(value, 2), "unit": "million edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_step_median_events": round(median_ms, 4),
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model.upper()} layer fwd+bwd, ogbn-mag-shaped synthetic graph (N={N_global}, E={E_global}, R=4), "
                                   f"feat={K}, heads={H}, self_loop, no optimizer step, layer flags: {args.variant}",
                       "edge_order": args.edge_order, "scale": args.scale, "device": torch.cuda.get_device_name(dev),
                       "torch": torch.__version__, "hip": torch.version.hip,
                       "layout_build_ms": None if layout_ms is None else round(layout_ms, 1),
                       "parallelism": "single GPU" if world == 1 else f"dst-range partition x{world}, RCCL all-to-all halo"},
            "roofline": roofline,
            "roofline_segment_gemm": roofline_gemm,
            "per_op_ms": per_op,
            "peak_memory_GB": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
        }
        if world == 1 and not args.no_variants and args.variant == "default" and args.model == "rgat":
            out["variants"] = other_variants(args, coo, dev, min(args.steps, 10), ms_per_step, value)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1 or os.environ.get("HET_FORCE_DIST") == "1":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
