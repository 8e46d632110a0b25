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
import datetime
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

PROF_ROUND = "r05"
PMC_OPTIONAL_LAUNCHES = ("HET_rgat_colsum_rows",)


def kernel_source_sha16():
    """Digest of het_amd/csrc/* (as profiles/tools/summarize.py records it with the counters)."""
    import hashlib
    root = os.path.join(ROOT, "het_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def make_views(args, world, prof_tag, default_heads, heads=None):
    """(pmc, hbm_view) for one workload: prof_tag names the committed counter file profiles/<round>/<prof_tag>_pmc.json
    ({default, compact, ..., rgcn, hgt})."""
    prof_dir = os.path.join(ROOT, "profiles", PROF_ROUND)
    tree_sha = kernel_source_sha16()
    heads = args.heads if heads is None else heads

    def pmc(kernel, field):
        """Per-launch PMC figure of `kernel` from the COMMITTED counter passes of this same command (profiles/<round>/,
        written by profiles/tools/collect.sh on an earlier box): rocprofv3 cannot run inside the timed process, so
        this is not an observation of this run -- the JSON says so (`traffic_source`).  None when no committed
        profile matches this workload.  `kernel` is a prefix of the profile's key (kernel name + grid size)."""
        path = os.path.join(prof_dir, f"{prof_tag}_pmc.json")
        if (args.scale != 1.0 or world != 1 or args.feat != 64 or heads != default_heads or args.edge_order != "src_dst"
                or not os.path.exists(path)):
            return None
        prof = json.load(open(path))
        if prof.get("kernel_source_sha16") != tree_sha:
            return None  # the kernels changed since those counters were collected: a stale figure is worse than none
        kernels = prof["kernels"]
        if isinstance(kernel, (tuple, list)):  # an op implemented by several launches per step: the sum over them
            parts = [(k_, pmc(k_, field)) for k_ in kernel]
            # (a launch the op only makes for some callers -- the bias column sums, which the layer takes from the self-loop's
            #  weight-gradient launch since round 5 -- counts when the profiled command made it)
            parts = [p_ for k_, p_ in parts if not (p_ is None and k_ in PMC_OPTIONAL_LAUNCHES)]
            return None if any(p_ is None for p_ in parts) else sum(parts)
        recs = [(int(k.rsplit("grid=", 1)[1]), v) for k, v in kernels.items() if k.startswith(kernel) and field in v]
        return max(recs, key=lambda r: r[0])[1][field] if recs else None  # the largest launch of that kernel

    def hbm_view(kernel, k_ms, nbytes, extra=None, pmc_name=None):
        ach = nbytes / (k_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "frac_of_measured_copy_rate": round(ach / HBM_COPY_GBS, 4),
             "kernel_ms": round(k_ms, 4), "algorithmic_bytes": int(nbytes),
             "traffic": pmc(pmc_name or kernel.split(" ")[0], "hbm_bytes_per_launch")}
        r["traffic_source"] = (f"profiles/{PROF_ROUND}/{prof_tag}_pmc.json (committed rocprofv3 --pmc passes of this command on "
                               "another box; not measured in this run)") if r["traffic"] else None
        if r["traffic"]:
            r["traffic_rate_GBps"] = round(r["traffic"] / (k_ms * 1e-3) / 1e9, 1)
        if extra:
            r.update(extra)
        return r

    return pmc, hbm_view




def gather_ceiling(nbytes, rows_gathered, row_bytes):
    """(rounds 3-4 printed a `ceiling_frac` from the guide's indexed-row rate here; VERDICT r04: the repository's own RGCN gather-sum
    beats it, so it is not a ceiling.  Replaced by a MEASURED reference: gather_rate_reference below.)"""
    return {}


def gather_rate_reference(g, dev, N, X):
    """A plain gather-sum (the RGCN layer's kernels: HET_segment_sum_packed + _long, no softmax, one output row per segment) over the
    SAME groupings and row tables the RGAT passes use, timed in this process: by destination over the feat rows (forward) and by
    (relation, source) row over the gradout rows (backward).  What the gather of the rows alone costs on this graph, on this box --
    the RGAT passes do that plus their attention arithmetic and their outputs."""
    from het_amd import _lib as HL
    from het_amd import kernels as HK
    s = g.get_separate_coo_original()
    ss = g.get_separate_unique_node_indices_single_sided()
    inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    S_row, S_col = ss["node_indices_row"].numel(), ss["node_indices_col"].numel()
    grp = HK.rgat_compact_groupings(s["col_indices"], inv["inverse_indices_row"], inv["inverse_indices_col"], N, S_row, S_col)
    if grp is None:
        return None
    feat, go = torch.randn(S_row, X, device=dev), torch.randn(N, X, device=dev)
    out_f, out_b = torch.zeros(N, X, device=dev), torch.zeros(S_row, X, device=dev)
    res = {}
    for name, gg, src, out in (("forward_ms", grp[0], feat, out_f), ("backward_ms", grp[1], go, out_b)):
        def run():
            HK._call(out, "het_rows_scatter_add_grouped", gg.handle, HK._p(src), X, HK._p(out), out.shape[0], HK._stream(out))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record()
        torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / 10, 4)
    res["what"] = ("het_rows_scatter_add_grouped (= HET_segment_sum_packed + _long, '+=' into the output rows) on the RGAT step's own "
                   "groupings: out[dst] += SUM feat_c[srow_e] (forward), out[srow] += SUM gradout[dst_e] (backward); 21.1 M gathered "
                   "256-byte rows each")
    return res


KT_NAMES = ("HET_rgat_backward_dst_pack", "HET_rgat_backward_drow_pass", "HET_rgat_backward_src_short", "HET_rgat_backward_src_long", "HET_rgat_backward_er_runs",
            "HET_rgat_backward_src", "HET_rgat_backward", "HET_rgat_aggregate_packs", "HET_rgat_aggregate_hubs",
            "HET_rgat_aggregate_finish", "HET_rgat_aggregate", "HET_gat_backward_src", "HET_gat_backward_grouped",
            "HET_gat_aggregate_grouped", "HET_seg_gemm_mfma<store>",
            "HET_seg_gemm_mfma<atomic>", "HET_seg_gemm_mfma<dot>", "HET_seg_gemm_mfma<rmw>", "HET_seg_dw_mfma", "HET_segment_sum",
            "HET_node_dx", "HET_node_rows_sum", "HET_colsum",
            "HET_hgt_aggregate_rows", "HET_hgt_backward_dst_rows", "HET_hgt_backward_src_short", "HET_hgt_backward_src_long")


def read_kernel_timers(ksteps):
    """{timer label: (avg ms per launch, launches per step, ms per step)} of the library's per-kernel HIP-event timers."""
    from het_amd import _lib as HL
    kt = {}
    for name in KT_NAMES:
        ms, n = HL.kernel_timing_read(name)
        if n:
            kt[name] = (ms / n, n / ksteps, ms / ksteps)
    return kt


def rgcn_rooflines(g, kt, E_local, N_local, K, X, hbm_view):
    """RGCN (BASELINE.json configs[1]): the step is two gather-sums (x[src] * norm per (relation, destination) forward,
    gradout[dst] * norm per (relation, source) backward) + GEMMs on the distinct rows.  a7 / a8 bytes per SURVEY.md 8d
    (U_src = distinct source nodes); the gather passes move one 256-byte row per edge like RGAT's."""
    sc = g.get_separate_coo_original()
    U_src = int(torch.unique(sc["row_indices"]).numel())
    R_ = g.get_num_rels()
    a7 = E_local * (3 * 8 + 4) + U_src * 4 * K + N_local * 4 * X + 4 * R_ * K * X
    a8 = E_local * 28 + (U_src + N_local) * 4 * (K + X) + 8 * R_ * K * X
    ss_ms = kt["HET_segment_sum"][0]  # one launch per op
    note = ("kernel_ms = the gather-sum launches of one op (average of the forward and the backward one); the op's products on "
            "the distinct rows are separate launches (kernel_ms: HET_node_rows_sum, the node-major pass of het_rgcn_layer_forward / "
            "_backward; HET_seg_gemm_mfma<rmw> with HET_RGCN_FUSED=0).  requested_bytes counts one row per edge: the rate at "
            "the kernels' load instructions, part of it served by L2 / Infinity Cache (the [N,64] table is 0.5 GB and "
            "the degrees are Zipf-skewed), so it may exceed what HBM delivers")
    pm = ("HET_segment_sum_packed", "HET_segment_sum_long")
    bwd = hbm_view("HET_segment_sum_packed + _long (backward_rgcn_layer1_separate_coo: gradout rows by (relation, source))",
                   ss_ms, a8, {"requested_bytes_one_row_per_edge": int(E_local * (4 * X + 12) + a8),
                               "requested_rate_GBps": round((E_local * (4 * X + 12) + a8) / (ss_ms * 1e-3) / 1e9, 1),
                               "note": note, **gather_ceiling(a8, E_local, 4 * X)}, pmc_name=pm)
    fwd = hbm_view("HET_segment_sum_packed + _long (rgcn_layer1_separate_coo: x rows by (relation, destination))", ss_ms, a7,
                   {"requested_bytes_one_row_per_edge": int(E_local * (4 * K + 12) + a7),
                    "requested_rate_GBps": round((E_local * (4 * K + 12) + a7) / (ss_ms * 1e-3) / 1e9, 1),
                    **gather_ceiling(a7, E_local, 4 * K)}, pmc_name=pm)
    return bwd, fwd


def hgt_rooflines(g, kt, E_local, N_local, K, X, H, hbm_view):
    """HGT (BASELINE.json configs[3]): attention + aggregation on the distinct (relation, source) rows (csrc/hgt_compact.hip).
    Algorithmic bytes: every tensor of the pass once (kv_c [S_row,2X], q / out / gradout [N,X], per-(node, head) scalars,
    two 8-byte indices per edge); the passes gather one 2X-float row (forward, destination side) or two X-float rows
    (source side) PER EDGE from tables larger than the Infinity Cache -- bytes_with_per_edge_row_gather counts those."""
    ss = g.get_separate_unique_node_indices_single_sided()
    S_row = int(ss["node_indices_row"].numel())
    kvb, nxb, nhb = S_row * 2 * X * 4, N_local * X * 4, N_local * H * 4
    f_b = kvb + 2 * nxb + nhb + E_local * 16
    d_b = kvb + 4 * nxb + 3 * nhb + E_local * 16            # reads kv_c, q, gradout, out, lsum; writes grad_q, pack2
    s_b = 2 * kvb + 2 * nxb + 2 * nhb + E_local * 16        # reads kv_c, q, gradout, pack2; writes grad_kv_c
    f_req = f_b + (E_local - S_row) * 2 * X * 4
    d_req = d_b + (E_local - S_row) * 2 * X * 4
    s_req = s_b + (E_local - N_local) * (2 * X * 4 + 8 * H)
    b_ms = sum(kt[n][2] for n in ("HET_hgt_backward_dst_rows", "HET_hgt_backward_src_short", "HET_hgt_backward_src_long") if n in kt)
    note = ("requested_bytes counts one gathered row per edge: the rate at the kernels' load instructions, part of it served by "
            "L2 / Infinity Cache (kv_c is 1.2 GB, q / gradout 0.5 GB each, degrees Zipf-skewed), so it may exceed what HBM "
            "delivers; `traffic` is what crossed the memory fabric")
    ex = lambda req, ms, rows: {"S_row": S_row, "requested_bytes_one_row_per_edge": int(req),  # noqa: E731
                                "requested_rate_GBps": round(req / (ms * 1e-3) / 1e9, 1), "note": note, **rows}
    bwd = hbm_view("HET_hgt_backward_dst_rows + _src_short + _src_long (het_hgt_backward_compact: softmax + aggregation backward "
                   "of the HGT layer, three launches)", b_ms, d_b + s_b, ex(d_req + s_req, b_ms, gather_ceiling(d_b + s_b, 2 * E_local, 2 * X * 4)),
                   pmc_name=("HET_hgt_backward_dst_rows", "HET_hgt_backward_src_short", "HET_hgt_backward_src_long"))
    f_ms = kt["HET_hgt_aggregate_rows"][2]
    fwd = hbm_view("HET_hgt_aggregate_rows (het_hgt_aggregate_compact: score, edge softmax, message aggregation)",
                   f_ms, f_b, ex(f_req, f_ms, gather_ceiling(f_b, E_local, 2 * X * 4)))
    return bwd, fwd


def other_model(model, args, coo, dev, world):
    """BASELINE.json configs[1] (RGCN) / configs[3] (HGT, 8 heads) on the same graph, timed like the headline (barrier, K steps,
    barrier) with fewer steps, with their own rooflines: the driver's default run then carries a number for every single-GPU
    config, not only the headline."""
    from het_amd import _lib as HL
    from het_amd.graph import HetGraph
    E, N, K = coo.num_edges, coo.num_nodes, args.feat
    X, H = K, 8 if model == "hgt" else 1
    g = HetGraph.from_integrated_coo(coo, full=model == "hgt")
    torch.manual_seed(0)
    extra = ()
    if model == "rgcn":
        from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
        layer = HET_EglRelGraphConv_EdgeParallel(K, X, g.get_num_rels()).to(dev)
        extra = (torch.rand(E, 1, device=dev),)
    else:
        from het_amd.layers import HET_HGTLayerHetero
        layer = HET_HGTLayerHetero(g.get_num_ntypes(), g.get_num_rels(), K, X, num_heads=H, dropout=0.0).to(dev)
    embed = torch.nn.Parameter(torch.empty(N, K, device=dev))
    torch.nn.init.xavier_uniform_(embed)
    go = torch.randn(N, X, device=dev)

    def step():
        for q in layer.parameters():
            q.grad = None
        embed.grad = None
        layer(g, embed, *extra).backward(go)

    steps = max(3, min(args.steps, 10))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    HL.kernel_timing(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    HL.kernel_timing(False)
    kt = read_kernel_timers(3)
    _, view = make_views(args, world, model, 8 if model == "hgt" else 4, heads=8 if model == "hgt" else 4)
    rb = rf = None
    if model == "rgcn" and "HET_segment_sum" in kt:
        rb, rf = rgcn_rooflines(g, kt, E, N, K, X, view)
    if model == "hgt" and "HET_hgt_aggregate_rows" in kt:
        rb, rf = hgt_rooflines(g, kt, E, N, K, X, H, view)
    res = {"config": f"{model.upper()} layer fwd+bwd, same graph, feat={K}" + (f", heads={H}" if model == "hgt" else "") +
                     " (BASELINE.json configs[%d]); layer swap: het_amd.layers.%s" % (3 if model == "hgt" else 1, type(layer).__name__),
           "ms_per_step": round(dt * 1e3, 4), "million_edges_per_s": round(E / dt / 1e6, 2), "steps": steps, "warmup": 3,
           "roofline": rb, "roofline_forward": rf,
           "kernel_ms": {k: {"avg_ms": round(v[0], 4), "launches_per_step": round(v[1], 2), "ms_per_step": round(v[2], 4)} for k, v in kt.items()}}
    del layer, embed, go, g
    torch.cuda.empty_cache()
    return res


def edge_order_random(args, dev):
    """The headline layer on the same synthetic graph with every relation's edges in GENERATION order (SURVEY.md 8d's wording;
    --edge-order random) instead of the sorted lists OGB ships: same edges, same counts, other list order."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=args.scale, edge_order="random")
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    g = HetGraph.from_integrated_coo(coo, full=args.variant.startswith("compact"))
    torch.manual_seed(0)
    layer = HET_RGATLayer(args.feat, args.feat, g.get_num_rels(), args.heads, self_loop=True, dropout=0.0, **layer_flags(args.variant)).to(dev)
    embed = torch.nn.Parameter(torch.empty(coo.num_nodes, args.feat, device=dev))
    torch.nn.init.xavier_uniform_(embed)
    go = torch.randn(coo.num_nodes, args.feat, device=dev)

    def step():
        for q in layer.parameters():
            q.grad = None
        embed.grad = None
        layer(g, embed).backward(go)

    steps = max(3, min(args.steps, 10))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    E = coo.num_edges
    del layer, embed, go, g
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 4), "million_edges_per_s": round(E / dt / 1e6, 2), "steps": steps,
            "what": "same layer, same edges, each relation's list in generation order (make_mag_like(edge_order='random'))"}


def dist_rehearsal(args, dev, ranks=8, feat=128, heads=4):
    """BASELINE.json configs[4] (RGAT, feat 128, destination-range partition over 8 GPUs) as far as ONE GPU goes: every rank of the
    8-way partition of the same graph as a logical rank of this process (het_amd.dist.LocalRanks) -- its plan, its local graph, the
    pack of its sends, the layer's own exchange path (forward_with_halo: projection of the halo rows piece by piece, aggregation;
    backward ordered around the return) and the unpack of the returned rows -- timed rank by rank with the all-to-all replaced by
    slicing.  max_ms is the compute input of an 8-GPU step; what it leaves out is the wire (halo_MB_* per rank and exchange) and
    RCCL itself, which no box available to this build can run with more than one rank (tests/test_gpu_dist.py skips those)."""
    import het_amd.dist as D
    from het_amd import plan as HP
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    t_wall = time.perf_counter()
    HP.clear()
    coo = make_mag_like(scale=args.scale, edge_order=args.edge_order)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    torch.manual_seed(0)
    layer = HET_RGATLayer(feat, feat, coo.num_rels, heads, self_loop=True, dropout=0.0).to(dev)
    lr = D.LocalRanks(coo, ranks, layer)
    x_own = [torch.nn.Parameter(torch.randn(p.n_own, feat, device=dev) * 0.1) for p in lr.plans]
    go = [torch.randn(p.n_own, feat, device=dev) for p in lr.plans]
    for r, p in enumerate(lr.plans):  # what the peers would have pushed
        lr.wire.push[r] = D._gather_rows(x_own[r].detach(), p.send_idx)
    steps, per_rank = max(3, min(args.steps, 10)), []
    for r, p in enumerate(lr.plans):
        back = torch.zeros(p.send_idx.numel(), feat, device=dev)  # (the rows the peers would return)

        def step():
            layer.zero_grad(set_to_none=True)
            x_own[r].grad = None
            D._gather_rows(x_own[r].detach(), p.send_idx)  # pack of this rank's own sends
            out = layer.forward_with_halo(lr.graphs[r], x_own[r], lr.halos[r])
            out.backward(go[r])
            D._scatter_add_rows(x_own[r].grad, p.send_idx, back)  # unpack of the returned rows

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            step()
        b.record()
        torch.cuda.synchronize()
        per_rank.append({"rank": r, "ms": round(a.elapsed_time(b) / steps, 4), "owned_nodes": int(p.n_own), "local_edges": int(p.num_local_edges),
                         "halo_rows_received": int(p.n_halo), "halo_rows_sent": int(p.send_idx.numel()),
                         "halo_MB_received_per_exchange": round(int(p.n_halo) * feat * 4 / 1e6, 2),
                         "halo_MB_sent_per_exchange": round(int(p.send_idx.numel()) * feat * 4 / 1e6, 2)})
    max_ms = max(q["ms"] for q in per_rank)
    E = coo.num_edges
    res = {"what": dist_rehearsal.__doc__.split("\n")[0].strip() + " ... (bench.py::dist_rehearsal)",
           "ranks": ranks, "feat": feat, "heads": heads, "pieces": int(lr.plans[0].chunks), "edge_cut": int(lr.plans[0].edge_cut),
           "per_rank_ms": [q["ms"] for q in per_rank], "max_ms": max_ms, "min_ms": min(q["ms"] for q in per_rank),
           "million_edges_per_s_if_the_exchange_were_free": round(E / max_ms / 1e3, 1),
           "exchange": "loopback (slicing), excluded from per_rank_ms; on the wire: halo_MB_* per rank and direction, 7 xGMI links per GPU",
           "per_rank": per_rank, "timed_steps_per_rank": steps, "wall_s": None}
    del lr, x_own, go, layer
    HP.clear()
    torch.cuda.empty_cache()
    res["wall_s"] = round(time.perf_counter() - t_wall, 1)
    return res


def dry_run_exchange(args):
    """`bench.py --gpus N --dry-run-exchange` under torch.distributed.run: the launch line, the rendezvous, every rank's plan of
    the N-way partition and one halo exchange each way, on HOST tensors over gloo (no GPU is touched, no layer runs): rank 0
    prints the `dist` object the real run prints, with checksums that prove every halo row arrived from its owner and every
    returned gradient row reached it.  For rehearsing rank counts a one-GPU box cannot hold (its GPU takes 6 processes)."""
    import torch.distributed as dist
    from het_amd.dist import HaloExchange, build_plan
    from het_amd.synth import make_mag_like
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // max(1, world)))
    if "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo")
    try:
        coo = make_mag_like(scale=args.scale, edge_order=args.edge_order)
        plan = build_plan(coo, rank, world)
        K = args.feat
        own = plan.node_order[int(plan.bounds[rank]): int(plan.bounds[rank + 1])]
        # feature row of global node v = v + column / 1024: a received halo row names the node it belongs to
        x_own = (own.to(torch.float64).unsqueeze(1) + torch.arange(K, dtype=torch.float64) / 1024.0).requires_grad_(True)
        x_local = HaloExchange.apply(x_own, plan, None)
        # halo_global holds renumbered ids: node_order maps them back to original ids
        want = plan.node_order[plan.halo_global].to(torch.float64)
        push_ok = bool(torch.equal(x_local[plan.n_own:, 0].detach(), want)) and bool(torch.equal(x_local[: plan.n_own].detach(), x_own.detach()))
        # backward: every rank returns gradient 1 for each of its halo rows; an owner's row gets 1 + the number of ranks that hold it
        x_local.backward(torch.ones_like(x_local))
        holders = torch.zeros(coo.num_nodes, dtype=torch.float64)
        for r in range(world):  # (every rank can derive every other rank's plan: no communication)
            pr = plan if r == rank else build_plan(coo, r, world)
            holders[pr.node_order[pr.halo_global]] += 1.0
        ret_ok = bool(torch.equal(x_own.grad[:, 0], 1.0 + holders[own]))
        mine = {"rank": rank, "owned_nodes": int(plan.n_own), "halo_rows_received": int(plan.n_halo), "halo_rows_sent": int(plan.send_idx.numel()),
                "halo_MB_sent_per_exchange": round(int(plan.send_idx.numel()) * K * 4 / 1e6, 2),
                "halo_MB_received_per_exchange": round(int(plan.n_halo) * K * 4 / 1e6, 2), "local_edges": int(plan.num_local_edges),
                "send_counts": plan.send_counts, "push_rows_verified": push_ok, "returned_rows_verified": ret_ok}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            print(json.dumps({"metric": "million edges/s (fwd+bwd) RGAT layer, ogbn-mag feat=64", "value": None, "unit": "million edges/s",
                              "n_gpus": world, "dry_run_exchange": True, "data": "synthetic",
                              "config": {"workload": f"partition plan + halo exchange rehearsal on host tensors (gloo), N={coo.num_nodes}, "
                                                     f"E={coo.num_edges}, feat={K}: no layer, no GPU, no timing",
                                         "scale": args.scale, "edge_order": args.edge_order,
                                         "parallelism": f"dst-range partition x{world}"},
                              "dist": {"ranks": world, "backend": "gloo (host tensors)", "edge_cut": int(plan.edge_cut),
                                       "edges_total": int(sum(r_["local_edges"] for r_ in allr)), "per_rank": allr,
                                       "all_rows_verified": all(r_["push_rows_verified"] and r_["returned_rows_verified"] for r_ in allr)}}))
        if not (push_ok and ret_ok):
            raise SystemExit(f"rank {rank}: halo rows did not arrive where they belong (push {push_ok}, return {ret_ok})")
    finally:
        dist.destroy_process_group()



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
    p.add_argument("--edge-order", default="src_dst", choices=["src_dst", "src", "random"],
                   help="order of a relation's edges in the synthetic lists (het_amd/synth.py): src_dst = sorted by (source, "
                        "destination) as the OGB lists and the reference's shipped slice are (headline); random = generation order "
                        "(SURVEY.md 8d's wording; reported beside the headline as edge_order_random); src = by source only (rounds 1-3)")
    p.add_argument("--dry-run-exchange", action="store_true",
                   help="multi-rank rehearsal WITHOUT a GPU: rendezvous (gloo), every rank's partition plan and one halo exchange each "
                        "way on host tensors with checksums, then the `dist` object -- no layer, no timing, value null")
    p.add_argument("--model", default="rgat", choices=["rgat", "rgcn", "hgt"],
                   help="rgat = the BASELINE.json metric; rgcn / hgt time BASELINE.json configs[1] / configs[3] (single GPU)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-variants", action="store_true", help="skip timing the other reference flag combinations")
    p.add_argument("--no-models", action="store_true", help="skip the RGCN / HGT legs (BASELINE.json configs[1], configs[3]) of the default run")
    p.add_argument("--no-dist-rehearsal", action="store_true",
                   help="skip the 8-way feat-128 partition rehearsal (BASELINE.json configs[4] on one GPU: dist_rehearsal) of the default run")
    p.add_argument("--cpu-scale", type=float, default=0.25, help="graph scale of the CPU-baseline sample (1.0 = the full workload: "
                   "one warm-up + one timed iteration, about 3 minutes of host time -- outside the default run)")
    a = p.parse_args()
    if a.heads is None:
        a.heads = 8 if a.model == "hgt" else 4
    return a


def layer_flags(variant):
    return dict(compact_as_of_node_flag=variant.startswith("compact"),
                compact_direct_indexing_flag=variant.startswith("compact"),
                multiply_among_weights_first_flag=variant.endswith("mulfirst"))


def gat_fwd_bytes(E, N, H, X, S_row=None, S_col=None):
    """Algorithmic bytes of relational_fused_gat_separate_coo (every API-visible tensor touched once,
    index arrays at their API width of 8 B).  kind 0 (S_row None): per edge reads col, eids, el, er (H), feat (X), writes
    exp (H); per node writes sum (H), ret (X).  Direct-index compact kind 4: per edge reads col, eids and the two
    inverse-index entries, writes exp; feat / el are [S_row, .] and er [S_col, H], read once each."""
    if S_row is None:
        return E * (2 * 8 + 4 * (2 * H + X) + 4 * H) + N * 4 * (H + X)
    return E * (4 * 8 + 4 * H) + S_row * 4 * (X + H) + S_col * 4 * H + N * 4 * (H + X)


def gat_bwd_bytes(E, N, H, X, S_row=None, S_col=None):
    """Algorithmic bytes of backward_relational_fused_gat_separate_coo.  kind 0: per edge reads col, eids
    (8 B each), el, er, exp (H floats each), feat (X); writes grad_el, grad_er (H), grad_feat (X); per node reads sum (H),
    ret, gradout (X each).  Kind 4: per edge reads col, eids, two inverse-index entries and exp; feat, el, grad_feat,
    grad_el live on the S_row (relation, source) rows, er / grad_er on the S_col (relation, destination) rows."""
    if S_row is None:
        return E * (2 * 8 + 4 * (3 * H + X) + 4 * (2 * H + X)) + N * 4 * (H + 2 * X)
    return E * (4 * 8 + 4 * H) + S_row * 4 * 2 * (X + H) + S_col * 4 * 2 * H + N * 4 * (H + 2 * X)


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
    for it in range(4):  # one warm-up + three timed (a quarter of the workload: ~14 s per step on the 128 host threads)
        t0 = time.perf_counter()
        out = OL.rgat_layer(x, W, al, ar, s["rel_ptrs"], s["row_indices"], s["col_indices"], N, 0.2, lw, None)
        torch.autograd.grad(out, [x, W, al, ar, lw], go)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(g.get_num_edges() / med / 1e6, 3), "unit": "million edges/s", "cores": torch.get_num_threads(),
            "kind": "port", "scale": args.cpu_scale, "sample_edges": g.get_num_edges(), "sample_seconds_per_step": round(med, 3),
            "sample_seconds_per_step_min": round(times[0], 3), "timed_iterations": len(times),
            "value_best_iteration": round(g.get_num_edges() / times[0] / 1e6, 3),
            "sample": f"oracle/layers.py rgat_layer fwd+bwd (torch CPU fp32, HET cross-relation softmax) on a mag-like "
                      f"graph at scale {args.cpu_scale} ({g.get_num_edges()} edges, {N} nodes), median of {len(times)} after 1 warm-up; "
                      f"os.cpu_count()={os.cpu_count()}"}


def other_variants(args, coo, dev, steps, default_ms, default_value):
    """The same layer under the reference's other flag combinations (identical outputs; its sweep runs them too,
    hrt/utils/_do_all_cases.sh:2-40).  Reported next to the headline, which stays on the default flags."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    res = {"default": {"ms_per_step": round(default_ms, 4), "million_edges_per_s": round(default_value, 2)}}
    g = HetGraph.from_integrated_coo(coo, full=True)
    E, N = coo.num_edges, coo.num_nodes
    flag_names = {"reference_op_sequence": "default flags, the reference's op sequence literally: only reference-named torch_hrt "
                                           "ops with the reference wrappers' zero-filled buffers (het_amd/backend/"
                                           "reference_protocol.py = RGAT/models.py:265-385 after the kernels/__init__.py swap)",
                  "op_by_op": "default flags, the reference's model code line by line (RGAT/models.py:265-385) on this package's "
                              "backend wrappers (het_amd.backend in place of hrt/python/backend: no fills, '=' gradients) -- "
                              "no one-node layer",
                  "op_by_op_compact": "--compact_as_of_node_flag --compact_direct_indexing_flag, the reference's model code line by "
                                      "line (RGAT/models.py:152-263) on this package's backend wrappers -- no one-node layer",
                  "compact": "--compact_as_of_node_flag --compact_direct_indexing_flag",
                  "mulfirst": "--multiply_among_weights_first_flag",
                  "compact_mulfirst": "--compact_as_of_node_flag --compact_direct_indexing_flag --multiply_among_weights_first_flag"}
    for variant in ("reference_op_sequence", "op_by_op", "op_by_op_compact", "compact", "mulfirst", "compact_mulfirst"):
        torch.manual_seed(0)
        if variant in ("reference_op_sequence", "op_by_op") and torch.cuda.mem_get_info(dev)[0] < 80 * 2**30 * args.scale:
            continue  # needs about ten [E,H,D] tensors (54 GB on ogbn-mag)
        flags = {"reference_op_sequence": {}, "op_by_op": {}, "op_by_op_compact": layer_flags("compact")}.get(variant)
        layer = HET_RGATLayer(args.feat, args.feat, g.get_num_rels(), args.heads, self_loop=True, dropout=0.0,
                              reference_op_sequence=variant == "reference_op_sequence",
                              **(layer_flags(variant) if flags is None else flags)).to(dev)
        layer.op_by_op = variant.startswith("op_by_op")
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
        torch.cuda.empty_cache()
    return res


def main():
    """Any failure of a rank -- an RCCL / HIP error included -- is printed with the rank and the failing call's traceback and
    ends the process with a non-zero exit code (torch.distributed.run then stops the other ranks): no retry, no fallback."""
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as ex:  # noqa: BLE001
        import traceback
        sys.stderr.write(f"[bench.py rank {os.environ.get('RANK', '0')}/{os.environ.get('WORLD_SIZE', '1')}] FAILED: "
                         f"{type(ex).__name__}: {ex}\n{traceback.format_exc()}")
        sys.stderr.flush()
        os._exit(1)  # (a rank stuck in a collective would keep a normal interpreter shutdown waiting)


def _stamp():
    return datetime.datetime.now(datetime.timezone.utc).isoformat(timespec="milliseconds")


def _main():
    args = parse()
    if args.dry_run_exchange:
        return dry_run_exchange(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The host-cores baseline runs FIRST (it is ~60 s of CPU work and touches no GPU): everything after `gpu_phase_start` is one
    # contiguous window of GPU work, which is what a utilisation sampler outside this process wants to see.
    wall = {"process_start": _stamp()}
    cpu_res = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(args)
        wall["cpu_baseline_end"] = _stamp()
    wall["gpu_phase_start"] = _stamp()
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run "
                         f"--nproc-per-node N (and pass the same N as --gpus)")
    # (rehearsals of the multi-rank flow on a one-GPU box: HET_DIST_BACKEND=gloo lets the ranks share the device)
    backend = os.environ.get("HET_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPUs are visible (RCCL needs one GPU per rank)")
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
    mm_name = "het_rgnn_relational_matmul"
    from het_amd import _lib as HL
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
    # per-kernel durations: a few more steps AFTER the timed region, with HIP-event pairs on the launch stream around the
    # library's dominant kernels (include/het_amd.h: het_kernel_timing_enable) -- the headline steps run without that
    # instrumentation (two event records per launch under a mutex)
    ksteps = max(1, min(args.steps, 5))
    HL.kernel_timing(True)
    for _ in range(ksteps):
        step()
    barrier()
    HL.kernel_timing(False)
    per_step = sorted(a.elapsed_time(b) for a, b in step_events)  # device time of every step (events, this rank)
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    # live per-kernel durations of the timed region: {kernel: (avg ms per launch, launches per step, ms per step)}; the roofline
    # objects use ms per step = the summed duration of the launches that together implement the op in one step
    kt = read_kernel_timers(ksteps)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = E_global / (dt / args.steps) / 1e6

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
        per_op["(sum of C-ABI calls; launches on the side stream overlap others)"] = round(sum(acc.values()), 3)

    pmc, hbm_view = make_views(args, world, args.variant if args.model == "rgat" else args.model,
                               8 if args.model == "hgt" else 4)

    # roofline of the step's dominant kernels: algorithmic bytes of the op the kernel implements (every
    # API-visible tensor once, indices at 8 B) / the kernel's own average duration in the timed region.  Only the
    # bytes of THAT op are counted -- work fused into the launch from other ops is reported beside it, not added.
    roofline = roofline_fwd = None
    if args.model == "rgat":
        S_row = S_col = None
        compact_flow = "HET_rgat_backward_src" in kt or "HET_gat_backward_src" in kt
        if compact_flow:
            ss = (runner.dl.graph if use_dist else g).get_separate_unique_node_indices_single_sided()
            S_row, S_col = int(ss["node_indices_row"].numel()), int(ss["node_indices_col"].numel())
        gather = {}
        if S_row is not None:
            # the [S_row, X] / [N, X] tables the passes gather from (0.9 / 0.5 GB) exceed the 256 MiB Infinity Cache, so
            # every edge's row really crosses the memory fabric: the bytes the kernel is REQUIRED to move
            gather = {"S_row": S_row, "S_col": S_col}
        bname = next((n for n in ("HET_rgat_backward_src", "HET_gat_backward_src", "HET_gat_backward_grouped") if n in kt), None)
        if bname:
            nb_ = gat_bwd_bytes(E_local, N_local, H, X, S_row, S_col)
            ex = dict(gather)
            if S_row is not None:
                req = nb_ + (E_local - S_row) * 4 * X  # one gradout row per edge instead of per source row
                ex.update(bytes_with_per_edge_row_gather=int(req),
                          frac_with_per_edge_row_gather=round(req / (kt[bname][2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          **gather_ceiling(nb_, E_local, 4 * X))
            pm, b_ms, what = bname, kt[bname][2], "0"
            if bname == "HET_rgat_backward_src":
                # every launch of the op: per-destination pack, short + long (relation, source) segments, grad_er (from the run
                # sums the forward left: HET_rgat_grad_er_runs; else the segmented sum of the per-edge term)
                runs = "HET_rgat_backward_er_runs" in kt or "HET_rgat_backward_drow_pass" in kt
                fused = "HET_rgat_backward_drow_pass" in kt  # (round 5: one pass per er row instead of dst pack + records + grad_er)
                pm = (("HET_rgat_drow_pass", "HET_rgat_colsum_rows") if fused else ("HET_rgat_dst_pack",)) + (
                    "HET_rgat_backward_src_coop", "HET_rgat_backward_src_long") + (
                    () if fused else ("HET_rgat_grad_er_runs" if runs else "HET_segment_sum_flat4",))
                b_ms = kt["HET_rgat_backward"][2] + (0.0 if runs else kt.get("HET_segment_sum", (0, 0, 0.0))[2])
                # the launches of the op run side by side on two streams (csrc/common.hip.h: HetFork), so their own durations
                # overlap: the op's time is the span between its entry and its return on the caller's stream (per_op_ms)
                op_ms = (per_op or {}).get("rgat_backward_compact_runs" if runs else "rgat_backward_compact")
                ex["launch_ms_overlapping"] = {n: round(kt[n][2], 4) for n in ("HET_rgat_backward_dst_pack", "HET_rgat_backward_drow_pass",
                                                                             "HET_rgat_backward_src_short", "HET_rgat_backward_src_long",
                                                                             "HET_rgat_backward_er_runs") if n in kt}
                if op_ms:
                    b_ms = op_ms
                bname = (("HET_rgat_drow_pass + " if fused else "HET_rgat_dst_pack + ") + "HET_rgat_backward_src_coop + _src_long"
                         + ("" if fused else " + " + ("HET_rgat_grad_er_runs" if runs else "HET_segment_sum")))
                what = "4: rows of the distinct (relation, node) projections; all launches of the op"
                ex["frac_with_per_edge_row_gather"] = round(req / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roofline = hbm_view(f"{bname} (backward_relational_fused_gat_separate_coo, kind {what})", b_ms, nb_, ex, pmc_name=pm)
        fname = next((n for n in ("HET_rgat_aggregate", "HET_gat_aggregate_grouped") if n in kt), None)
        if fname:
            nf_ = gat_fwd_bytes(E_local, N_local, H, X, S_row, S_col)
            f_ms = kt[fname][2]  # (the timers are read by prefix: every launch of the op)
            ex = dict(gather)
            if S_row is not None:
                req = nf_ + (E_local - S_row) * 4 * X
                ex.update(bytes_with_per_edge_row_gather=int(req),
                          frac_with_per_edge_row_gather=round(req / (kt[fname][2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          **gather_ceiling(nf_, E_local, 4 * X))
            pmf = fname
            if fname == "HET_rgat_aggregate":  # (one launch, or packs + hub work items + hub finish when the run sums are on)
                runs = "HET_rgat_aggregate_hubs" in kt or "HET_rgat_aggregate_packs" in kt
                pmf = ("HET_rgat_aggregate_runs_packed", "HET_rgat_aggregate_hub_items", "HET_rgat_finish_hubs") if runs else "HET_rgat_aggregate_coop"
                if runs:
                    fname = "HET_rgat_aggregate_runs_packed + _hub_items + HET_rgat_finish_hubs"
                    ex["launch_ms_overlapping"] = {n: round(kt[n][2], 4) for n in ("HET_rgat_aggregate_packs", "HET_rgat_aggregate_hubs",
                                                                                 "HET_rgat_aggregate_finish") if n in kt}
                    op_ms = (per_op or {}).get("rgat_aggregate_compact_runs")  # (entry to return on the caller's stream: see the backward)
                    if op_ms:
                        f_ms = op_ms
                        ex["frac_with_per_edge_row_gather"] = round(req / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roofline_fwd = hbm_view(f"{fname} (relational_fused_gat_separate_coo, kind {4 if S_row else 0}{'; also leaves the per-run sums grad_er is formed from' if isinstance(pmf, tuple) else ''})",
                                    f_ms, nf_, ex, pmc_name=pmf)
    if args.model == "rgat" and world == 1 and not use_dist and roofline is not None and roofline_fwd is not None and S_row is not None:
        ref = gather_rate_reference(g, dev, N_local, X)
        if ref:
            roofline["gather_rate_reference_ms"], roofline_fwd["gather_rate_reference_ms"] = ref["backward_ms"], ref["forward_ms"]
            roofline["gather_rate_reference_what"] = roofline_fwd["gather_rate_reference_what"] = ref["what"]
    if args.model == "rgcn" and "HET_segment_sum" in kt and not use_dist:
        roofline, roofline_fwd = rgcn_rooflines(g, kt, E_local, N_local, K, X, hbm_view)
    if args.model == "hgt" and "HET_hgt_aggregate_rows" in kt and not use_dist:
        roofline, roofline_fwd = hgt_rooflines(g, kt, E_local, N_local, K, X, H, hbm_view)
    kernel_ms = {k: {"avg_ms": round(v[0], 4), "launches_per_step": round(v[1], 2), "ms_per_step": round(v[2], 4)} for k, v in kt.items()}

    # the reference-named ops exactly as the reference's model code calls them (kind 0, [E,H,D] feat), each launched a
    # few times on its own after the timed region: these are the SURVEY 8(d) worked figures (a4 7.28 GB, a5 13.86 GB,
    # a1 172.9 GFLOP at C3).  Entry-point HIP events on the launch stream: the whole op (its fills and every kernel).
    roofline_ops = roofline_gemm = None
    if args.model == "rgat" and world == 1 and not use_dist:
        sc = g.get_separate_coo_original()
        d_src = {"separate_coo_rel_ptrs": sc["rel_ptrs"], "separate_coo_node_indices": sc["row_indices"],
                 "separate_coo_eids": sc["eids"]}
        Wp = torch.randn(g.get_num_rels(), H, K, X // H, device=dev) * 0.1
        retp = torch.empty(E_local, H, X // H, device=dev)
        HK.event_timers[mm_name] = []
        HL.kernel_timing(True)
        with torch.no_grad():
            for _ in range(6):
                HK.K.rgnn_relational_matmul(d_src, 0, Wp, embed.detach(), retp, True)
        torch.cuda.synchronize()
        HL.kernel_timing(False)
        HK.event_timers.pop(mm_name)
        ms, n = HL.kernel_timing_read("HET_seg_gemm_mfma<store>")
        g_ms = ms / max(1, n)
        flops = 2.0 * E_local * K * X
        tf = flops / (g_ms * 1e-3) / 1e12 if n else 0.0  # (n == 0: not a matrix-core shape -- the LDS-tiled FMA kernel ran)
        busy = pmc("HET_seg_gemm_mfma<64, 2, false, 0, false>", "mfma_busy_frac")
        roofline_gemm = None if not n else {"bound": "mfma", "kernel": "HET_seg_gemm_mfma (rgnn_relational_matmul, kind 0, E rows, K=X=%d)" % K,
                         "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "kernel_ms": round(g_ms, 4), "flops": flops,
                         "mfma_busy_frac_pmc": busy,
                         "traffic": pmc("HET_seg_gemm_mfma<64, 2, false, 0, false>", "hbm_bytes_per_launch"),
                         "traffic_source": f"profiles/{PROF_ROUND}/{args.variant}_pmc.json (committed; not measured in this run)" if busy else None}
        # a4 / a5 on the per-edge tensor retp just written (feat_src_per_edge), reference argument order
        el = torch.randn(E_local, H, device=dev)
        er = torch.randn(E_local, H, device=dev)
        sm, ex_, rt = torch.empty(N_local, H, device=dev), torch.empty(E_local, H, device=dev), torch.empty(N_local, H, X // H, device=dev)
        gfe, gel, ger = torch.empty_like(retp), torch.empty_like(el), torch.empty_like(er)
        fn, bn = "het_relational_fused_gat_separate_coo", "het_backward_relational_fused_gat_separate_coo"
        HK.event_timers[fn], HK.event_timers[bn] = [], []
        with torch.no_grad():
            for _ in range(5):
                HK.K.relational_fused_gat_separate_coo(sc["eids"], sc["rel_ptrs"], sc["row_indices"], sc["col_indices"], 0, {},
                                                       retp, el, er, sm, ex_, rt, 0.2)
                HK.K.backward_relational_fused_gat_separate_coo(sc["eids"], sc["rel_ptrs"], sc["row_indices"], sc["col_indices"],
                                                                0, {}, retp, el, er, sm, ex_, rt, go.view(N_local, H, X // H),
                                                                gfe, gel, ger, 0.2)
        torch.cuda.synchronize()
        f_ev, b_ev = HK.event_timers.pop(fn)[1:], HK.event_timers.pop(bn)[1:]
        f_ms = sum(a.elapsed_time(b) for a, b, _ in f_ev) / len(f_ev)
        b_ms = sum(a.elapsed_time(b) for a, b, _ in b_ev) / len(b_ev)
        fb, bb = gat_fwd_bytes(E_local, N_local, H, X), gat_bwd_bytes(E_local, N_local, H, X)
        roofline_ops = {
            "what": "the reference-named kind-0 ops on [E,H,D] inputs, one op per entry-point call (all its kernels and fills)",
            "a4 relational_fused_gat_separate_coo": {"op_ms": round(f_ms, 4), "algorithmic_bytes": fb,
                                                     "achieved_GBps": round(fb / f_ms / 1e6, 1), "frac": round(fb / f_ms / 1e6 / HBM_PEAK_GBS, 4)},
            "a5 backward_relational_fused_gat_separate_coo": {"op_ms": round(b_ms, 4), "algorithmic_bytes": bb,
                                                              "achieved_GBps": round(bb / b_ms / 1e6, 1), "frac": round(bb / b_ms / 1e6 / HBM_PEAK_GBS, 4)}}
        del Wp, retp, el, er, sm, ex_, rt, gfe, gel, ger

    dist_info = None
    if use_dist:
        # what every rank moved and waited for: a few more steps with the halo contexts' wait timers on
        import torch.distributed as dist
        halo = runner.dl.halo
        halo.timing = []
        tsteps = max(1, min(args.steps, 5))
        for _ in range(tsteps):
            step()
        barrier()
        waits = {}
        for what, a, b in halo.timing:
            waits[what] = waits.get(what, 0.0) + a.elapsed_time(b) / tsteps
        halo.timing = None
        p_ = runner.dl.plan
        mine = {"rank": rank, "owned_nodes": int(p_.n_own), "halo_rows_received": int(p_.n_halo), "halo_rows_sent": int(p_.send_idx.numel()),
                "halo_MB_sent_per_exchange": round(int(p_.send_idx.numel()) * K * 4 / 1e6, 2),
                "halo_MB_received_per_exchange": round(int(p_.n_halo) * K * 4 / 1e6, 2), "local_edges": int(E_local),
                "exposed_wait_ms": {k_: round(v_, 4) for k_, v_ in waits.items()}}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        dist_info = {"ranks": world, "backend": backend + (" (RCCL)" if backend == "nccl" else ""),
                     "overlap": os.environ.get("HET_DIST_OVERLAP", "1") == "1",
                     "pieces": int(p_.chunks),
                     "exchange": "x[halo] forward and grad_x[halo] backward per step, each as `pieces` all_to_all_single calls (piece c = "
                                 "the c-th part of the rows of every peer; the layer projects a piece's rows when it has landed); "
                                 "exposed_wait_ms = time the launch stream waited per piece (0 when hidden or synchronous)",
                     "per_rank": allr,
                     "max_exposed_wait_ms": round(max(sum(r_["exposed_wait_ms"].values()) for r_ in allr), 4)}

    if roofline is not None and roofline_ops is not None:
        # (the driver's record keeps the contract objects whole: the op-level figures of the reference-named ops ride in `roofline`)
        roofline["reference_named_ops"] = {k_: {"op_ms": v_["op_ms"], "frac": v_["frac"], "achieved_GBps": v_["achieved_GBps"],
                                                "algorithmic_bytes": v_["algorithmic_bytes"]}
                                           for k_, v_ in roofline_ops.items() if isinstance(v_, dict)}
    from het_amd import plan as HP
    plan_bytes = HP.cached_bytes()
    if rank == 0:
        out = {
            "metric": f"million edges/s (fwd+bwd) {args.model.upper()} layer, ogbn-mag feat=64",
            "value": round(value, 2), "unit": "million edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_step_median_events": round(median_ms, 4),
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model.upper()} layer fwd+bwd, ogbn-mag-shaped synthetic graph (N={N_global}, E={E_global}, R=4), "
                                   f"feat={K}, heads={H}, self_loop, no optimizer step, layer flags: {args.variant}; LAYER SWAP: the model "
                                   f"script builds het_amd.layers.{'HET_RGATLayer' if args.model == 'rgat' else 'HET_EglRelGraphConv_EdgeParallel' if args.model == 'rgcn' else 'HET_HGTLayerHetero'} "
                                   "(the reference's layer class name and arguments); swapping only hrt/python/backend or only "
                                   "kernels/__init__.py under the reference's own model code gives variants.op_by_op / "
                                   "variants.reference_op_sequence",
                       "edge_order": args.edge_order, "scale": args.scale, "device": torch.cuda.get_device_name(dev),
                       "torch": torch.__version__, "hip": torch.version.hip,
                       "layout_build_ms": None if layout_ms is None else round(layout_ms, 1),
                       "parallelism": "single GPU" if world == 1 else f"dst-range partition x{world}, RCCL all-to-all halo"},
            "roofline": roofline,
            "roofline_forward": roofline_fwd,
            "roofline_segment_gemm": roofline_gemm,
            "roofline_reference_named_ops": roofline_ops,
            "kernel_ms": kernel_ms,
            "per_op_ms": per_op,
            "dist": dist_info,
            # torch's allocator peak; the groupings are part of it when the library allocates through torch (het_set_allocator:
            # the default of het_amd.kernels), else hipMalloc'ed by the library and added here
            "peak_memory_GB": round((torch.cuda.max_memory_allocated(dev) + (0 if HL.allocator_is_external() else plan_bytes)) / 2**30, 2),
            "peak_memory_detail_GB": {"torch_allocator_peak": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                                      "groupings": round(plan_bytes / 2**30, 2),
                                      "groupings_inside_torch_allocator": bool(HL.allocator_is_external())},
        }
        if world == 1 and not args.no_variants and args.variant == "default" and args.model == "rgat":
            out["variants"] = other_variants(args, coo, dev, min(args.steps, 10), ms_per_step, value)
        if world == 1 and not use_dist and not args.no_models and args.model == "rgat" and args.feat % 8 == 0:
            # BASELINE.json configs[1] and configs[3] in the same line (about 10 s each)
            out["models"] = {m: other_model(m, args, coo, dev, world) for m in ("rgcn", "hgt")}
        if world == 1 and not use_dist and not args.no_variants and args.model == "rgat" and args.edge_order != "random":
            out["edge_order_random"] = edge_order_random(args, dev)
        if world == 1 and not use_dist and not args.no_dist_rehearsal and args.model == "rgat" and args.variant == "default":
            out["dist_rehearsal"] = dist_rehearsal(args, dev)
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
        wall["gpu_phase_end"] = _stamp()
        out["wall_clock_utc"] = wall
        print(json.dumps(out))
    if world > 1 or os.environ.get("HET_FORCE_DIST") == "1":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
