"""The whole RGAT layer (RGAT/models.py:16-385) as ONE autograd node.

Same ops, same values and gradients as the op-by-op composition in het_amd/layers.py (which follows the reference's
model code and stays the fallback); what the single node buys is control over the backward pass: every consumer of the
layer input accumulates its gradient into ONE buffer (self-loop first with plain stores, then the two projections with
their atomic epilogues) and the weight gradients into one buffer, instead of autograd allocating a gradient per
consumer and summing them with elementwise kernels; the output ``h + loop_message + h_bias`` is one pass.  On ogbn-mag
this removes ~1 ms of elementwise adds and fills per step.

Two dataflows:
  distinct rows (kinds 3 / 4)   projections on the unique (relation, node) rows, gathered by the GAT kernels through the
                           (relation, source) -> row map; el folded into the compact GAT backward.  This is what the
                           reference's --compact_as_of_node_flag selects AND what the default flags run on here: inside
                           one autograd node no per-edge tensor is visible to the caller, the values are the same
                           (feat_src_per_edge[e] == feat_compact[row(e)] bit for bit), and the [E,H,D] tensor and its
                           gradient (5.4 GB each on ogbn-mag) are never written.  The unique lists are the ones the
                           reference's RGAT script builds for every run (RGAT/train_dgl.py:160, unconditional); a graph
                           that lacks them gets them from the device-side builders once.
  per edge (kind 0)        the round-1 dataflow of the default flags: [E,H,D] projections (distinct-row projection +
                           broadcast), el / er / exp / grad_el in the destination-grouped order of the GAT kernels.  Kept
                           behind HET_RGAT_PER_EDGE=1 for A/B runs and as the path tests/ compare the other with.
and, for either, ``mulfirst`` (--multiply_among_weights_first_flag, RGAT/models.py:300-326): er = x[dst] . (W . attn_r)
as a row-dot product on the distinct (relation, destination) rows instead of a projection followed by a dot.  The two
forms of er are the same real number (associativity); the reference offers the flag because it is cheaper, and so the
node takes it whenever the one-head row-dot kernels cover the shape (HET_RGAT_LITERAL_ER=1 keeps (x . W) . attn_r).
"""
import os

import torch as th

from ..plan import consistent as _consistent_plan

from .. import kernels as _k
from ..kernels import K


_OFFS = {}
PER_EDGE = os.environ.get("HET_RGAT_PER_EDGE") == "1"      # default flags on the per-edge (kind 0) dataflow
LITERAL_ER = os.environ.get("HET_RGAT_LITERAL_ER") == "1"  # er = (x . W) . attn_r unless the layer flag asks otherwise
RUN_SUMS = os.environ.get("HET_RGAT_RUN_SUMS", "1") != "0"  # A/B switch: grad_er from the forward's run sums
OVERLAP = os.environ.get("HET_RGAT_OVERLAP", "1") != "0"  # independent launches on a second HIP stream (see _side_stream)
ATTN_GRAD_IN_PASS = os.environ.get("HET_RGAT_ATTN_GRAD_IN_PASS", "1") != "0"  # grad_attn_l from the source-row kernels
NODE_ORDER = os.environ.get("HET_RGAT_NODE_ORDER", "1") != "0"  # node-major pass over nodes sorted by relation presence
NODE_GEMM = os.environ.get("HET_RGAT_NODE_GEMM", "1") != "0"  # backward GEMMs per node (csrc/node_gemm.hip); 0: per relation
# A/B: the self-loop weight gradient with the other weight gradients beside the node-major pass instead of at the start of the
# backward, where it stretches the two short per-destination passes (profiles/r04/default_timeline.txt): 3.97 -> 4.02 ms, kept off
LOOP_DW_LATE = os.environ.get("HET_RGAT_LOOP_DW_LATE", "0") == "1"
# the bias gradient (column sums of grad_h) from the self-loop's weight-gradient launch, which streams grad_h anyway, instead of
# a pass of its own inside the gather op (0.088 ms on ogbn-mag).  HET_RGAT_BIAS_IN_DW=0: A/B
BIAS_IN_DW = os.environ.get("HET_RGAT_BIAS_IN_DW", "1") != "0"


def _mulfirst_shape_ok(H, Kd):
    """er = x[dst] . (W . attn_r) through the one-head row-dot kernels (seg_rowdot.hip)."""
    return H in (1, 2, 4, 8) and Kd >= 4 * H and Kd & (Kd - 1) == 0 and Kd <= 256


def _has_single_sided_lists(g):
    return "unique_node_indices_single_sided" in getattr(g, "graph_data", {}).get("separate", {})


def effective_flags(g, W, compact, direct, mulfirst):
    """(compact, direct, mulfirst) the node runs with for the layer flags given (see the module docstring)."""
    R, H, Kd, D = W.shape
    if not compact and not PER_EDGE and (_has_single_sided_lists(g) or hasattr(g, "generate_separate_unique_node_indices_single_sided_for_each_etype")):
        compact, direct = True, True
    if not mulfirst and not LITERAL_ER and _mulfirst_shape_ok(H, Kd):
        mulfirst = True
    return bool(compact), bool(direct), bool(mulfirst)


def _destinations_below(col, nd):
    """Whether every destination id is < nd (cached per graph: one reduction + host read the first time)."""
    return col.numel() == 0 or _k._derived_get("col_max", (col,), lambda: int(col.max())) < nd


def _lists(g):
    s = g.get_separate_coo_original()
    by_src = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"], "separate_coo_eids": s["eids"]}
    by_dst = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"], "separate_coo_eids": s["eids"]}
    return s, by_src, by_dst


def _compact_dicts(g, direct):
    ss = g.get_separate_unique_node_indices_single_sided()
    if direct:
        inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
        d = {"edata_idx_to_inverse_idx_row": inv["inverse_indices_row"], "edata_idx_to_inverse_idx_col": inv["inverse_indices_col"]}
        return ss, 4, d, d
    fwd = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"], "unique_srcs_and_dests_rel_ptrs_col": ss["rel_ptrs_col"],
           "unique_srcs_and_dests_node_indices_row": ss["node_indices_row"],
           "unique_srcs_and_dests_node_indices_col": ss["node_indices_col"]}
    bwd = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"], "unique_srcs_and_dests_rel_col": ss["rel_ptrs_col"],
           "unique_srcs_and_dests_node_indices_row": ss["node_indices_row"],
           "unique_srcs_and_dests_node_indices_col": ss["node_indices_col"]}
    return ss, 3, fwd, bwd


def _edge_rows(g, ss, direct, rp, row, col, eids):
    """(feat row, er row) of every edge position, [E] int64 each: the graph's inverse indices when it carries them
    (indexed by edata id == position after canonicalize_eids), else located in the unique lists once and cached."""
    sep = getattr(g, "graph_data", {}).get("separate", {}).get("unique_node_indices_single_sided", {})
    if "inverse_indices_row" in sep and getattr(g, "sequential_eids_format", None) == "separate_coo":
        return sep["inverse_indices_row"], sep["inverse_indices_col"]
    maps = (ss["rel_ptrs_row"], ss["node_indices_row"], ss["rel_ptrs_col"], ss["node_indices_col"])
    return _k._src_rows_by_position(3, maps, rp, row, eids), _k._dst_rows_by_position(3, maps, rp, col, eids)


def rgat_layer_fused_ok(g, x, W, slope, compact, mulfirst=False):
    """Shapes / state for which every op of the node runs on its fast path (else use the op-by-op composition)."""
    R, H, Kd, D = W.shape
    if not (_k._plan.is_enabled() and x.is_cuda and x.dim() == 2 and slope >= 0 and g.get_num_edges() > 0
            and _k.gat_grouped_shape_ok(H, D)):
        return False
    compact, _, mulfirst_eff = effective_flags(g, W, compact, True, mulfirst)
    if mulfirst and not _mulfirst_shape_ok(H, Kd):
        return False
    mulfirst = mulfirst_eff
    if not _k.matmul_attn_dot_ok(H, Kd, D):
        # widths the matrix-core projection does not take (e.g. the 8 output classes of the reference CLI's defaults): the
        # distinct-row dataflow with er from the folded weight runs on the any-shape GEMM + a row-dot for el
        return compact and mulfirst and H & (H - 1) == 0
    if compact or mulfirst:
        return True
    _, _, by_dst = _lists(g)
    return _k.matmul_attn_dot_only_ok(by_dst, W, x)


_SIDE = {}


def _side_stream(dev):
    """A second HIP stream per device for launches that do not depend on each other and are bound by different units: the
    weight-gradient passes (HBM-bound: they stream x / feat_c / gradient rows once) beside the node-major input-gradient pass
    (matrix-core-bound), the self-loop GEMM (HBM-bound) beside the projection GEMM.  The caller brackets the side work with
    events: it starts after everything it reads and the main stream waits for it before anything reads its outputs (or frees
    its inputs), so the allocator never sees a cross-stream use."""
    s = _SIDE.get(dev)
    if s is None:
        s = _SIDE[dev] = th.cuda.Stream(device=dev)
    return s


PIECEWISE_PUSH = os.environ.get("HET_DIST_PIECEWISE", "1") != "0"  # project the halo rows piece by piece as they arrive


def _halo_pieces(g, ss, plan):
    """The unique (relation, source) list of a partition's local graph cut by when the source node's row of x is there: the
    owned nodes together with piece 0 of the halo exchange, then every further piece (local ids n_own + halo_chunk_ptr[c] ..).
    Per cut a relation-bucketed list (rel_ptrs [R+1], node ids, row index in the full list), built once per graph."""
    key = ("halo_pieces", plan.chunks, plan.n_own, plan.n_halo)
    hit = g._plans.get(key)
    if hit is None:
        rp_row, nodes = ss["rel_ptrs_row"], ss["node_indices_row"]
        R = rp_row.numel() - 1
        rel = th.repeat_interleave(th.arange(R, device=nodes.device), rp_row[1:] - rp_row[:-1])
        # (the owned sources -- 5 % of a rank's rows on ogbn-mag at 8 ranks -- go with the first piece: a launch less per step, and
        #  a rank's step at 8 ranks is short enough for every launch to count, profiles/r04/dist_rank_share_halo.txt)
        cuts = [(0, plan.n_own + plan.halo_chunk_ptr[1])] + [(plan.n_own + plan.halo_chunk_ptr[c], plan.n_own + plan.halo_chunk_ptr[c + 1])
                                                             for c in range(1, plan.chunks)]
        hit = []
        for a, b in cuts:
            sel = th.nonzero((nodes >= a) & (nodes < b)).flatten()  # ascending: still relation-major
            rp_c = th.zeros(R + 1, dtype=th.int64, device=nodes.device)
            rp_c[1:] = th.cumsum(th.bincount(rel[sel], minlength=R), 0)
            hit.append((rp_c, nodes[sel].contiguous(), sel.contiguous()))
        assert sum(int(h[2].numel()) for h in hit) == nodes.numel()
        g._plans[key] = hit
    return hit


@_consistent_plan
class RgatLayerFunction(th.autograd.Function):
    @staticmethod
    def forward(ctx, g, compact, direct, mulfirst, slope, num_dst, halo, x, W, attn_l, attn_r, loop_w, bias):
        x, W, attn_l, attn_r = x.contiguous(), W.contiguous(), attn_l.contiguous(), attn_r.contiguous()
        if halo is not None:
            # multi-GPU (het_amd/dist.py): x holds the owned rows; the halo rows of x_local arrive through an all-to-all
            # that is in flight until halo.finish_push() -- everything before that reads owned rows only
            assert compact and mulfirst and loop_w is not None, "rgat_layer_halo_ok guards this path"
            x = halo.start_push(x)
        s, by_src, by_dst = _lists(g)
        rp, row, col, eids = s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"]
        E, N = eids.numel(), x.shape[0]
        R, H, Kd, D = W.shape
        X = H * D
        new = lambda *shape: th.empty(shape, dtype=x.dtype, device=x.device)
        sm, ret = new(N, H), new(N, H, D)
        nd = N if num_dst is None else min(int(num_dst), N)
        offs = h = None
        if loop_w is not None:
            loop_w = loop_w.contiguous()
            offs = _OFFS.get((nd, x.device))  # built once per (rows, device): no per-step host-to-device copy
            if offs is None:
                if len(_OFFS) > 64:
                    _OFFS.clear()
                offs = _OFFS[(nd, x.device)] = th.tensor([0, nd], dtype=th.int64, device=x.device)
        wa = None
        if mulfirst:  # RGAT/models.py:300-326: the attention vector folded into the weight, [R,H,K,1]
            wa = th.bmm(W.view(-1, Kd, D), attn_r.view(-1, D, 1)).view(R, H, Kd, 1)
        if compact:
            ss = g.get_separate_unique_node_indices_single_sided()
            d_row = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"], "unique_srcs_and_dests_node_indices": ss["node_indices_row"]}
            d_col = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"], "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}
            featc = new(ss["node_indices_row"].numel(), H, D)
            elc = new(featc.shape[0], H)
            erc = new(ss["node_indices_col"].numel(), H)
            # (the destination side and the self-loop read rows of destination nodes only -- owned rows on a partition)
            side = None
            fused_loop = loop_w is not None and _k.rows_linear_bias_ok(Kd, X)
            # (the three products below read the same rows of x; one node-major pass that reads them once was built and measured in
            #  round 5 -- 1 GB less traffic, the same step time: exp/node_fwd.hip.txt)
            if OVERLAP and halo is None and fused_loop and mulfirst:
                # er_c (a row-dot) and the self-loop GEMM are HBM-bound streams of rows: on the side stream beside the projection
                main, side = th.cuda.current_stream(x.device), _side_stream(x.device)
                # everything the side stream WRITES is allocated before the fork: a block handed out later may have been freed by
                # a tensor whose last main-stream kernel was enqueued after the fork -- the side stream would not wait for it
                h = x.new_empty((nd, X))
                side.wait_stream(main)
            if mulfirst:
                with th.cuda.stream(side if side is not None else th.cuda.current_stream(x.device)):
                    K.rgnn_relational_matmul(d_col, 1, wa, x, erc.view(-1, H, 1), True)
                saved = (featc, elc, erc)
            else:
                featd = new(erc.shape[0], H, D)
                _k.matmul_attn_dot(d_col, 1, W, x, featd, attn_r, erc)
                saved = (featc, elc, erc, featd)
            if fused_loop:
                # self-loop + bias first (bias in the GEMM epilogue); the aggregation adds its rows into h in place: no
                # separate h = ret + loop + bias pass and no zero fill of ret (read by the backward only where edges point)
                bias_c = None if bias is None else bias.contiguous()
                if side is not None:  # (h: allocated, and later freed, under the main stream; the side stream only fills it)
                    with th.cuda.stream(side):
                        _k.rows_linear_bias(offs, x[:nd], loop_w, bias_c, out=h)
                else:
                    h = _k.rows_linear_bias(offs, x[:nd], loop_w, bias_c)
            dot_ok = _k.matmul_attn_dot_ok(H, Kd, D)
            if halo is not None and halo.chunks > 1 and PIECEWISE_PUSH and dot_ok:
                # the exchange arrives in pieces (het_amd/dist.py: DistPlan.chunks): the rows whose source node is owned are
                # projected at once, the rows of piece c as soon as piece c is there -- piece c + 1 is on the wire meanwhile
                for c, (rp_c, nodes_c, rows_c) in enumerate(_halo_pieces(g, ss, halo.plan)):
                    halo.wait_push_piece(c)
                    _k.matmul_attn_dot_rows(rp_c, nodes_c, rows_c, W, x, featc, attn_l, elc)
                halo.finish_push()
            elif halo is not None:
                halo.finish_push()
            if halo is not None and halo.chunks > 1 and PIECEWISE_PUSH and dot_ok:
                pass  # (projected above)
            elif dot_ok:
                _k.matmul_attn_dot(d_row, 1, W, x, featc, attn_l, elc)  # el_c = <feat_c, attn_l[r]> from the GEMM epilogue
            else:  # other widths: any-shape projection, then el_c as a row-dot over the relation-bucketed rows
                K.rgnn_relational_matmul(d_row, 1, W, x, featc, True)
                K.rgnn_relational_matmul_no_scatter_gather_list(ss["rel_ptrs_row"], attn_l.unsqueeze(-1), featc, elc.view(-1, H, 1))
            if side is not None:
                th.cuda.current_stream(x.device).wait_stream(side)
            # edge softmax + aggregation straight from the compact tables: no exp [E,H] tensor (csrc/gat_compact.hip)
            srow, drow = _edge_rows(g, ss, direct, rp, row, col, eids)
            # (run sums: grad_er from S_col rows the forward leaves instead of a per-edge term -- csrc/gat_compact.hip)
            run_sums = RUN_SUMS and _k.rgat_runs_shape_ok(H, D)
            grp = _k.rgat_compact_groupings(col, srow, drow, N, featc.shape[0], erc.shape[0], rel_ptrs=rp if run_sums else None,
                                            drow_nodes=ss["node_indices_col"], drow_rel_ptrs=ss["rel_ptrs_col"])
            # (elc IS <featc, attn_l[relation of the row]>: the pass may form it from the rows it gathers -- kernels.py)
            ctx.runs = _k.rgat_aggregate_compact(grp, featc, elc, erc, sm, ret, slope, h_inout=h, num_rels=R,
                                                 attn_l=attn_l.contiguous() if run_sums else None, feat_rel_ptrs=ss["rel_ptrs_row"] if run_sums else None)
            ctx.grp = grp
            ex = x.new_empty(0)
        else:
            # el / er are produced directly in the destination-grouped order of the GAT kernels (rank of every position):
            # the aggregation pass forms exp from two coalesced streams, no exp pass, no per-edge 16-byte gathers
            rank = _k.gat_rank_of_position(rp, row, col, eids, N)
            by_dst_s = {"separate_coo_rel_ptrs": rp, "separate_coo_node_indices": col, "separate_coo_eids": rank}
            feat, el_s, er_s, exs = new(E, H, D), new(E, H), new(E, H), new(E, H)
            _k.matmul_attn_dot(by_src, 0, W, x, feat, attn_l, el_s, dot_rows=rank)
            if mulfirst:
                K.rgnn_relational_matmul(by_dst_s, 0, wa, x, er_s.view(E, H, 1), True)
                saved = (feat, exs)
            else:
                comp = _k.matmul_attn_dot(by_dst_s, 0, W, x, None, attn_r, er_s)
                assert comp is not None
                saved = (feat, exs, comp)
            used = _k.fused_gat_forward(eids, rp, row, col, 0, {}, feat, None, None, sm, None, ret, slope, exs,
                                        el_sorted=el_s, er_sorted=er_s)
            assert used
            ex = x.new_empty(0)
        if h is None:
            out = ret.view(N, X)[:nd]
            loop = None
            if loop_w is not None:
                loop = new(nd, X)
                K.rgnn_relational_matmul_no_scatter_gather_list(offs, loop_w.view(1, 1, Kd, X), x[:nd], loop)
            h = _k.rows_add_bias(out, loop, None if bias is None else bias.contiguous()) if (loop is not None or bias is not None) else out.clone()
        ctx.halo = halo
        ctx.g, ctx.compact, ctx.mulfirst, ctx.slope, ctx.nd = g, compact, mulfirst, slope, nd
        ctx.has_loop, ctx.has_bias = loop_w is not None, bias is not None
        ctx.save_for_backward(x, W, attn_l, attn_r, loop_w if loop_w is not None else x.new_empty(0), offs if offs is not None else eids,
                              sm, ex, ret, *saved)
        return h

    @staticmethod
    def backward(ctx, grad_h):
        x, W, attn_l, attn_r, loop_w, offs, sm, ex, ret, *saved = ctx.saved_tensors
        g, nd, slope = ctx.g, ctx.nd, ctx.slope
        s, by_src, by_dst = _lists(g)
        rp, row, col, eids = s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"]
        N, Kd = x.shape
        E = eids.numel()
        R, H, _, D = W.shape
        X = H * D
        grad_h = grad_h.contiguous()
        if ctx.halo is not None:
            return RgatLayerFunction._backward_with_halo(ctx, grad_h)
        grad_bias = grad_h.sum(0) if (ctx.has_bias and not ctx.compact) else None
        Wt = th.transpose(W, 2, 3).contiguous()
        if ctx.compact and ctx.mulfirst and NODE_GEMM and _k.rgat_node_gemm_ok(R, H, Kd, D) and _destinations_below(col, nd):
            return RgatLayerFunction._backward_node_major(ctx, grad_h, Wt)
        grad_W = th.zeros_like(W)
        # one input-gradient buffer: the self-loop writes its rows with plain stores, the projections add to it
        grad_loop = None
        if ctx.has_loop:
            grad_x = th.empty_like(x) if nd == N else th.zeros_like(x)
            grad_loop = th.empty_like(loop_w)
            _k.matmul_no_scatter_gather_backward(offs, loop_w.view(1, 1, Kd, X).transpose(2, 3).contiguous(), x[:nd], grad_h,
                                                 grad_x[:nd], grad_loop.view(1, 1, Kd, X), accumulate=False)
        else:
            grad_x = th.zeros_like(x)
        # every edge points at one of the first nd nodes (blocks, partitions: checked once per graph)?  then the
        # per-destination tensors of the distinct-row backward are their first nd rows
        dst_prefix = ctx.compact and _destinations_below(col, nd)
        if nd == N or dst_prefix:
            go = grad_h.view(nd, H, D)
        else:  # rows of non-destination nodes receive no gradient
            go = th.zeros((N, H, D), dtype=x.dtype, device=x.device)
            go.view(N, X)[:nd] = grad_h
        ndp = nd if dst_prefix else N
        mulfirst = ctx.mulfirst
        if mulfirst:
            wa_t = th.bmm(W.view(-1, Kd, D), attn_r.view(-1, D, 1)).view(R, H, 1, Kd)  # [R,H,K,1] transposed(2,3): same memory
            grad_wa = th.zeros((R, H, Kd, 1), dtype=x.dtype, device=x.device)
        if ctx.compact:
            featc, elc, erc = saved[:3]
            featd = None if mulfirst else saved[3]
            ss = g.get_separate_unique_node_indices_single_sided()
            d_row = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"], "unique_srcs_and_dests_node_indices": ss["node_indices_row"]}
            d_col = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"], "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}
            g_featc, g_elc, g_erc = th.empty_like(featc), th.empty_like(elc), th.empty_like(erc)  # all three overwritten
            if ctx.has_bias:  # the bias gradient (column sums of grad_h) from the pass that reads every gradout row anyway
                grad_bias = th.empty(X, dtype=x.dtype, device=x.device)
            _k.rgat_backward_compact(ctx.grp, featc, elc, erc, sm[:ndp], ret[:ndp], go, g_featc, g_elc, g_erc, slope, fold_attn_l=attn_l,
                                     row_rel_ptrs=ss["rel_ptrs_row"], grad_bias=grad_bias if ctx.has_bias else None, bias_rows=nd,
                                     runs=ctx.runs, drow_nodes=ss["node_indices_col"])
            grad_attn_l, grad_attn_r = th.empty_like(attn_l), th.empty_like(attn_r)
            _k.matmul_no_scatter_gather_backward(ss["rel_ptrs_row"], attn_l.unsqueeze(2), featc, g_elc, None,
                                                 grad_attn_l.unsqueeze(-1), accumulate=False)
            _k.matmul_backward(d_row, 1, Wt, x, g_featc, grad_x, grad_W, True, accumulate=True, distinct_rows=True)
            if mulfirst:
                _k.matmul_backward(d_col, 1, wa_t, x, g_erc.view(-1, H, 1), grad_x, grad_wa, True, accumulate=True, distinct_rows=True)
            else:
                g_featd = th.empty_like(featd)
                _k.matmul_no_scatter_gather_backward(ss["rel_ptrs_col"], attn_r.unsqueeze(2), featd, g_erc, g_featd,
                                                     grad_attn_r.unsqueeze(-1), accumulate=False)
                _k.matmul_backward(d_col, 1, Wt, x, g_featd, grad_x, grad_W, True, accumulate=True, distinct_rows=True)
        else:
            feat, exs = saved[:2]
            # the edges' grad_el (= grad_er) in the kernel's destination-grouped order: sequential stores, and the
            # (relation, destination) sums of the er side read contiguous runs instead of scattered 16-byte pieces
            rank = _k.gat_rank_of_position(rp, row, col, eids, N)
            by_dst = {"separate_coo_rel_ptrs": rp, "separate_coo_node_indices": col, "separate_coo_eids": rank}
            g_feat, g_el = th.empty_like(feat), th.empty_like(exs)
            grad_attn_l = th.zeros_like(attn_l)
            if R <= 8:
                _k.fused_gat_backward(eids, rp, row, col, 0, {}, feat, None, None, sm, None, ret, go, g_feat, None, None, slope,
                                      exs, fold_attn_l=attn_l, grad_fold_attn_l=grad_attn_l, grad_el_sorted=g_el)
            else:
                g_el_e = th.empty_like(exs)
                _k.fused_gat_backward(eids, rp, row, col, 0, {}, feat, None, None, sm, None, ret, go, g_feat, g_el_e, g_el_e, slope,
                                      exs, fold_attn_l=attn_l, grad_el_sorted=g_el)
                by_eid = {"separate_coo_rel_ptrs": rp, "separate_coo_node_indices": eids, "separate_coo_eids": eids}
                _k.matmul_backward(by_eid, 0, attn_l.unsqueeze(2), feat, g_el_e, None, grad_attn_l.unsqueeze(-1), False,
                                   accumulate=False)
            _k.matmul_backward(by_src, 0, Wt, x, g_feat, grad_x, grad_W, True, accumulate=True)
            if mulfirst:
                _k.matmul_backward(by_dst, 0, wa_t, x, g_el.view(E, H, 1), grad_x, grad_wa, True, accumulate=True)
            else:
                grad_attn_r = th.zeros_like(attn_r)
                ok = _k.matmul_attn_dot_only_backward(by_dst, Wt, x, attn_r, g_el, grad_x, grad_W, comp_rows=saved[2],
                                                      grad_dot_w=grad_attn_r, accumulate=True)
                assert ok, "the grouping of the forward pass is gone"
        if mulfirst:  # through wa[r,h,k] = SUM_d W[r,h,k,d] * attn_r[r,h,d]
            grad_W.addcmul_(grad_wa, attn_r.view(R, H, 1, D))
            grad_attn_r = (W * grad_wa).sum(2)
        return None, None, None, None, None, None, None, grad_x, grad_W, grad_attn_l, grad_attn_r, grad_loop, grad_bias


    @staticmethod
    def _backward_node_major(ctx, grad_h, Wt):
        """The backward on the distinct-row dataflow with every term of the INPUT gradient gathered per node
        (csrc/node_gemm.hip): one pass stores grad_x (self-loop + relation projections + the folded attention vector) -- instead
        of a self-loop pass and one read-modify-write launch per relation and side."""
        x, W, attn_l, attn_r, loop_w, offs, sm, ex, ret, featc, elc, erc = ctx.saved_tensors
        g, nd, slope = ctx.g, ctx.nd, ctx.slope
        N, Kd = x.shape
        R, H, _, D = W.shape
        X = H * D
        ss = g.get_separate_unique_node_indices_single_sided()
        rp_row = ss["rel_ptrs_row"]
        row_map = _k.node_row_map(rp_row, ss["node_indices_row"], N)
        dst_map = _k.node_row_map(ss["rel_ptrs_col"], ss["node_indices_col"], N)
        go = grad_h.view(nd, H, D)  # every edge points at one of the first nd nodes (checked by the caller)
        # (the weight gradient of attn_l from the same pass when the forward left run sums: csrc/gat_compact.hip ga_block_reduce;
        #  grad_el_c then has no reader left -- its other consumer, the gradient through el, is folded into grad_feat_c -- and is
        #  not written at all)
        attn_in_pass = ATTN_GRAD_IN_PASS and ctx.runs is not None and R <= 8
        g_featc, g_erc = th.empty_like(featc), th.empty_like(erc)  # overwritten
        g_elc = None if (attn_in_pass and _k._lib.has("het_grouping_note_stream")) else th.empty_like(elc)  # (a round-5 library)
        grad_bias = th.empty(X, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        grad_loop = th.empty_like(loop_w) if ctx.has_loop else None
        # (the self-loop product names each of the nd output rows once: the column sums of its gradout rows ARE the bias gradient)
        # (offs = [0, nd] by construction in forward())
        bias_in_dw = BIAS_IN_DW and ctx.has_bias and ctx.has_loop and _k._lib.has("het_rows_matmul_backward_dw_colsum")
        main, side = th.cuda.current_stream(x.device), _side_stream(x.device) if OVERLAP else None
        if side is not None and ctx.has_loop and not LOOP_DW_LATE:
            # the self-loop weight gradient needs x and grad_h only: an HBM-bound stream of rows beside the gather passes below
            side.wait_stream(main)
            with th.cuda.stream(side):
                _k.rows_matmul_backward_dw(offs, None, x[:nd], grad_h, grad_loop.view(1, 1, Kd, X), accumulate=False,
                                           colsum=grad_bias if bias_in_dw else None)
        grad_attn_l = th.empty_like(attn_l)
        _k.rgat_backward_compact(ctx.grp, featc, elc, erc, sm[:nd], ret[:nd], go, g_featc, g_elc, g_erc, slope, fold_attn_l=attn_l,
                                 row_rel_ptrs=rp_row, grad_bias=None if bias_in_dw else grad_bias, bias_rows=nd, runs=ctx.runs,
                                 drow_nodes=ss["node_indices_col"], grad_attn_l=grad_attn_l if attn_in_pass else None)
        wa_t = th.bmm(W.view(-1, Kd, D), attn_r.view(-1, D, 1)).view(R, H, Kd)  # wa[r,h,:] = W[r,h] . attn_r[r,h]
        grad_x = th.empty_like(x)
        grad_W, grad_wa = th.empty_like(W), th.empty((R, H, Kd), dtype=x.dtype, device=x.device)
        gh = grad_h if ctx.has_loop else None
        loop_wt = loop_w.t().contiguous() if ctx.has_loop else None
        d_col = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"], "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}

        def weight_gradients():
            # per product (four launches; each reads its own rows of x / feat_c -- a node-major pass that reads x once was
            # measured in five forms and lost: it multiplies zero rows wherever a node has no row in a relation, exp/node_dw.hip.txt)
            if not attn_in_pass:
                _k.matmul_no_scatter_gather_backward(rp_row, attn_l.unsqueeze(2), featc, g_elc, None, grad_attn_l.unsqueeze(-1),
                                                     accumulate=False)
            if ctx.has_loop and (side is None or LOOP_DW_LATE):
                _k.rows_matmul_backward_dw(offs, None, x[:nd], grad_h, grad_loop.view(1, 1, Kd, X), accumulate=False,
                                           colsum=grad_bias if bias_in_dw else None)
            _k.rows_matmul_backward_dw(rp_row, ss["node_indices_row"], x, g_featc.view(-1, X), grad_W, accumulate=False)
            _k.matmul_backward(d_col, 1, wa_t.view(R, H, 1, Kd), x, g_erc.view(-1, H, 1), None, grad_wa.view(R, H, Kd, 1), True,
                               accumulate=False)

        # (on a block only the first nd nodes carry the self-loop term: they stay in front, so that term's tiles are whole too)
        order = _k.node_order_by_presence(row_map, dst_map, split=nd if nd < N else None) if NODE_ORDER else None

        def input_gradient():
            _k.rgat_node_backward_dx(0, N, nd, gh, loop_wt, g_featc.view(-1, X), Wt, row_map, g_erc, wa_t, dst_map, grad_x,
                                     node_order=order)
        if side is not None:
            # the weight gradients (HBM-bound streams of rows) on the side stream while the node-major pass (matrix-core-bound)
            # runs on this one; both read g_featc / g_erc / grad_h, neither writes what the other reads
            side.wait_stream(main)
            with th.cuda.stream(side):
                weight_gradients()
            input_gradient()
            main.wait_stream(side)
        else:
            input_gradient()
            weight_gradients()
        grad_W.addcmul_(grad_wa.unsqueeze(-1), attn_r.view(R, H, 1, D))  # through wa[r,h,k] = SUM_d W[r,h,k,d] * attn_r[r,h,d]
        grad_attn_r = (W * grad_wa.unsqueeze(-1)).sum(2)
        return None, None, None, None, None, None, None, grad_x, grad_W, grad_attn_l, grad_attn_r, grad_loop, grad_bias

    @staticmethod
    def _backward_with_halo(ctx, grad_h):
        """The backward on a partition, ordered around the reverse halo exchange: the input gradient through the
        (relation, source) projection -- the only one that reaches halo rows -- is formed first and its halo rows leave with
        an all-to-all; the weight gradients, the destination side and the rest of the self-loop run while it is in flight;
        the returned rows are added to the owners' gradients at the end."""
        x, W, attn_l, attn_r, loop_w, offs, sm, ex, ret, featc, elc, erc = ctx.saved_tensors
        g, nd, slope, halo = ctx.g, ctx.nd, ctx.slope, ctx.halo
        N, Kd = x.shape
        R, H, _, D = W.shape
        X = H * D
        ss = g.get_separate_unique_node_indices_single_sided()
        rp_row, rows_node = ss["rel_ptrs_row"], ss["node_indices_row"]
        d_col = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"], "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}
        Wt = th.transpose(W, 2, 3).contiguous()
        s_coo = g.get_separate_coo_original()
        assert _destinations_below(s_coo["col_indices"], nd), "a partition's edges point at owned nodes"
        go = grad_h.view(nd, H, D)  # (so the per-destination tensors of the backward are their first nd rows)
        g_featc, g_elc, g_erc = th.empty_like(featc), th.empty_like(elc), th.empty_like(erc)
        grad_bias = th.empty(X, dtype=x.dtype, device=x.device) if ctx.has_bias else None
        wa_t = th.bmm(W.view(-1, Kd, D), attn_r.view(-1, D, 1)).view(R, H, 1, Kd)
        grad_x = th.empty_like(x)
        grad_W, grad_loop = th.empty_like(W), th.empty_like(loop_w)
        grad_attn_l = th.empty_like(attn_l)
        node_major = NODE_GEMM and _k.rgat_node_gemm_ok(R, H, Kd, D)
        # as on one GPU (_backward_node_major): the weight gradients are HBM-bound streams of rows -- on the side stream beside the
        # gather passes and the matrix-core-bound node pass; the self-loop's needs x and grad_h only and starts at once
        main, side = th.cuda.current_stream(x.device), (_side_stream(x.device) if OVERLAP and x.is_cuda else None)
        bias_in_dw = BIAS_IN_DW and ctx.has_bias and _k._lib.has("het_rows_matmul_backward_dw_colsum")  # (as in _backward_node_major)
        if side is not None:
            side.wait_stream(main)
            with th.cuda.stream(side):
                _k.rows_matmul_backward_dw(offs, None, x[:nd], grad_h, grad_loop.view(1, 1, Kd, X), accumulate=False,
                                           colsum=grad_bias if bias_in_dw else None)
        if not node_major:
            grad_x[nd:].zero_()  # halo rows: only the projection's input gradient adds to them
            _k.rows_matmul_backward_dx(offs, None, loop_w.t().contiguous().view(1, 1, X, Kd), grad_h, grad_x[:nd], atomic=False)
        attn_in_pass = ATTN_GRAD_IN_PASS and ctx.runs is not None and R <= 8 and x.is_cuda
        _k.rgat_backward_compact(ctx.grp, featc, elc, erc, sm[:nd], ret[:nd], go, g_featc, g_elc, g_erc, slope, fold_attn_l=attn_l,
                                 row_rel_ptrs=rp_row, grad_bias=None if bias_in_dw else grad_bias, bias_rows=nd, runs=ctx.runs,
                                 drow_nodes=ss["node_indices_col"], grad_attn_l=grad_attn_l if attn_in_pass else None)
        grad_wa = th.empty((R, H, Kd, 1), dtype=x.dtype, device=x.device) if node_major else th.zeros((R, H, Kd, 1), dtype=x.dtype, device=x.device)

        def weight_gradients():
            if side is None:
                _k.rows_matmul_backward_dw(offs, None, x[:nd], grad_h, grad_loop.view(1, 1, Kd, X), accumulate=False,
                                           colsum=grad_bias if bias_in_dw else None)
            _k.rows_matmul_backward_dw(rp_row, rows_node, x, g_featc.view(-1, X), grad_W, accumulate=False)
            if not attn_in_pass:
                _k.matmul_no_scatter_gather_backward(rp_row, attn_l.unsqueeze(2), featc, g_elc, None, grad_attn_l.unsqueeze(-1),
                                                     accumulate=False)
            if node_major:  # the er side's weight gradient alone (its input gradient is a term of the node pass)
                _k.matmul_backward(d_col, 1, wa_t, x, g_erc.view(-1, H, 1), None, grad_wa, True, accumulate=False)

        if node_major:
            # one pass per node range (csrc/node_gemm.hip): the halo rows first -- only the (relation, source) projections reach
            # them -- so that they leave with the all-to-all while the owned rows (self-loop + projections + folded attention
            # vector, every term in one store) and the weight gradients are formed
            row_map = _k.node_row_map(rp_row, rows_node, N)
            dst_map = _k.node_row_map(ss["rel_ptrs_col"], ss["node_indices_col"], N)
            loop_wt = loop_w.t().contiguous()
            order = _k.node_order_by_presence(row_map, dst_map, split=nd) if NODE_ORDER else None
            args = (grad_h, loop_wt, g_featc.view(-1, X), Wt, row_map, g_erc, wa_t.view(R, H, Kd), dst_map, grad_x, order)
            _k.rgat_node_backward_dx(nd, N, nd, *args)
            halo.start_return(grad_x)
            if side is not None:
                side.wait_stream(main)  # (g_featc / g_erc are complete)
                with th.cuda.stream(side):
                    weight_gradients()
            _k.rgat_node_backward_dx(0, nd, nd, *args)
            if side is None:
                weight_gradients()
        else:
            _k.rows_matmul_backward_dx(rp_row, rows_node, Wt, g_featc.view(-1, X), grad_x, atomic=2)  # rows of a relation: distinct nodes
            halo.start_return(grad_x)
            weight_gradients()
            # the er side's weight gradient and, on the per-relation path, its input gradient (owned rows only)
            _k.matmul_backward(d_col, 1, wa_t, x, g_erc.view(-1, H, 1), grad_x, grad_wa, True, accumulate=True, distinct_rows=True)
        if side is not None:
            main.wait_stream(side)
        grad_W.addcmul_(grad_wa, attn_r.view(R, H, 1, D))  # through wa[r,h,k] = SUM_d W[r,h,k,d] * attn_r[r,h,d]
        grad_attn_r = (W * grad_wa).sum(2)
        grad_own = halo.finish_return(grad_x[:nd])
        return None, None, None, None, None, None, None, grad_own, grad_W, grad_attn_l, grad_attn_r, grad_loop, grad_bias


def rgat_layer_halo_ok(g, x_own, W, slope, compact, mulfirst=False):
    """Whether the one-node layer can run the halo exchange itself (forward_with_halo): the distinct-row dataflow with er
    from the folded weight, and the matrix-core shapes of the split backward GEMMs."""
    R, H, Kd, D = W.shape
    if not rgat_layer_fused_ok(g, x_own, W, slope, compact, mulfirst):
        return False
    c, _, m = effective_flags(g, W, compact, True, mulfirst)
    return c and m and _k.rows_linear_bias_ok(Kd, H * D) and _k.rows_matmul_backward_split_ok(H, Kd, D)


def rgat_layer_fused(g, x, W, attn_l, attn_r, loop_w, bias, slope, compact, direct, num_dst=None, mulfirst=False, halo=None):
    compact, direct, mulfirst = effective_flags(g, W, compact, direct, mulfirst)
    if compact and not _has_single_sided_lists(g):
        g.generate_separate_unique_node_indices_single_sided_for_each_etype()
    return RgatLayerFunction.apply(g, compact, direct, mulfirst, float(slope), num_dst, halo, x, W, attn_l, attn_r, loop_w, bias)
