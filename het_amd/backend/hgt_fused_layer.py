"""HGT attention + aggregation (HGT/models.py:120-262) as ONE autograd node on the distinct (relation, source) rows.

The op-by-op composition in het_amd/layers.py follows the reference's model code: typed K / Q / V projections of every
node, relation_att applied per (relation, destination) row, an inner product per edge, the edge softmax and the message
product fused with the aggregation -- with [E,H] / [E,H,dk] tensors between them.  Here, for graphs whose relations are
canonical edge types (one source node type each -- what the reference's HGT requires too, HGT/models.py:31-52):

  w_kv[r] = [ K_st(r) . att'[r] . pri[r] / sqrt(dk)  |  V_st(r) . msg[r] ]     [in, 2*H*dk], built per step from the parameters
                                                       with torch ops (autograd carries the gradient back to k_linears,
                                                       relation_att, relation_pri, v_linears, relation_msg) -- the
                                                       reference's --multiply_among_weights_first_flag idea
                                                       (HGT/models.py:124-151) applied to the source side
  kv_c    = h[src of row] . w_kv[r of row]            one segment GEMM over the S_row distinct (relation, source) rows
  q       = typed projection of the nodes
  new_h   = het_hgt_aggregate_compact(kv_c, q)        softmax + aggregation, csrc/hgt_compact.hip

att' = relation_att for --hgt_fused_attn_score_flag (s = <k . att, q>), its transpose otherwise (s = <q . att, k>).
Same function of the parameters as the composition (associativity of the matrix products): tests/ check both against
oracle/layers.py::hgt_layer.  HET_HGT_FUSED=0 keeps the composition.
"""
import os

import torch as th

from ..plan import consistent as _consistent_plan

from .. import kernels as _k
from ..kernels import K
from .rgat_fused_layer import OVERLAP, _edge_rows, _has_single_sided_lists, _side_stream
from .rgnn_layers_and_funcs import rgnn_relational_matmul_no_scatter_gather_list as B_matmul_no_scatter_gather

FUSED = os.environ.get("HET_HGT_FUSED", "1") != "0"
# the layer's input gradient in one node-major pass per node type (csrc/node_sum.hip) instead of one read-modify-write launch
# per relation on the 2X-wide source rows + the destination-side projection's own pass (HET_HGT_NODE_DX=0: the round-3 form)
REFORK = os.environ.get("HET_HGT_REFORK", "1") != "0"  # diagnostic: 0 = the round-4 hazard of exp/hgt_cold_lag.py back in
NODE_DX = os.environ.get("HET_HGT_NODE_DX", "1") != "0"


def _node_dx_plan(G, ss, offs, K_in, X, dst):
    """Per graph, cached on it: what the node-major input gradient needs -- [R,N] int32 row maps of the (relation, source) list,
    the rank of every node in the list of destinations with in-edges (compact form), the nodes sorted by (node type, which
    sources they have rows in) so that 32-node tiles are homogeneous, and per node type its position range + relations.
    None when the graph or the shapes are outside the pass (sampled blocks with type runs, too many relations per type)."""
    key = ("hgt_node_dx", K_in, X, dst is not None)
    hit = G._plans.get(key)
    if hit is not None:
        return hit or None
    plan = False
    orig = G.graph_data["original"]
    if orig.get("node_segment_types") is None and X in (32, 64) and K_in in (32, 64):
        st, _ = G.get_rel_node_types()
        st_l, offs_l = st.tolist(), offs.tolist()
        T, R, N = len(offs_l) - 1, len(st_l), offs_l[-1]
        rels_of = [[r for r in range(R) if st_l[r] == t] for t in range(T)]
        if all(_k.node_rows_matmul_sum_ok(1 + 2 * len(rs), X, K_in) for rs in rels_of):
            row_map = _k.node_row_map(ss["rel_ptrs_row"], ss["node_indices_row"], N)
            dev = row_map.device
            dst_rank = None
            has_dst = [True] * T
            if dst is not None:
                dst_nodes, _, run_ptrs = dst
                dst_rank = th.full((N,), -1, dtype=th.int32, device=dev)
                dst_rank[dst_nodes] = th.arange(dst_nodes.numel(), dtype=th.int32, device=dev)
                rl = run_ptrs.tolist()
                has_dst = [rl[t + 1] > rl[t] for t in range(T)]
            w = (1 << th.arange(R, device=dev, dtype=th.int64)).view(R, 1)
            mask = ((row_map >= 0).to(th.int64) * w).sum(0)
            if dst_rank is not None:
                mask = mask | ((dst_rank >= 0).to(th.int64) << R)
            typ = th.searchsorted(offs[1:].contiguous().to(dev), th.arange(N, device=dev), right=True)
            order = th.argsort(mask | (typ << (R + 1)), stable=True).to(th.int32).contiguous()
            plan = dict(row_map=row_map, dst_rank=dst_rank, order=order, offs=offs_l, rels_of=rels_of, has_dst=has_dst)
    G._plans[key] = plan
    return plan or None


def hgt_fused_ok(G, h, num_heads, d_k):
    """Graphs (full, or sampled blocks large enough to keep their groupings) with the unique (relation, node) lists (built on
    demand) and canonical relations, on the GPU, shapes the row kernels are built for."""
    if not (FUSED and _k._plan.is_enabled() and h.is_cuda and h.dim() == 2 and hasattr(G, "graph_data") and G.get_num_edges() > 0):
        return False
    if not (_has_single_sided_lists(G) or hasattr(G, "generate_separate_unique_node_indices_single_sided_for_each_etype")):
        return False
    if _head_groups(num_heads, _padded_head(d_k)) is None:
        return False
    try:
        G.get_rel_node_types()
    except ValueError:  # a relation mixes node types: no single K / V projection per relation to fold
        return False
    return True


def _head_groups(num_heads: int, d_pad: int):
    """Number of head groups the attention runs in: 1 when the row kernels take all heads' rows at once (up to 128 floats per
    row), else the smallest split into equal groups they do take -- heads are independent up to the output projection, so 8
    heads of 32 floats (out = 256) run as two passes over 4 heads each (ogbn-mag: 53 -> 27 ms per step; the op-by-op composition
    before).  None: no split fits (one head wider than 128 floats)."""
    for groups in (1, 2, 4, 8):
        if num_heads % groups == 0 and _k.hgt_compact_shape_ok(num_heads // groups, d_pad):
            return groups
    return None


def _padded_head(d_k: int) -> int:
    """Head width the row kernels run with: a power of two >= 8.  Narrower / odd heads (the 8 classes of the reference CLI's
    last layer split over 4 heads: d_k = 2) are zero-padded -- zero columns in the folded weights and in the typed projection
    of q add nothing to a score, and the padded components of the aggregated messages are dropped."""
    return max(8, 1 << max(0, int(d_k) - 1).bit_length())


def _pad_heads(w, num_heads, d_k, d_pad, parts=1):
    """[..., parts * H * d_k] -> [..., parts * H * d_pad] with zero columns after every head's d_k."""
    if d_pad == d_k:
        return w
    lead = w.shape[:-1]
    return th.nn.functional.pad(w.reshape(*lead, parts * num_heads, d_k), (0, d_pad - d_k)).reshape(*lead, parts * num_heads * d_pad)


FOLD_KERNEL = os.environ.get("HET_HGT_FOLD_KERNEL", "1") == "1"  # A/B: the torch composition below instead of the two HIP launches


class _FoldSourceWeights(th.autograd.Function):
    """fold_source_weights as one HIP launch and its backward as two (csrc/hgt_fold.hip): written with torch ops the folding is ~14
    launches of a few microseconds each per step and ~24 more in autograd's backward -- 0.3 ms of HGT's 6.4 ms step on ogbn-mag."""

    @staticmethod
    def forward(ctx, k_lin, v_lin, rel_att, rel_msg, rel_pri, src_type, fused_attn):
        args = tuple(t.contiguous() for t in (k_lin, v_lin, rel_att, rel_msg, rel_pri))
        ctx.save_for_backward(*args, src_type)
        ctx.transpose = not fused_attn
        return _k.hgt_fold_source_weights(*args, src_type, ctx.transpose)

    @staticmethod
    def backward(ctx, grad_w):
        *args, src_type = ctx.saved_tensors
        return (*_k.hgt_fold_source_weights_backward(grad_w.contiguous(), *args, src_type, ctx.transpose), None, None)


def fold_source_weights(k_lin, v_lin, rel_att, rel_msg, rel_pri, src_type, num_heads, fused_attn):
    """w_kv [R,1,in,2*H*dk] (module docstring).  k_lin / v_lin [T,1,in,H*dk]; rel_att / rel_msg [R,H,dk,dk]; rel_pri [R,H]."""
    if (FOLD_KERNEL and k_lin.is_cuda and k_lin.dtype == th.float32 and k_lin.dim() == 4 and k_lin.shape[1] == 1
            and src_type.is_cuda and src_type.dtype == th.int64 and _k._lib.has("het_hgt_fold_source_weights")):
        return _FoldSourceWeights.apply(k_lin, v_lin, rel_att, rel_msg, rel_pri, src_type.contiguous(), fused_attn)
    R, H, dk, _ = rel_att.shape
    K_in = k_lin.shape[2]
    heads = lambda w: w.index_select(0, src_type).view(R, K_in, H, dk).permute(0, 2, 1, 3)  # [R,H,in,dk]
    att = rel_att if fused_attn else rel_att.transpose(2, 3)
    mu = (rel_pri / (dk ** 0.5)).view(R, H, 1, 1)
    wk = th.matmul(heads(k_lin), att * mu)   # [R,H,in,dk]
    wm = th.matmul(heads(v_lin), rel_msg)
    flat = lambda w: w.permute(0, 2, 1, 3).reshape(R, K_in, H * dk)
    return th.cat([flat(wk), flat(wm)], dim=2).unsqueeze(1).contiguous()


# Destinations WITHOUT in-edges receive nothing: their rows of new_h are zero, and so are their rows of the layer output
# (the typed output projection has no bias).  When they are many (on the ogbn-mag-like graph 58 % of the nodes), q, lsum,
# new_h and the output projection live on the S_dst destinations that do have in-edges: the row kernels are agnostic (their
# "node" is then the rank of the destination in the sorted list of those), the two typed projections run on S_dst rows.
COMPACT_DST_BELOW = float(os.environ.get("HET_HGT_COMPACT_DST", "0.8"))  # use the lists when S_dst < this fraction of N


@_consistent_plan
class HgtAttentionFunction(th.autograd.Function):
    @staticmethod
    def forward(ctx, G, num_heads, offs, h, w_kv, q_w, dst):
        """h [N,in]; w_kv [R,1,in,2X]; q_w [T,1,in,X] typed projection of the destination side (one weight per run of offs).
        dst = None: rows of the result are nodes.  dst = (dst_nodes, rank_of_edge, run_ptrs) (kernels.destination_lists):
        rows are the destinations with in-edges, in that order."""
        h, w_kv, q_w = h.contiguous(), w_kv.contiguous(), q_w.contiguous()
        N, K_in = h.shape
        X = q_w.shape[3]
        H = num_heads
        s = G.get_separate_coo_original()
        ss = G.get_separate_unique_node_indices_single_sided()
        rp_row, rows_node = ss["rel_ptrs_row"], ss["node_indices_row"]
        S_row = rows_node.numel()
        new = lambda *shape: th.empty(shape, dtype=h.dtype, device=h.device)
        # the destination-side projection beside the source-row one (rgat_fused_layer._side_stream: independent launches, the
        # tensors are allocated and freed under the main stream)
        main, side = th.cuda.current_stream(h.device), (_side_stream(h.device) if OVERLAP and h.is_cuda else None)
        if dst is None:
            ND, keys = N, s["col_indices"]
            q = new(ND, X)
        else:
            dst_nodes, keys, run_ptrs = dst
            ND = dst_nodes.numel()
            q = new(ND, 1, X)
        kv_c = new(S_row, 1, 2 * X)
        if side is not None:
            side.wait_stream(main)
        with th.cuda.stream(side if side is not None else main):
            if dst is None:
                K.rgnn_relational_matmul_no_scatter_gather_list(offs, q_w, h, q)
            else:
                K.rgnn_relational_matmul({"unique_srcs_and_dests_rel_ptrs": run_ptrs, "unique_srcs_and_dests_node_indices": dst_nodes},
                                         1, q_w, h, q, True)
        q = q.view(ND, X)
        K.rgnn_relational_matmul({"unique_srcs_and_dests_rel_ptrs": rp_row, "unique_srcs_and_dests_node_indices": rows_node},
                                 1, w_kv, h, kv_c, True)
        if side is not None:
            main.wait_stream(side)
        srow, _ = _edge_rows(G, ss, True, s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"])
        grp = _k.hgt_compact_groupings(keys, srow, ND, S_row)
        lsum, out = new(ND, H), new(ND, X)
        _k.hgt_aggregate_compact(grp, kv_c, q, lsum, out)
        ctx.G, ctx.H, ctx.grp, ctx.compact_dst = G, H, grp, dst is not None
        ctx.save_for_backward(h, w_kv, q_w, offs, q, kv_c, lsum, out, *(() if dst is None else (dst[0], dst[2])))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, w_kv, q_w, offs, q, kv_c, lsum, out, *lists = ctx.saved_tensors
        G, H = ctx.G, ctx.H
        N, K_in = h.shape
        X = q_w.shape[3]
        ss = G.get_separate_unique_node_indices_single_sided()
        rp_row, rows_node = ss["rel_ptrs_row"], ss["node_indices_row"]
        grad_out = grad_out.contiguous()
        g_kv, g_q = th.empty_like(kv_c), th.empty_like(q)
        _k.hgt_backward_compact(ctx.grp, kv_c, q, lsum, out, grad_out, g_kv, g_q)
        # one input-gradient buffer for both consumers of h
        qwt = q_w.transpose(2, 3).contiguous()
        wt = w_kv.transpose(2, 3).contiguous()
        split_kv = _k.rows_matmul_backward_split_ok(1, K_in, 2 * X)
        # the weight gradients (HBM-bound streams of rows) on the side stream beside the input-gradient chain
        main, side = th.cuda.current_stream(h.device), (_side_stream(h.device) if OVERLAP and h.is_cuda else None)
        grad_wkv = th.empty_like(w_kv) if split_kv else None
        if side is not None and split_kv:
            side.wait_stream(main)
            with th.cuda.stream(side):
                _k.rows_matmul_backward_dw(rp_row, rows_node, h, g_kv.view(-1, 2 * X), grad_wkv, accumulate=False)
        nplan = _node_dx_plan(G, ss, offs, K_in, X, (lists[0], None, lists[1]) if ctx.compact_dst else None) if (NODE_DX and split_kv) else None
        if nplan is not None:
            # every consumer of h adds its term in ONE pass over the nodes: the destination-side projection's gradient rows
            # (g_q . Q_t^T) and, per relation the node is a source of, the two halves of its [k' | m] gradient row
            grad_h, grad_qw = th.empty_like(h), th.empty_like(q_w)  # (allocated under the main stream)
            g_kv2, g_q2 = g_kv.view(-1, 2 * X), g_q.view(-1, X)
            wt2 = wt.view(-1, 2 * X, K_in)
            if side is not None and REFORK:
                # grad_qw (and, on a graph seen for the first time, the plan's temporaries) were allocated AFTER the fork above: the
                # allocator may have handed out blocks whose previous main-stream use was enqueued after that fork (the kernels
                # that build the plan), and the side stream would write grad_qw while they still run -- seen once in ~40 cold
                # runs as rows of grad_h missing (the node order came out corrupted).  Fork again: free when nothing is pending.
                side.wait_stream(main)
            with th.cuda.stream(side if side is not None else main):  # the other two weight gradients beside the node pass
                if side is None:
                    _k.rows_matmul_backward_dw(rp_row, rows_node, h, g_kv2, grad_wkv, accumulate=False)
                if ctx.compact_dst:
                    _k.rows_matmul_backward_dw(lists[1], lists[0], h, g_q2, grad_qw, accumulate=False)
                else:
                    _k.rows_matmul_backward_dw(offs, None, h, g_q2, grad_qw, accumulate=False)
            for t, rels in enumerate(nplan["rels_of"]):
                a, b = nplan["offs"][t], nplan["offs"][t + 1]
                srcs = []
                if nplan["has_dst"][t]:
                    srcs.append((g_q2, 0, nplan["dst_rank"], qwt[t, 0]))
                for r in rels:
                    srcs.append((g_kv2, 0, nplan["row_map"][r], wt2[r, :X]))
                    srcs.append((g_kv2, X, nplan["row_map"][r], wt2[r, X:]))
                if not srcs:
                    grad_h[a:b].zero_()  # (a node type that neither sends nor receives: its rows of the gradient are zero)
                elif b > a:
                    _k.node_rows_matmul_sum(a, b, srcs, grad_h, nplan["order"])
            if side is not None:
                main.wait_stream(side)
            return None, None, None, grad_h, grad_wkv, grad_qw, None
        if not ctx.compact_dst:  # the typed projection writes every row with plain stores, the source-row GEMM adds to it
            grad_h, grad_qw = th.empty_like(h), th.empty_like(q_w)
            _k.matmul_no_scatter_gather_backward(offs, qwt, h, g_q, grad_h, grad_qw, accumulate=False)
        else:
            dst_nodes, run_ptrs = lists
            grad_h = th.zeros_like(h)
            if _k.rows_matmul_backward_split_ok(1, K_in, X):
                grad_qw = th.empty_like(q_w)
                if side is not None:
                    side.wait_stream(main)  # (grad_qw was allocated after the first fork: see the node-major branch)
                    with th.cuda.stream(side):
                        _k.rows_matmul_backward_dw(run_ptrs, dst_nodes, h, g_q, grad_qw, accumulate=False)
                _k.rows_matmul_backward_dx(run_ptrs, dst_nodes, qwt, g_q, grad_h, atomic=False)  # distinct nodes, first writer: "="
                if side is None:
                    _k.rows_matmul_backward_dw(run_ptrs, dst_nodes, h, g_q, grad_qw, accumulate=False)
            else:
                grad_qw = th.zeros_like(q_w)
                _k.matmul_backward({"unique_srcs_and_dests_rel_ptrs": run_ptrs, "unique_srcs_and_dests_node_indices": dst_nodes}, 1,
                                   qwt, h, g_q.view(-1, 1, X), grad_h, grad_qw, True, accumulate=True, distinct_rows=True)
        if split_kv:
            _k.rows_matmul_backward_dx(rp_row, rows_node, wt, g_kv.view(-1, 2 * X), grad_h, atomic=2)  # rows of a relation: distinct nodes
            if side is None:
                _k.rows_matmul_backward_dw(rp_row, rows_node, h, g_kv.view(-1, 2 * X), grad_wkv, accumulate=False)
        else:
            grad_wkv = th.zeros_like(w_kv)
            _k.matmul_backward({"unique_srcs_and_dests_rel_ptrs": rp_row, "unique_srcs_and_dests_node_indices": rows_node}, 1, wt, h,
                               g_kv, grad_h, grad_wkv, True, accumulate=True, distinct_rows=True)
        if side is not None:
            main.wait_stream(side)  # (a wait for a stream without new work is free)
        return None, None, None, grad_h, grad_wkv, grad_qw, None


@_consistent_plan
class RowsLinearScatter(th.autograd.Function):
    """out[rows[i], :] = x_c[i, :] . w[t(i)] for the listed rows, zero elsewhere: a typed projection of compact rows written
    straight into the dense result (and its gradient read straight from the dense gradient) -- no index_copy / index_select
    passes.  rows are distinct and sorted, ``run_ptrs`` [T+1] splits them into the runs of ``w`` [T,1,K,X]."""

    @staticmethod
    def forward(ctx, run_ptrs, rows, num_out_rows, x_c, w):
        x_c, w = x_c.contiguous(), w.contiguous()
        out = th.zeros((num_out_rows, w.shape[3]), dtype=w.dtype, device=w.device)
        # (the "transposed weight" slot of the dX entry point takes [T,1,D_in,K_out]: exactly w)
        _k.rows_matmul_backward_dx(run_ptrs, rows, w, x_c, out, atomic=False)
        ctx.save_for_backward(run_ptrs, rows, x_c, w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        run_ptrs, rows, x_c, w = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        T, _, K_in, X_out = w.shape
        g_x = th.empty((x_c.shape[0], 1, K_in), dtype=w.dtype, device=w.device)
        K.rgnn_relational_matmul({"unique_srcs_and_dests_rel_ptrs": run_ptrs, "unique_srcs_and_dests_node_indices": rows}, 1,
                                 w.transpose(2, 3).contiguous(), grad_out, g_x, True)  # g_x[i] = grad_out[rows[i]] . w[t]^T
        g_wt = th.empty((T, 1, X_out, K_in), dtype=w.dtype, device=w.device)
        _k.rows_matmul_backward_dw(run_ptrs, rows, grad_out, x_c, g_wt, accumulate=False)  # SUM grad_out[rows[i]]^T (x) x_c[i]
        return None, None, None, g_x.view_as(x_c), g_wt.transpose(2, 3)


def hgt_layer_fused(G, h, offs, q_w, a_w, k_lin, v_lin, rel_att, rel_msg, rel_pri, num_heads, fused_attn):
    """The HGT layer [N, out]: attention + aggregation as one node, then the typed output projection ``a_w`` [T,1,X,out] (one
    weight per run of offs; the caller folds sigmoid(skip) in) -- on the destinations with in-edges only when those are a
    minority of the nodes (the other rows of the output are zero by construction)."""
    if not _has_single_sided_lists(G):
        G.generate_separate_unique_node_indices_single_sided_for_each_etype()
    st, _ = G.get_rel_node_types()
    w_kv = fold_source_weights(k_lin, v_lin, rel_att, rel_msg, rel_pri, st, num_heads, fused_attn)
    N = h.shape[0]
    d_k = q_w.shape[3] // num_heads
    d_pad = _padded_head(d_k)
    w_kv, q_w = _pad_heads(w_kv, num_heads, d_k, d_pad, parts=2), _pad_heads(q_w, num_heads, d_k, d_pad)
    unpad = (lambda t: t) if d_pad == d_k else (lambda t: t.view(t.shape[0], num_heads, d_pad)[..., :d_k].reshape(t.shape[0], -1))
    col = G.get_separate_coo_original()["col_indices"]
    dst = _k.destination_lists(col, offs)
    groups = _head_groups(num_heads, d_pad)

    def attention(dst_lists):
        if groups == 1:
            return HgtAttentionFunction.apply(G, num_heads, offs, h, w_kv, q_w, dst_lists)
        # rows wider than the row kernels take: the heads in `groups` passes (columns of a head group in [k' | m] and in q)
        X, Xg, Hg = num_heads * d_pad, num_heads * d_pad // groups, num_heads // groups
        parts = []
        for i in range(groups):
            w_g = th.cat([w_kv[..., i * Xg:(i + 1) * Xg], w_kv[..., X + i * Xg:X + (i + 1) * Xg]], dim=-1).contiguous()
            parts.append(HgtAttentionFunction.apply(G, Hg, offs, h, w_g, q_w[..., i * Xg:(i + 1) * Xg].contiguous(), dst_lists))
        return th.cat(parts, dim=1)

    if dst[0].numel() >= COMPACT_DST_BELOW * N:
        new_h = unpad(attention(None))
        return B_matmul_no_scatter_gather(offs, a_w, new_h)
    new_h_c = unpad(attention(dst))
    if _k.rows_matmul_backward_split_ok(1, a_w.shape[3], a_w.shape[2]):
        return RowsLinearScatter.apply(dst[2], dst[0], N, new_h_c, a_w)
    out_c = B_matmul_no_scatter_gather(dst[2], a_w, new_h_c)  # rows of a type are a contiguous piece of the sorted list
    return th.zeros((N, out_c.shape[1]), dtype=out_c.dtype, device=out_c.device).index_copy(0, dst[0], out_c)
