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
from .rgat_fused_layer import _edge_rows, _has_single_sided_lists

FUSED = os.environ.get("HET_HGT_FUSED", "1") != "0"


def hgt_fused_ok(G, h, num_heads, d_k):
    """Graphs (full, or sampled blocks large enough to keep their groupings) with the unique (relation, node) lists (built on
    demand) and canonical relations, on the GPU, shapes the row kernels are built for."""
    if not (FUSED and _k._plan.is_enabled() and h.is_cuda and h.dim() == 2 and hasattr(G, "graph_data") and G.get_num_edges() > 0):
        return False
    if not (_has_single_sided_lists(G) or hasattr(G, "generate_separate_unique_node_indices_single_sided_for_each_etype")):
        return False
    if not _k.hgt_compact_shape_ok(num_heads, d_k):
        return False
    try:
        G.get_rel_node_types()
    except ValueError:  # a relation mixes node types: no single K / V projection per relation to fold
        return False
    return True


def fold_source_weights(k_lin, v_lin, rel_att, rel_msg, rel_pri, src_type, num_heads, fused_attn):
    """w_kv [R,1,in,2*H*dk] (module docstring).  k_lin / v_lin [T,1,in,H*dk]; rel_att / rel_msg [R,H,dk,dk]; rel_pri [R,H]."""
    R, H, dk, _ = rel_att.shape
    K_in = k_lin.shape[2]
    heads = lambda w: w.index_select(0, src_type).view(R, K_in, H, dk).permute(0, 2, 1, 3)  # [R,H,in,dk]
    att = rel_att if fused_attn else rel_att.transpose(2, 3)
    mu = (rel_pri / (dk ** 0.5)).view(R, H, 1, 1)
    wk = th.matmul(heads(k_lin), att * mu)   # [R,H,in,dk]
    wm = th.matmul(heads(v_lin), rel_msg)
    flat = lambda w: w.permute(0, 2, 1, 3).reshape(R, K_in, H * dk)
    return th.cat([flat(wk), flat(wm)], dim=2).unsqueeze(1).contiguous()


@_consistent_plan
class HgtAttentionFunction(th.autograd.Function):
    @staticmethod
    def forward(ctx, G, num_heads, offs, h, w_kv, q_w):
        """h [N,in]; w_kv [R,1,in,2X]; q_w [T,1,in,X] typed projection of the destination side (one weight per run of offs)."""
        h, w_kv, q_w = h.contiguous(), w_kv.contiguous(), q_w.contiguous()
        N, K_in = h.shape
        X = q_w.shape[3]
        H, D = num_heads, X // num_heads
        s = G.get_separate_coo_original()
        ss = G.get_separate_unique_node_indices_single_sided()
        rp_row, rows_node = ss["rel_ptrs_row"], ss["node_indices_row"]
        S_row = rows_node.numel()
        new = lambda *shape: th.empty(shape, dtype=h.dtype, device=h.device)
        q = new(N, X)
        K.rgnn_relational_matmul_no_scatter_gather_list(offs, q_w, h, q)
        kv_c = new(S_row, 1, 2 * X)
        K.rgnn_relational_matmul({"unique_srcs_and_dests_rel_ptrs": rp_row, "unique_srcs_and_dests_node_indices": rows_node},
                                 1, w_kv, h, kv_c, True)
        srow, _ = _edge_rows(G, ss, True, s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"])
        grp = _k.hgt_compact_groupings(s["col_indices"], srow, N, S_row)
        lsum, out = new(N, H), new(N, X)
        _k.hgt_aggregate_compact(grp, kv_c, q, lsum, out)
        ctx.G, ctx.H, ctx.grp = G, H, grp
        ctx.save_for_backward(h, w_kv, q_w, offs, q, kv_c, lsum, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, w_kv, q_w, offs, q, kv_c, lsum, out = ctx.saved_tensors
        G, H = ctx.G, ctx.H
        N, K_in = h.shape
        X = q_w.shape[3]
        ss = G.get_separate_unique_node_indices_single_sided()
        rp_row, rows_node = ss["rel_ptrs_row"], ss["node_indices_row"]
        grad_out = grad_out.contiguous()
        g_kv, g_q = th.empty_like(kv_c), th.empty_like(q)
        _k.hgt_backward_compact(ctx.grp, kv_c, q, lsum, out, grad_out, g_kv, g_q)
        # one input-gradient buffer: the typed projection writes every row with plain stores, the source-row GEMM adds to it
        grad_h, grad_qw = th.empty_like(h), th.empty_like(q_w)
        _k.matmul_no_scatter_gather_backward(offs, q_w.transpose(2, 3).contiguous(), h, g_q, grad_h, grad_qw, accumulate=False)
        wt = w_kv.transpose(2, 3).contiguous()
        if _k.rows_matmul_backward_split_ok(1, K_in, 2 * X):
            grad_wkv = th.empty_like(w_kv)
            _k.rows_matmul_backward_dx(rp_row, rows_node, wt, g_kv.view(-1, 2 * X), grad_h, atomic=2)  # rows of a relation: distinct nodes
            _k.rows_matmul_backward_dw(rp_row, rows_node, h, g_kv.view(-1, 2 * X), grad_wkv, accumulate=False)
        else:
            grad_wkv = th.zeros_like(w_kv)
            _k.matmul_backward({"unique_srcs_and_dests_rel_ptrs": rp_row, "unique_srcs_and_dests_node_indices": rows_node}, 1, wt, h,
                               g_kv, grad_h, grad_wkv, True, accumulate=True)
        return None, None, None, grad_h, grad_wkv, grad_qw


def hgt_attention_fused(G, h, offs, q_w, k_lin, v_lin, rel_att, rel_msg, rel_pri, num_heads, fused_attn):
    """new_h [N, H*dk] of the HGT layer (before the typed output projection)."""
    if not _has_single_sided_lists(G):
        G.generate_separate_unique_node_indices_single_sided_for_each_etype()
    st, _ = G.get_rel_node_types()
    w_kv = fold_source_weights(k_lin, v_lin, rel_att, rel_msg, rel_pri, st, num_heads, fused_attn)
    return HgtAttentionFunction.apply(G, num_heads, offs, h, w_kv, q_w)
