"""Fused GAT (hetero edge-softmax + aggregation) behind torch.autograd; mirrors
/root/reference/hrt/python/backend/rgat_layers_and_funcs.py (classes :8-648,
wrappers :651-890)."""
import torch as th

from ..plan import consistent as _consistent_plan

from .. import kernels as _k
from ..kernels import K

__all__ = [
    "RelationalFusedGatCSR", "RelationalFusedGatSeparateCOO",
    "RelationalFusedGatCompactAsOfNodeSeparateCOODualUniqueNodeList",
    "RelationalFusedGatCompactAsOfNodeSeparateCOODualUniqueNodeListDirectIndexing",
    "relational_fused_gat_csr", "relational_fused_gat_separate_coo",
    "relational_fused_gat_compact_as_of_node_separate_coo_dual_unique_node_list",
    "relational_fused_gat_compact_as_of_node_separate_coo_single_sided",
    "relational_fused_gat_separate_coo_with_attn_l", "relational_fused_gat_separate_coo_with_attn_l_ok",
    "relational_fused_gat_compact_with_attn_l", "relational_fused_gat_compact_with_attn_l_ok",
]


@_consistent_plan
class RelationalFusedGatCSR(th.autograd.Function):
    # reference: rgat_layers_and_funcs.py:8-115
    @staticmethod
    def forward(ctx, incsr_row_ptr, incsr_col_indices, incsr_eids, incsr_reltypes, outcsr_row_ptr, outcsr_col_indices,
                outcsr_eids, outcsr_reltypes, unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices,
                feat_src, el, er, s, exp, ret, slope):
        ctx.save_for_backward(outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes,
                              unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, feat_src, el, er, s,
                              exp, ret)
        ctx.slope = slope
        K.relational_fused_gat_csr(incsr_row_ptr, incsr_col_indices, incsr_eids, incsr_reltypes,
                                   unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, feat_src, el, er,
                                   s, exp, ret, slope, False)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        (outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes, u_rel_ptrs, u_nodes, feat_src, el, er, s,
         exp, ret) = ctx.saved_tensors
        grad_el = th.zeros_like(el, memory_format=th.contiguous_format)
        grad_er = th.zeros_like(er, memory_format=th.contiguous_format)
        grad_feat_src = th.zeros_like(feat_src, memory_format=th.contiguous_format)
        K.backward_relational_fused_gat_csr(outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes, u_rel_ptrs,
                                            u_nodes, feat_src, el, er, s, exp, ret, gradout.contiguous(), grad_feat_src,
                                            grad_el, grad_er, ctx.slope, False)
        return (None,) * 10 + (grad_feat_src, grad_el, grad_er, None, None, None, None)


@_consistent_plan
class _FusedGatSeparateCOO(th.autograd.Function):
    """Shared body of the separate-COO classes: ``kind`` and the two dicts select the row maps."""

    @staticmethod
    def forward(ctx, eids, rel_ptrs, row, col, kind, fwd_dict, bwd_dict, feat_src, el, er, s, exp, ret, slope):
        H = el.shape[1]
        D = feat_src.numel() // max(1, feat_src.shape[0] * H)
        # private extra output: exp in destination-grouped order, streamed by the backward (kind 0 only)
        exp_sorted = th.empty_like(exp) if (kind == 0 and slope >= 0 and _k.gat_grouped_shape_ok(H, D)) else None
        used = _k.fused_gat_forward(eids, rel_ptrs, row, col, kind, fwd_dict, feat_src, el, er, s, exp, ret, slope,
                                    exp_sorted)
        ctx.save_for_backward(eids, rel_ptrs, row, col, feat_src, el, er, s, exp, ret, exp_sorted if used else None)
        ctx.kind, ctx.bwd_dict, ctx.slope = kind, bwd_dict, slope
        ctx.alias_ok = _k.gat_grouped_shape_ok(H, D)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        eids, rel_ptrs, row, col, feat_src, el, er, s, exp, ret, exp_sorted = ctx.saved_tensors
        if ctx.kind == 0:  # every gradient row is written exactly once by the op: no zero-fill needed
            grad_el, grad_feat_src = th.empty_like(el), th.empty_like(feat_src)
            # grad_er == grad_el element for element (both rows belong to the same edge): one buffer
            grad_er = grad_el if ctx.alias_ok else th.empty_like(er)
        else:
            grad_el = th.zeros_like(el, memory_format=th.contiguous_format)
            grad_er = th.zeros_like(er, memory_format=th.contiguous_format)
            grad_feat_src = th.zeros_like(feat_src, memory_format=th.contiguous_format)
        _k.fused_gat_backward(eids, rel_ptrs, row, col, ctx.kind, ctx.bwd_dict, feat_src, el, er, s, exp, ret,
                              gradout.contiguous(), grad_feat_src, grad_el, grad_er, ctx.slope, exp_sorted)
        return None, None, None, None, None, None, None, grad_feat_src, grad_el, grad_er, None, None, None, None


@_consistent_plan
class _FusedGatSeparateCOOWithAttnL(th.autograd.Function):
    """el = <feat_src, attn_l[r]> (the D_out = 1 segment GEMM of RGAT/models.py:288-296) and the fused GAT op under
    ONE autograd node, so that the two gradients that meet in feat_src are written by one store: the GAT backward
    kernel adds grad_el * attn_l[r] while it writes a * gradout[dst] (include/het_amd.h: fold_attn_l).  Same
    values as RgnnRelationalMatmul + RelationalFusedGatSeparateCOO; saves the separate gradient tensor, its
    read-modify-write pass and the autograd accumulation over [E,H,D]."""

    @staticmethod
    def forward(ctx, eids, rel_ptrs, row, col, feat_src, attn_l, er, s, exp, ret, slope, el):
        E, H = eids.numel(), attn_l.shape[1]
        if el is None:  # else: el = <feat_src, attn_l[r]> as produced by rgnn_relational_matmul_with_attn_dot(folded=True)
            by_eid = {"separate_coo_rel_ptrs": rel_ptrs, "separate_coo_node_indices": eids, "separate_coo_eids": eids}
            el = th.empty((E, H), dtype=feat_src.dtype, device=feat_src.device)
            K.rgnn_relational_matmul(by_eid, 0, attn_l.unsqueeze(-1), feat_src, el, False)
        exp_sorted = th.empty_like(exp)
        used = _k.fused_gat_forward(eids, rel_ptrs, row, col, 0, {}, feat_src, el, er, s, exp, ret, slope, exp_sorted)
        if not used:
            raise RuntimeError("relational_fused_gat_separate_coo_with_attn_l needs the destination-grouped kernels")
        ctx.save_for_backward(eids, rel_ptrs, row, col, feat_src, attn_l, el, er, s, exp, ret, exp_sorted)
        ctx.slope = slope
        return ret

    @staticmethod
    def backward(ctx, gradout):
        eids, rel_ptrs, row, col, feat_src, attn_l, el, er, s, exp, ret, exp_sorted = ctx.saved_tensors
        grad_el, grad_feat_src = th.empty_like(el), th.empty_like(feat_src)
        if attn_l.shape[0] <= 8:  # the weight gradient of the folded product from the same pass (R <= 8 in registers)
            grad_attn_l = th.zeros_like(attn_l)
            _k.fused_gat_backward(eids, rel_ptrs, row, col, 0, {}, feat_src, el, er, s, exp, ret, gradout.contiguous(),
                                  grad_feat_src, grad_el, grad_el, ctx.slope, exp_sorted, fold_attn_l=attn_l,
                                  grad_fold_attn_l=grad_attn_l)
        else:
            _k.fused_gat_backward(eids, rel_ptrs, row, col, 0, {}, feat_src, el, er, s, exp, ret, gradout.contiguous(),
                                  grad_feat_src, grad_el, grad_el, ctx.slope, exp_sorted, fold_attn_l=attn_l)
            by_eid = {"separate_coo_rel_ptrs": rel_ptrs, "separate_coo_node_indices": eids, "separate_coo_eids": eids}
            grad_attn_l = th.empty_like(attn_l)
            _k.matmul_backward(by_eid, 0, attn_l.unsqueeze(2), feat_src, grad_el, None, grad_attn_l.unsqueeze(-1), False,
                               accumulate=False)
        return None, None, None, None, grad_feat_src, grad_attn_l, grad_el, None, None, None, None, None


@_consistent_plan
class _FusedGatCompactWithAttnL(th.autograd.Function):
    """The compact-as-of-node counterpart of _FusedGatSeparateCOOWithAttnL: el_compact = <feat_compact, attn_l[r]> over
    the (relation, source) rows and the fused GAT op (kinds 3 / 4) under one autograd node; the gradient through el is
    added to the gradient of feat_compact by the store of the GAT backward (fold_attn_l on compact rows)."""

    @staticmethod
    def forward(ctx, eids, rel_ptrs, row, col, kind, fwd_dict, bwd_dict, row_rel_ptrs, feat_compact, attn_l, er, s, exp,
                ret, slope):
        el = th.empty((feat_compact.shape[0], attn_l.shape[1]), dtype=feat_compact.dtype, device=feat_compact.device)
        K.rgnn_relational_matmul_no_scatter_gather_list(row_rel_ptrs, attn_l.unsqueeze(-1), feat_compact, el)
        _k.fused_gat_forward(eids, rel_ptrs, row, col, kind, fwd_dict, feat_compact, el, er, s, exp, ret, slope, None)
        ctx.save_for_backward(eids, rel_ptrs, row, col, row_rel_ptrs, feat_compact, attn_l, el, er, s, exp, ret)
        ctx.kind, ctx.bwd_dict, ctx.slope = kind, bwd_dict, slope
        return ret

    @staticmethod
    def backward(ctx, gradout):
        eids, rel_ptrs, row, col, row_rel_ptrs, feat, attn_l, el, er, s, exp, ret = ctx.saved_tensors
        grad_el, grad_er = th.zeros_like(el), th.zeros_like(er)
        grad_feat = th.zeros_like(feat)
        _k.fused_gat_backward(eids, rel_ptrs, row, col, ctx.kind, ctx.bwd_dict, feat, el, er, s, exp, ret,
                              gradout.contiguous(), grad_feat, grad_el, grad_er, ctx.slope, None, fold_attn_l=attn_l,
                              fold_row_rel_ptrs=row_rel_ptrs)
        grad_attn_l = th.empty_like(attn_l)
        _k.matmul_no_scatter_gather_backward(row_rel_ptrs, attn_l.unsqueeze(2), feat, grad_el, None,
                                             grad_attn_l.unsqueeze(-1), accumulate=False)
        return (None,) * 8 + (grad_feat, grad_attn_l, grad_er, None, None, None, None)


def relational_fused_gat_compact_with_attn_l_ok(g, feat_compact, attn_l, negative_slope):
    """Shapes / state for which the compact GAT backward runs on its groupings (the fold lives there)."""
    H, D = attn_l.shape[1], attn_l.shape[2]
    return (_k._plan.is_enabled() and negative_slope >= 0 and _k.gat_grouped_shape_ok(H, D) and feat_compact.is_cuda
            and g.get_num_edges() > 0)


def relational_fused_gat_compact_with_attn_l(g, feat_compact, attn_l, er_compact, negative_slope, compact_direct_indexing_flag):
    d = g.get_separate_coo_original()
    ss = g.get_separate_unique_node_indices_single_sided()
    N = g.get_num_nodes()
    exp = er_compact.new_empty([g.get_num_edges()] + list(er_compact.size()[1:]))
    s = er_compact.new_empty([N] + list(er_compact.size()[1:]))
    ret = th.empty([N] + list(feat_compact.size()[1:]), dtype=feat_compact.dtype, device=feat_compact.device)
    if compact_direct_indexing_flag:
        inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
        kind = 4
        dd = {"edata_idx_to_inverse_idx_row": inv["inverse_indices_row"], "edata_idx_to_inverse_idx_col": inv["inverse_indices_col"]}
        fwd = bwd = dd
    else:
        kind = 3
        fwd = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"], "unique_srcs_and_dests_rel_ptrs_col": ss["rel_ptrs_col"],
               "unique_srcs_and_dests_node_indices_row": ss["node_indices_row"],
               "unique_srcs_and_dests_node_indices_col": ss["node_indices_col"]}
        bwd = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"], "unique_srcs_and_dests_rel_col": ss["rel_ptrs_col"],
               "unique_srcs_and_dests_node_indices_row": ss["node_indices_row"],
               "unique_srcs_and_dests_node_indices_col": ss["node_indices_col"]}
    return _FusedGatCompactWithAttnL.apply(d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"], kind, fwd, bwd,
                                           ss["rel_ptrs_row"], feat_compact.contiguous(), attn_l.contiguous(),
                                           er_compact.contiguous(), s, exp, ret, negative_slope)


def relational_fused_gat_separate_coo_with_attn_l_ok(g, feat, attn_l, negative_slope):
    """Whether the fused node applies: kind 0 shapes of the destination-grouped kernels, slope >= 0."""
    H = attn_l.shape[1]
    D = attn_l.shape[2]
    return (_k._plan.is_enabled() and negative_slope >= 0 and _k.gat_grouped_shape_ok(H, D) and feat.is_cuda
            and g.get_num_edges() > 0)


def relational_fused_gat_separate_coo_with_attn_l(g, feat, attn_l, er, negative_slope, el=None):
    d = g.get_separate_coo_original()
    exp = er.new_empty(er.shape)
    s = er.new_empty([g.get_num_nodes()] + list(er.size()[1:]))
    ret = th.empty([g.get_num_nodes()] + list(feat.size()[1:]), dtype=feat.dtype, device=feat.device)
    return _FusedGatSeparateCOOWithAttnL.apply(d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"],
                                               feat.contiguous(), attn_l.contiguous(), er.contiguous(), s, exp, ret,
                                               negative_slope, None if el is None else el.detach())


class RelationalFusedGatSeparateCOO:
    # reference: rgat_layers_and_funcs.py:233-322
    @staticmethod
    def apply(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices, feat_src,
              el, er, s, exp, ret, slope):
        return _FusedGatSeparateCOO.apply(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices,
                                          separate_coo_col_indices, 0, {}, {}, feat_src, el, er, s, exp, ret, slope)


class RelationalFusedGatCompactAsOfNodeSeparateCOODualUniqueNodeList:
    # reference: rgat_layers_and_funcs.py:325-438 (kind 3; note the backward's dict key spelling)
    @staticmethod
    def apply(eids, rel_ptrs, row, col, rel_ptr_row, node_indices_row, rel_ptr_col, node_indices_col, feat_src, el, er,
              s, exp, ret, slope):
        fwd = {"unique_srcs_and_dests_rel_ptrs": rel_ptr_row, "unique_srcs_and_dests_rel_ptrs_col": rel_ptr_col,
               "unique_srcs_and_dests_node_indices_row": node_indices_row,
               "unique_srcs_and_dests_node_indices_col": node_indices_col}
        bwd = {"unique_srcs_and_dests_rel_ptrs": rel_ptr_row, "unique_srcs_and_dests_rel_col": rel_ptr_col,
               "unique_srcs_and_dests_node_indices_row": node_indices_row,
               "unique_srcs_and_dests_node_indices_col": node_indices_col}
        return _FusedGatSeparateCOO.apply(eids, rel_ptrs, row, col, 3, fwd, bwd, feat_src, el, er, s, exp, ret, slope)


class RelationalFusedGatCompactAsOfNodeSeparateCOODualUniqueNodeListDirectIndexing:
    # reference: rgat_layers_and_funcs.py:441-544 (kind 4)
    @staticmethod
    def apply(eids, rel_ptrs, row, col, inverse_indices_row, inverse_indices_col, feat_src, el, er, s, exp, ret, slope):
        d = {"edata_idx_to_inverse_idx_row": inverse_indices_row, "edata_idx_to_inverse_idx_col": inverse_indices_col}
        return _FusedGatSeparateCOO.apply(eids, rel_ptrs, row, col, 4, d, d, feat_src, el, er, s, exp, ret, slope)


def _alloc(g, feat, el):
    exp = el.new_empty([g.get_num_edges()] + list(el.size()[1:]))
    s = el.new_empty([g.get_num_nodes()] + list(el.size()[1:]))
    ret = th.empty([g.get_num_nodes()] + list(feat.size()[1:]), dtype=feat.dtype, device=feat.device)
    return exp, s, ret


def relational_fused_gat_csr(graph, feat_src, el, er, slope):
    # reference: rgat_layers_and_funcs.py:651-683
    i, o = graph.get_in_csr(), graph.get_out_csr()
    u = graph.get_separate_unique_node_indices()
    exp, s, ret = _alloc(graph, feat_src, el)
    return RelationalFusedGatCSR.apply(i["row_ptrs"], i["col_indices"], i["eids"], i["rel_types"], o["row_ptrs"],
                                       o["col_indices"], o["eids"], o["rel_types"], u["rel_ptrs"], u["node_indices"],
                                       feat_src.contiguous(), el.contiguous(), er.contiguous(), s, exp, ret, slope)


def relational_fused_gat_separate_coo(g, feat, el, er, negative_slope):
    # reference: rgat_layers_and_funcs.py:725-749
    d = g.get_separate_coo_original()
    exp, s, ret = _alloc(g, feat, el)
    return RelationalFusedGatSeparateCOO.apply(d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"],
                                               feat.contiguous(), el.contiguous(), er.contiguous(), s, exp, ret,
                                               negative_slope)


def relational_fused_gat_compact_as_of_node_separate_coo_dual_unique_node_list(g, feat_compact, el_compact, er_compact,
                                                                               negative_slope):
    # reference: rgat_layers_and_funcs.py:752-789
    return relational_fused_gat_compact_as_of_node_separate_coo_single_sided(g, feat_compact, el_compact, er_compact,
                                                                             negative_slope, False)


def relational_fused_gat_compact_as_of_node_separate_coo_single_sided(g, feat_compact, el_compact, er_compact,
                                                                      negative_slope, compact_direct_indexing_flag):
    # reference: rgat_layers_and_funcs.py:826-890
    d = g.get_separate_coo_original()
    exp, s, ret = _alloc(g, feat_compact, el_compact)
    f, l, r = feat_compact.contiguous(), el_compact.contiguous(), er_compact.contiguous()
    if compact_direct_indexing_flag:
        inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
        return RelationalFusedGatCompactAsOfNodeSeparateCOODualUniqueNodeListDirectIndexing.apply(
            d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"], inv["inverse_indices_row"],
            inv["inverse_indices_col"], f, l, r, s, exp, ret, negative_slope)
    ss = g.get_separate_unique_node_indices_single_sided()
    return RelationalFusedGatCompactAsOfNodeSeparateCOODualUniqueNodeList.apply(
        d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"], ss["rel_ptrs_row"], ss["node_indices_row"],
        ss["rel_ptrs_col"], ss["node_indices_col"], f, l, r, s, exp, ret, negative_slope)
