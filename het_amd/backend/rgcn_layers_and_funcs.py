"""RGCN ops behind torch.autograd; mirrors
/root/reference/hrt/python/backend/rgcn_layers_and_funcs.py:481-824."""
import os

import torch as th

from ..plan import consistent as _consistent_plan

from .. import kernels as _k
from ..kernels import K

# the layer as two library calls (include/het_amd.h: het_rgcn_layer_forward / _backward); HET_RGCN_FUSED=0: the a7 / a8 pair
FUSED = os.environ.get("HET_RGCN_FUSED", "1") != "0"

__all__ = [
    "RgcnLayer1SeparateCoo", "RgcnLayerFused", "rgcn_layer1_separate_coo", "RGCNNodeMeanAggregationCompactAsOfNodeSeparateCOO",
    "RGCNNodeMeanAggregationCompactAsOfNodeDirectIndexingSeparateCOO",
    "rgcn_node_mean_aggregation_compact_as_of_node_separate_coo_single_sided",
]


@_consistent_plan
class RgcnLayer1SeparateCoo(th.autograd.Function):
    # reference: rgcn_layers_and_funcs.py:481-567 (the out-CSR arguments are carried but unused there too)
    @staticmethod
    def forward(ctx, separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices,
                outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes, x, weight, norm, ret):
        ctx.save_for_backward(separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices,
                              separate_coo_col_indices, weight, norm, x)
        K.rgcn_layer1_separate_coo(separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices,
                                   separate_coo_col_indices, x, weight, norm, ret)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        rel_ptrs, eids, row, col, weight, norm, x = ctx.saved_tensors
        grad_x = th.zeros_like(x, memory_format=th.contiguous_format)
        grad_weight = th.zeros_like(weight, memory_format=th.contiguous_format)
        grad_norm = th.zeros_like(norm, memory_format=th.contiguous_format)
        K.backward_rgcn_layer1_separate_coo(rel_ptrs, eids, row, col, x, th.transpose(weight, 1, 2).contiguous(), norm,
                                            grad_norm, grad_x, gradout.contiguous(), grad_weight)
        return None, None, None, None, None, None, None, None, grad_x, grad_weight, grad_norm, None


@_consistent_plan
class RgcnLayer1SeparateCooBias(th.autograd.Function):
    """RgcnLayer1SeparateCoo with the layer's bias (RGCN/RGCN.py:338-340, ``node_repr + h_bias``) inside the node: the op
    accumulates into ``ret`` (reference contract), so ``ret`` starts from the bias rows instead of zeros -- one fill instead
    of a fill and an elementwise add; the bias gradient is the column sum of gradout."""

    @staticmethod
    def forward(ctx, rel_ptrs, eids, row, col, num_nodes, x, weight, norm, bias):
        ctx.save_for_backward(rel_ptrs, eids, row, col, weight, norm, x)
        ret = bias.to(weight.dtype).expand(num_nodes, weight.size(2)).contiguous()
        K.rgcn_layer1_separate_coo(rel_ptrs, eids, row, col, x, weight, norm, ret)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        rel_ptrs, eids, row, col, weight, norm, x = ctx.saved_tensors
        gradout = gradout.contiguous()
        grad_x = th.zeros_like(x, memory_format=th.contiguous_format)
        grad_weight = th.zeros_like(weight, memory_format=th.contiguous_format)
        grad_norm = th.zeros_like(norm, memory_format=th.contiguous_format)
        K.backward_rgcn_layer1_separate_coo(rel_ptrs, eids, row, col, x, th.transpose(weight, 1, 2).contiguous(), norm,
                                            grad_norm, grad_x, gradout, grad_weight)
        return None, None, None, None, None, grad_x, grad_weight, grad_norm, gradout.sum(0)


@_consistent_plan
class RgcnLayerFused(th.autograd.Function):
    """rgcn_layer1_separate_coo (+ the layer's bias, RGCN/RGCN.py:338-340) as two library calls: the sums of the scaled source
    rows per (relation, destination) are kept from the forward -- the weight gradient is formed from them (half as many rows as
    the (relation, source) sums of a8, and x is not read again) -- the output and the input gradient are written once by a pass
    over the nodes (no zero-filled buffers, no read-modify-write per relation, no transposed-weight / bias-sum torch kernels):
    ogbn-mag, feat 64: 3.0 -> 2.3 ms per step.  Same values as RgcnLayer1SeparateCooBias up to the order of the fp32 sums."""

    @staticmethod
    def forward(ctx, plan, x, weight, norm, bias):
        ret, ssum = _k.rgcn_layer_forward(plan, x, weight, norm, bias)
        ctx.plan, ctx.has_bias = plan, bias is not None
        ctx.save_for_backward(weight, norm, ssum)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        weight, norm, ssum = ctx.saved_tensors
        grad_x, grad_w, grad_bias = _k.rgcn_layer_backward(ctx.plan, ssum, weight.transpose(1, 2).contiguous(), norm,
                                                           gradout.contiguous(), ctx.has_bias, want_x=ctx.needs_input_grad[1])
        # (the reference's a8 takes a grad_norm buffer and leaves it as it was given: zeros)
        grad_norm = th.zeros_like(norm) if ctx.needs_input_grad[3] else None
        return None, grad_x, grad_w, grad_norm, grad_bias


def _fused_plan(graph, s, x, weight, norm, bias):
    """The plan of the two-call layer when it applies: float32 GPU tensors of the graph's size, shapes the node-major pass takes,
    groupings switched on."""
    if not (FUSED and x.is_cuda and x.dim() == 2 and weight.dim() == 3 and _k._plan.is_enabled()):
        return None
    N, E = graph.get_num_nodes(), s["eids"].numel()
    R, Kin, D = weight.shape
    if not (E > 0 and x.shape == (N, Kin) and norm.numel() == E and x.dtype == weight.dtype == norm.dtype == th.float32 and
            (bias is None or (bias.dtype == th.float32 and bias.numel() == D)) and R == s["rel_ptrs"].numel() - 1 and
            _k.rgcn_layer_ok(R, Kin, D)):
        return None
    return _k.rgcn_layer_plan(s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], N)


def rgcn_layer1_separate_coo(graph, x, weight, norm, bias=None):
    # reference: rgcn_layers_and_funcs.py:570-601
    s = graph.get_separate_coo_original()
    plan = _fused_plan(graph, s, x, weight, norm, bias)
    if plan is not None:
        return RgcnLayerFused.apply(plan, x.contiguous(), weight.contiguous(), norm.contiguous(),
                                    None if bias is None else bias.contiguous())
    if bias is not None:
        return RgcnLayer1SeparateCooBias.apply(s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], graph.get_num_nodes(),
                                               x.contiguous(), weight.contiguous(), norm.contiguous(), bias)
    o = graph.get_out_csr()
    ret = th.zeros((graph.get_num_nodes(), weight.size(2)), dtype=weight.dtype, device=weight.device)
    return RgcnLayer1SeparateCoo.apply(s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], o["row_ptrs"],
                                       o["col_indices"], o["eids"], o["rel_types"], x.contiguous(), weight.contiguous(),
                                       norm.contiguous(), ret)


@_consistent_plan
class _RGCNCompactAgg(th.autograd.Function):
    @staticmethod
    def forward(ctx, eids, rel_ptrs, row, col, map_a, map_b, feat_src, enorm, ret, direct):
        ctx.save_for_backward(eids, rel_ptrs, row, col, map_a, map_b, feat_src, enorm, ret)
        ctx.direct = direct
        d = {"inverse_indices_row": map_a} if direct else {"rel_ptrs_row": map_a, "node_indices_row": map_b}
        K.rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(eids, rel_ptrs, row, col, d, feat_src, enorm, ret, direct)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        eids, rel_ptrs, row, col, map_a, map_b, feat_src, enorm, ret = ctx.saved_tensors
        d = {"inverse_indices_row": map_a} if ctx.direct else {"rel_ptrs_row": map_a, "node_indices_row": map_b}
        grad_feat_src = th.zeros_like(feat_src, memory_format=th.contiguous_format)
        K.backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(
            eids, rel_ptrs, row, col, d, feat_src, enorm, ret, gradout.contiguous(), grad_feat_src, ctx.direct)
        return None, None, None, None, None, None, grad_feat_src, None, None, None


class RGCNNodeMeanAggregationCompactAsOfNodeSeparateCOO(th.autograd.Function):
    # reference: rgcn_layers_and_funcs.py:604-681
    @staticmethod
    def forward(ctx, separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                separate_unique_node_indices_rel_ptrs, separate_unique_node_indices_node_indices, feat_src, enorm, ret):
        raise RuntimeError("use .apply")

    @classmethod
    def apply(cls, eids, rel_ptrs, row, col, u_rel_ptrs, u_node_indices, feat_src, enorm, ret):  # noqa: D102
        return _RGCNCompactAgg.apply(eids, rel_ptrs, row, col, u_rel_ptrs, u_node_indices, feat_src, enorm, ret, False)


class RGCNNodeMeanAggregationCompactAsOfNodeDirectIndexingSeparateCOO(th.autograd.Function):
    # reference: rgcn_layers_and_funcs.py:684-754
    @staticmethod
    def forward(ctx, *a):
        raise RuntimeError("use .apply")

    @classmethod
    def apply(cls, eids, rel_ptrs, row, col, inverse_indices_row, feat_src, enorm, ret):  # noqa: D102
        return _RGCNCompactAgg.apply(eids, rel_ptrs, row, col, inverse_indices_row, inverse_indices_row, feat_src, enorm,
                                     ret, True)


def rgcn_node_mean_aggregation_compact_as_of_node_separate_coo_single_sided(g, feat_compact_src, enorm,
                                                                            compact_direct_indexing_flag):
    # reference: rgcn_layers_and_funcs.py:782-824
    s = g.get_separate_coo_original()
    ret = th.empty([g.get_num_nodes()] + list(feat_compact_src.size()[1:]), dtype=feat_compact_src.dtype,
                   device=feat_compact_src.device)
    if compact_direct_indexing_flag:
        inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
        return RGCNNodeMeanAggregationCompactAsOfNodeDirectIndexingSeparateCOO.apply(
            s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], inv["inverse_indices_row"],
            feat_compact_src.contiguous(), enorm.contiguous(), ret)
    ss = g.get_separate_unique_node_indices_single_sided()
    return RGCNNodeMeanAggregationCompactAsOfNodeSeparateCOO.apply(
        s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], ss["rel_ptrs_row"], ss["node_indices_row"],
        feat_compact_src.contiguous(), enorm.contiguous(), ret)
