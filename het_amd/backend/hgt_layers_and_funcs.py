"""HGT ops behind torch.autograd; mirrors /root/reference/hrt/python/backend/hgt_layers_and_funcs.py
(HGTFullGraphHeteroAttentionOps :9-121, HGTFullGraphMessageCalcEdgeSoftmaxAndMessageMeanAggregationCOO :124-295,
wrappers :425-503) and the inner-product classes of rgnn_layers_and_funcs.py:192-418, 499-591."""
import torch as th

from ..plan import consistent as _consistent_plan

from .. import kernels as _k
from ..kernels import K

__all__ = [
    "HGTFullGraphHeteroAttentionOps", "HGTFullGraphMessageCalcEdgeSoftmaxAndMessageMeanAggregationCOO",
    "hgt_full_graph_hetero_attention_ops_coo",
    "hgt_full_graph_message_calc_edge_softmax_and_message_mean_aggregation_coo",
    "hgt_full_graph_edge_softmax_and_message_mean_aggregation_csr",
    "RgnnInnerProductEdgeAndNode", "RgnnInnerProductNodeCompactAndNode",
    "RgnnInnerProductNodeCompactAndNodeWithDirectIndexing", "rgnn_inner_product_right_node",
]


@_consistent_plan
class HGTFullGraphHeteroAttentionOps(th.autograd.Function):
    # reference: hgt_layers_and_funcs.py:9-121 (the in-CSR arguments are carried for its vertex-parallel dq kernel)
    @staticmethod
    def forward(ctx, incsr_row_ptrs, incsr_col_indices, incsr_eids, incsr_reltypes, separate_coo_row_indices,
                separate_coo_col_indices, separate_coo_eids, separate_coo_relptrs, applied_klinear_node_features,
                applied_qlinear_node_features, attn_score_weight):
        E, H = separate_coo_row_indices.numel(), attn_score_weight.size(1)
        score = th.empty((E, H), dtype=attn_score_weight.dtype, device=attn_score_weight.device)
        inner = th.empty((E, H, attn_score_weight.size(3)), dtype=score.dtype, device=score.device)
        K.hgt_full_graph_hetero_attention_ops_coo(separate_coo_row_indices, separate_coo_col_indices, separate_coo_eids,
                                                  separate_coo_relptrs, applied_klinear_node_features,
                                                  applied_qlinear_node_features, attn_score_weight, inner, score)
        ctx.save_for_backward(incsr_row_ptrs, incsr_col_indices, incsr_eids, incsr_reltypes, separate_coo_row_indices,
                              separate_coo_col_indices, separate_coo_eids, separate_coo_relptrs,
                              applied_klinear_node_features, applied_qlinear_node_features, inner, attn_score_weight)
        return score

    @staticmethod
    def backward(ctx, grad_score):
        (ip, ic, ie, ir, row, col, eids, relptrs, k, q, inner, weight) = ctx.saved_tensors
        grad_w = th.zeros_like(weight, memory_format=th.contiguous_format)
        grad_k = th.zeros_like(k, memory_format=th.contiguous_format)
        grad_q = th.zeros_like(q, memory_format=th.contiguous_format)
        K.backward_hgt_full_graph_hetero_attention_ops_coo(ip, ic, ie, ir, row, col, eids, relptrs, grad_w,
                                                           th.transpose(weight, 2, 3).contiguous(), k, q, inner,
                                                           grad_score.contiguous(), grad_k, grad_q)
        return None, None, None, None, None, None, None, None, grad_k, grad_q, grad_w


@_consistent_plan
class HGTFullGraphMessageCalcEdgeSoftmaxAndMessageMeanAggregationCOO(th.autograd.Function):
    # reference: hgt_layers_and_funcs.py:124-295
    @staticmethod
    def forward(ctx, incsr_row_ptrs, incsr_col_indices, incsr_eids, incsr_reltypes, separate_coo_relptrs,
                separate_coo_row_indices, separate_coo_col_indices, separate_coo_eids, message_generation_weights,
                inputs, unnormalized_attn_score, edgesoftmax_sum_per_node, mu,
                mu_softmax_applied_unnormalized_attn_score, normalized_attn_score, new_h):
        K.hgt_full_graph_edge_softmax_ops_separate_coo(separate_coo_row_indices, separate_coo_col_indices,
                                                       separate_coo_eids, separate_coo_relptrs, unnormalized_attn_score,
                                                       mu, edgesoftmax_sum_per_node,
                                                       mu_softmax_applied_unnormalized_attn_score, normalized_attn_score)
        K.hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
            separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices, inputs,
            message_generation_weights, normalized_attn_score, new_h)
        ctx.save_for_backward(separate_coo_relptrs, separate_coo_row_indices, separate_coo_col_indices, separate_coo_eids,
                              message_generation_weights, inputs, unnormalized_attn_score, mu, normalized_attn_score,
                              new_h)
        return new_h

    @staticmethod
    def backward(ctx, gradout):
        (relptrs, row, col, eids, weights, inputs, score, mu, a, new_h) = ctx.saved_tensors
        grad_w = th.zeros_like(weights, memory_format=th.contiguous_format)
        grad_input = th.zeros_like(inputs, memory_format=th.contiguous_format)
        grad_a = th.empty_like(a, memory_format=th.contiguous_format)
        grad_score = th.empty_like(score, memory_format=th.contiguous_format)
        grad_mu = th.zeros_like(mu, memory_format=th.contiguous_format)
        K.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
            relptrs, eids, row, col, inputs, th.transpose(weights, 2, 3).contiguous(), a, new_h, grad_input, grad_w,
            grad_a, gradout.contiguous())
        tmp = th.empty([new_h.shape[0], score.shape[1]], dtype=a.dtype, device=a.device)
        K.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(row, col, eids, relptrs, score, a,
                                                                                grad_a, mu, grad_score, grad_mu, tmp)
        return (None,) * 8 + (grad_w, grad_input, grad_score, None, grad_mu, None, None, None)


def hgt_full_graph_hetero_attention_ops_coo(graph, weight, applied_klinear_node_features, applied_qlinear_node_features):
    # reference: hgt_layers_and_funcs.py:425-443
    s, i = graph.get_separate_coo_original(), graph.get_in_csr()
    return HGTFullGraphHeteroAttentionOps.apply(i["row_ptrs"], i["col_indices"], i["eids"], i["rel_types"],
                                                s["row_indices"], s["col_indices"], s["eids"], s["rel_ptrs"],
                                                applied_klinear_node_features.contiguous(),
                                                applied_qlinear_node_features.contiguous(), weight.contiguous())


def hgt_full_graph_message_calc_edge_softmax_and_message_mean_aggregation_coo(relation_meg_weight, inputs, graph, mu,
                                                                              unnormalized_attn_score):
    # reference: hgt_layers_and_funcs.py:446-503
    s, i = graph.get_separate_coo_original(), graph.get_in_csr()
    N, H = graph.get_num_nodes(), relation_meg_weight.size(1)
    new_h = th.zeros(N, H, relation_meg_weight.size(3), dtype=relation_meg_weight.dtype, device=relation_meg_weight.device)
    sum_per_node = th.empty(N, mu.size(1), dtype=relation_meg_weight.dtype, device=relation_meg_weight.device)
    score = unnormalized_attn_score.contiguous()
    m, a = th.empty_like(score), th.empty_like(score)
    return HGTFullGraphMessageCalcEdgeSoftmaxAndMessageMeanAggregationCOO.apply(
        i["row_ptrs"], i["col_indices"], i["eids"], i["rel_types"], s["rel_ptrs"], s["row_indices"], s["col_indices"],
        s["eids"], relation_meg_weight.contiguous(), inputs.contiguous(), score, sum_per_node, mu.contiguous(), m, a, new_h)


def hgt_full_graph_edge_softmax_and_message_mean_aggregation_csr(graph, message_per_edge, unnormalized_attn_score, mu):
    """The reference's UNFUSED aggregation of HGT (hgt_layers_and_funcs.py:506-570, reached from HGT/models.py:271 when
    ``--fused_message_mean_aggregation_flag`` is off: per-edge messages [E,H,dk] from a kind-0 rgnn_relational_matmul, then
    the CSR edge softmax + aggregation ops ``hgt_full_graph_edge_softmax_ops_csr`` / ``hgt_full_graph_message_mean_aggregation_csr``,
    which are outside SURVEY.md 8b's must-export set and not built).  Named here so that the reference's model code fails
    with an explanation, not an AttributeError."""
    from .._lib import HetUnsupported
    raise HetUnsupported("hgt_full_graph_edge_softmax_and_message_mean_aggregation_csr: the CSR edge-softmax / aggregation ops of HGT's "
                         "unfused path (fused_message_mean_aggregation_flag=False) are not built; use the reference's default, "
                         "fused_message_mean_aggregation_flag=True (hgt_full_graph_message_calc_edge_softmax_and_message_mean_"
                         "aggregation_coo) -- same layer output, no [E,H,dk] message tensor")


@_consistent_plan
class _InnerProduct(th.autograd.Function):
    @staticmethod
    def forward(ctx, kind, map_a, map_b, rel_ptrs, eids, row, col, left, right, ret):
        ctx.kind = kind
        ctx.save_for_backward(map_a, map_b, rel_ptrs, eids, row, col, left, right)
        K.rgnn_inner_product_right_node_separatecoo(_ip_dict(kind, map_a, map_b), kind, rel_ptrs, eids, row, col, left,
                                                    right, ret)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        map_a, map_b, rel_ptrs, eids, row, col, left, right = ctx.saved_tensors
        grad_left = th.empty_like(left, memory_format=th.contiguous_format)
        grad_right = th.empty_like(right, memory_format=th.contiguous_format)
        _k.inner_product_backward(_ip_dict(ctx.kind, map_a, map_b), ctx.kind, rel_ptrs, eids, row, col, left, right,
                                  gradout.contiguous(), grad_left, grad_right, accumulate=False)
        return None, None, None, None, None, None, None, grad_left, grad_right, None


def _ip_dict(kind, map_a, map_b):
    if kind == 0:
        return {}
    if kind == 1:
        return {"unique_srcs_and_dests_rel_ptrs": map_a, "unique_srcs_and_dests_node_indices": map_b}
    return {"edata_idx_to_inverse_idx": map_a}


class RgnnInnerProductEdgeAndNode:
    # reference: rgnn_layers_and_funcs.py:351-418
    @staticmethod
    def apply(rel_ptrs, eids, row, col, left_edge_data, right_node_vectors, ret):
        return _InnerProduct.apply(0, rel_ptrs, rel_ptrs, rel_ptrs, eids, row, col, left_edge_data, right_node_vectors, ret)


class RgnnInnerProductNodeCompactAndNode:
    # reference: rgnn_layers_and_funcs.py:192-270
    @staticmethod
    def apply(u_rel_ptrs, u_node_indices, rel_ptrs, eids, row, col, left_node_compact_data, right_node_vectors, ret):
        return _InnerProduct.apply(1, u_rel_ptrs, u_node_indices, rel_ptrs, eids, row, col, left_node_compact_data,
                                   right_node_vectors, ret)


class RgnnInnerProductNodeCompactAndNodeWithDirectIndexing:
    # reference: rgnn_layers_and_funcs.py:273-348
    @staticmethod
    def apply(edata_index_to_inverse_index, rel_ptrs, eids, row, col, left_node_compact_data, right_node_vectors, ret):
        return _InnerProduct.apply(2, edata_index_to_inverse_index, edata_index_to_inverse_index, rel_ptrs, eids, row,
                                   col, left_node_compact_data, right_node_vectors, ret)


def rgnn_inner_product_right_node(graph, left_side_data, right_node_vectors, compact_as_of_node_kind, left_mapper_suffix):
    # reference: rgnn_layers_and_funcs.py:499-591
    s = graph.get_separate_coo_original()
    ret = th.empty([s["eids"].numel(), right_node_vectors.size(1)], dtype=right_node_vectors.dtype,
                   device=right_node_vectors.device)
    left, right = left_side_data.contiguous(), right_node_vectors.contiguous()
    if compact_as_of_node_kind == 0:
        return RgnnInnerProductEdgeAndNode.apply(s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], left, right, ret)
    if compact_as_of_node_kind == 1:
        ss = graph.get_separate_unique_node_indices_single_sided()
        return RgnnInnerProductNodeCompactAndNode.apply(ss["rel_ptrs" + left_mapper_suffix],
                                                        ss["node_indices" + left_mapper_suffix], s["rel_ptrs"],
                                                        s["eids"], s["row_indices"], s["col_indices"], left, right, ret)
    if compact_as_of_node_kind == 2:
        inv = graph.get_separate_unique_node_indices_single_sided_inverse_idx()
        return RgnnInnerProductNodeCompactAndNodeWithDirectIndexing.apply(
            inv["inverse_indices" + left_mapper_suffix], s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"],
            left, right, ret)
    raise NotImplementedError
