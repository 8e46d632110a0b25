"""The reference's own op sequence, literally: what a checkout of the reference runs after the one-line swap of
``hrt/python/kernels/__init__.py`` to this library -- and nothing more.

Every call below is a ``torch.ops.torch_hrt.<name>`` op the reference registers (``kernels.REGISTERED_OPS``), with the
reference wrappers' buffer protocol: outputs allocated by the wrapper (``th.zeros`` / ``new_empty`` exactly where the
reference uses them), gradients zero-filled and accumulated into ("+="), weights transposed with
``th.transpose(...).contiguous()``, the self-loop as ``th.matmul``, bias as a torch add.  None of this package's
extension entry points (``*_attn_dot``, ``fold_attn_l``, sorted ``exp``, the one-node layer) is used, so the time of
this path is the drop-in number: ``bench.py`` reports it as ``variants.reference_op_sequence``.

  RefRgnnRelationalMatmul          hrt/python/backend/rgnn_layers_and_funcs.py:8-73, wrapper :423-471
  RefRelationalFusedGatSeparateCOO hrt/python/backend/rgat_layers_and_funcs.py:233-316, wrapper :725-749
  rgat_layer_reference_sequence    hrt/python/RGAT/models.py:265-385 (default flags / --multiply_among_weights_first_flag)
"""
import torch as th

from ..plan import consistent as _consistent_plan

from ..kernels import K

__all__ = ["RefRgnnRelationalMatmul", "RefRelationalFusedGatSeparateCOO", "ref_rgnn_relational_matmul",
           "ref_relational_fused_gat_separate_coo", "rgat_layer_reference_sequence"]


@_consistent_plan
class RefRgnnRelationalMatmul(th.autograd.Function):
    @staticmethod
    def forward(ctx, separate_coo_relptrs, separate_coo_node_indices, separate_coo_eids, weights, inputs, ret,
                input_num_head_one_flag):
        ctx.save_for_backward(separate_coo_relptrs, separate_coo_node_indices, separate_coo_eids, weights, inputs)
        ctx.input_num_head_one_flag = input_num_head_one_flag
        K.rgnn_relational_matmul(
            {"separate_coo_rel_ptrs": separate_coo_relptrs, "separate_coo_node_indices": separate_coo_node_indices,
             "separate_coo_eids": separate_coo_eids},
            0, weights, inputs, ret, input_num_head_one_flag)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        relptrs, node_indices, eids, weights, inputs = ctx.saved_tensors
        grad_weight = th.zeros_like(weights, memory_format=th.contiguous_format)
        grad_input = th.zeros_like(inputs, memory_format=th.contiguous_format)
        K.backward_rgnn_relational_matmul(
            {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids},
            0, th.transpose(weights, 2, 3).contiguous(), inputs, gradout.contiguous(), grad_input, grad_weight,
            ctx.input_num_head_one_flag)
        return None, None, None, grad_weight, grad_input, None, None


@_consistent_plan
class RefRelationalFusedGatSeparateCOO(th.autograd.Function):
    @staticmethod
    def forward(ctx, eids, rel_ptrs, row, col, feat_src, el, er, s, exp, ret, slope):
        ctx.save_for_backward(eids, rel_ptrs, row, col, feat_src, el, er, s, exp, ret)
        ctx.slope = slope
        K.relational_fused_gat_separate_coo(eids, rel_ptrs, row, col, 0, {}, feat_src, el, er, s, exp, ret, slope)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        eids, rel_ptrs, row, col, feat_src, el, er, s, exp, ret = ctx.saved_tensors
        grad_el = th.zeros_like(el, memory_format=th.contiguous_format)
        grad_er = th.zeros_like(er, memory_format=th.contiguous_format)
        grad_feat_src = th.zeros_like(feat_src, memory_format=th.contiguous_format)
        K.backward_relational_fused_gat_separate_coo(eids, rel_ptrs, row, col, 0, {}, feat_src, el, er, s, exp, ret,
                                                     gradout.contiguous(), grad_feat_src, grad_el, grad_er, ctx.slope)
        return None, None, None, None, grad_feat_src, grad_el, grad_er, None, None, None, None


def ref_rgnn_relational_matmul(arg_tensor_dict, weights, inputs, input_num_head_one_flag):
    ret = th.zeros((arg_tensor_dict["separate_coo_node_indices"].numel(), weights.size(1), weights.size(3)),
                   dtype=weights.dtype, device=weights.device)
    return RefRgnnRelationalMatmul.apply(
        arg_tensor_dict["separate_coo_rel_ptrs"], arg_tensor_dict["separate_coo_node_indices"],
        arg_tensor_dict["separate_coo_eids"], weights.contiguous(), inputs.contiguous(), ret, input_num_head_one_flag)


def ref_relational_fused_gat_separate_coo(g, feat, el, er, negative_slope):
    d = g.get_separate_coo_original()
    exp = el.new_empty([g.get_num_edges()] + list(el.size()[1:]))
    s = el.new_empty([g.get_num_nodes()] + list(el.size()[1:]))
    ret = th.empty([g.get_num_nodes()] + list(feat.size()[1:]), dtype=feat.dtype, device=feat.device)
    return RefRelationalFusedGatSeparateCOO.apply(d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"], feat, el, er,
                                                  s, exp, ret, negative_slope)


def rgat_layer_reference_sequence(layer, g, inputs):
    """HET_RGATLayer.forward of the reference, non-compact branch (RGAT/models.py:265-385), on ``layer``'s parameters."""
    s = g.get_separate_coo_original()
    by_src = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"], "separate_coo_eids": s["eids"]}
    by_dst = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"], "separate_coo_eids": s["eids"]}
    by_eid = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["eids"], "separate_coo_eids": s["eids"]}
    H = layer.num_heads
    feat_src_per_edge = ref_rgnn_relational_matmul(by_src, layer.conv_weights, inputs, True)
    el = ref_rgnn_relational_matmul(by_eid, layer.attn_l.unsqueeze(-1), feat_src_per_edge, False)
    if layer.multiply_among_weights_first_flag:
        # (the reference passes InputNumHeadOneFlag=False here, which only addresses memory correctly for one head,
        # SURVEY.md Q5; the input has one head)
        er = ref_rgnn_relational_matmul(by_dst, layer._w_attn_r(), inputs, True)
    else:
        feat_dst_per_edge = ref_rgnn_relational_matmul(by_dst, layer.conv_weights, inputs, True)
        er = ref_rgnn_relational_matmul(by_eid, layer.attn_r.unsqueeze(-1), feat_dst_per_edge, False)
    h = ref_relational_fused_gat_separate_coo(g, feat_src_per_edge, el.view(-1, H), er.view(-1, H), layer.leaky_relu_slope)
    h = h.view(-1, layer.out_feat)
    if layer.self_loop:
        h = h + th.matmul(inputs, layer.loop_weight)
    if layer.bias:
        h = h + layer.h_bias
    if layer.activation:
        h = layer.activation(h)
    return layer.dropout(h)
