"""Typed linear projections (segment GEMM) behind torch.autograd.

Mirrors /root/reference/hrt/python/backend/rgnn_layers_and_funcs.py: class names,
``forward(ctx, *index_tensors, *data, *outputs, *flags)`` argument order, the
transposed-weight / zero-filled-gradient protocol of ``backward`` (:43-73) and
the wrappers that allocate the outputs (:423-493)."""
import torch as th

from .. import kernels as _k
from ..kernels import K

__all__ = [
    "RgnnRelationalMatmul", "RgnnRelationalMatmulNoScatterGatherList", "RgnnRelationalMatmulCompactAsOfNode",
    "rgnn_relational_matmul", "rgnn_relational_matmul_no_scatter_gather_list",
]


class RgnnRelationalMatmul(th.autograd.Function):
    # reference: rgnn_layers_and_funcs.py:8-73
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
        # the reference zero-fills both and lets the op accumulate (:52-70); with accumulate=False the op
        # overwrites them (zeroing internally only where its kernels need it)
        grad_weight = th.empty_like(weights, memory_format=th.contiguous_format)
        grad_input = th.empty_like(inputs, memory_format=th.contiguous_format)
        _k.matmul_backward(
            {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids},
            0, th.transpose(weights, 2, 3).contiguous(), inputs, gradout.contiguous(), grad_input, grad_weight,
            ctx.input_num_head_one_flag, accumulate=False)
        return None, None, None, grad_weight, grad_input, None, None


class RgnnRelationalMatmulNoScatterGatherList(th.autograd.Function):
    # reference: rgnn_layers_and_funcs.py:76-118
    @staticmethod
    def forward(ctx, ntype_offset_ptrs, weights, inputs, ret):
        K.rgnn_relational_matmul_no_scatter_gather_list(ntype_offset_ptrs, weights, inputs, ret)
        ctx.save_for_backward(ntype_offset_ptrs, weights, inputs)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        ntype_offset_ptrs, weights, inputs = ctx.saved_tensors
        grad_weight = th.empty_like(weights, memory_format=th.contiguous_format)
        grad_input = th.empty_like(inputs, memory_format=th.contiguous_format)
        _k.matmul_no_scatter_gather_backward(
            ntype_offset_ptrs, th.transpose(weights, 2, 3).contiguous(), inputs, gradout.contiguous(), grad_input,
            grad_weight, accumulate=False)
        return None, grad_weight, grad_input, None


class RgnnRelationalMatmulCompactAsOfNode(th.autograd.Function):
    # reference: rgnn_layers_and_funcs.py:121-189
    @staticmethod
    def forward(ctx, unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, weight, node_feat, ret,
                input_num_head_one_flag):
        ctx.save_for_backward(unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, weight, node_feat)
        ctx.input_num_head_one_flag = input_num_head_one_flag
        K.rgnn_relational_matmul(
            {"unique_srcs_and_dests_rel_ptrs": unique_srcs_and_dests_rel_ptrs,
             "unique_srcs_and_dests_node_indices": unique_srcs_and_dests_node_indices},
            1, weight, node_feat, ret, input_num_head_one_flag)
        return ret

    @staticmethod
    def backward(ctx, gradout):
        rel_ptrs, node_indices, weight, node_feat = ctx.saved_tensors
        grad_weight = th.empty_like(weight, memory_format=th.contiguous_format)
        grad_node_feat = th.empty_like(node_feat, memory_format=th.contiguous_format)
        _k.matmul_backward(
            {"unique_srcs_and_dests_rel_ptrs": rel_ptrs, "unique_srcs_and_dests_node_indices": node_indices},
            1, th.transpose(weight, 2, 3).contiguous(), node_feat, gradout.contiguous(), grad_node_feat, grad_weight,
            ctx.input_num_head_one_flag, accumulate=False)
        return None, None, grad_weight, grad_node_feat, None, None


def rgnn_relational_matmul(arg_tensor_dict, weights, inputs, input_num_head_one_flag, compact_as_of_node_kind):
    # reference: rgnn_layers_and_funcs.py:423-471 (allocates ret zero-filled; the op overwrites every row, so
    # th.empty is enough here)
    if compact_as_of_node_kind == 1:
        ret = th.empty((int(arg_tensor_dict["unique_srcs_and_dests_node_indices"].numel()), weights.size(1), weights.size(3)),
                       dtype=weights.dtype, device=weights.device)
        return RgnnRelationalMatmulCompactAsOfNode.apply(
            arg_tensor_dict["unique_srcs_and_dests_rel_ptrs"], arg_tensor_dict["unique_srcs_and_dests_node_indices"],
            weights.contiguous(), inputs.contiguous(), ret, input_num_head_one_flag)
    if compact_as_of_node_kind == 0:
        ret = th.empty((arg_tensor_dict["separate_coo_node_indices"].numel(), weights.size(1), weights.size(3)),
                       dtype=weights.dtype, device=weights.device)
        return RgnnRelationalMatmul.apply(
            arg_tensor_dict["separate_coo_rel_ptrs"], arg_tensor_dict["separate_coo_node_indices"],
            arg_tensor_dict["separate_coo_eids"], weights.contiguous(), inputs.contiguous(), ret,
            input_num_head_one_flag)
    raise NotImplementedError


def rgnn_relational_matmul_no_scatter_gather_list(ntype_offset_ptrs, weights, inputs):
    # reference: rgnn_layers_and_funcs.py:474-493.  ret is [rows, D] for single-head weights (as the
    # reference), [rows, H, D] otherwise.
    H, D = weights.size(1), weights.size(3)
    shape = (inputs.size(0), D) if H == 1 and inputs.dim() == 2 else (inputs.size(0), H, D)
    ret = th.empty(shape, dtype=weights.dtype, device=weights.device)
    return RgnnRelationalMatmulNoScatterGatherList.apply(ntype_offset_ptrs, weights.contiguous(), inputs.contiguous(), ret)
