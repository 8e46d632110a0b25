"""Typed linear projections (segment GEMM) behind torch.autograd.

Mirrors /root/reference/hrt/python/backend/rgnn_layers_and_funcs.py: class names,
``forward(ctx, *index_tensors, *data, *outputs, *flags)`` argument order, the
transposed-weight / zero-filled-gradient protocol of ``backward`` (:43-73) and
the wrappers that allocate the outputs (:423-493)."""
import torch as th

from ..plan import consistent as _consistent_plan

from .. import kernels as _k
from ..kernels import K

__all__ = [
    "RgnnRelationalMatmul", "RgnnRelationalMatmulNoScatterGatherList", "RgnnRelationalMatmulCompactAsOfNode",
    "rgnn_relational_matmul", "rgnn_relational_matmul_no_scatter_gather_list",
    "rgnn_relational_matmul_with_attn_dot", "rgnn_relational_matmul_with_attn_dot_ok",
    "rgnn_relational_matmul_attn_dot_only", "rgnn_relational_matmul_attn_dot_only_ok",
]


@_consistent_plan
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


@_consistent_plan
class _RgnnRelationalMatmulWithAttnDot(th.autograd.Function):
    """Per-edge projection (RgnnRelationalMatmul, one input head, kind 0) that also returns the attention term
    dot[e, h] = <feat[e, h, :], attn[r, h, :]> from the GEMM epilogue (RGAT/models.py:288-296 forms it with a second
    segment GEMM over the tensor just written).  ``folded``: the caller hands ``dot`` and ``attn`` to
    relational_fused_gat_separate_coo_with_attn_l, whose backward delivers the gradients through the dot product
    (into grad feat and grad attn) itself -- ``dot`` is then marked non-differentiable here."""

    @staticmethod
    def forward(ctx, relptrs, node_indices, eids, weights, inputs, attn, folded):
        E, H, D = node_indices.numel(), weights.size(1), weights.size(3)
        ret = th.empty((E, H, D), dtype=weights.dtype, device=weights.device)
        dot = th.empty((E, H), dtype=weights.dtype, device=weights.device)
        d = {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids}
        _k.matmul_attn_dot(d, 0, weights, inputs, ret, attn, dot)
        ctx.folded = folded
        ctx.set_materialize_grads(False)  # an unused output's gradient stays None (no [E,H,D] zero tensor)
        if folded:
            ctx.save_for_backward(relptrs, node_indices, eids, weights, inputs)
            ctx.mark_non_differentiable(dot)
        else:
            ctx.save_for_backward(relptrs, node_indices, eids, weights, inputs, attn, ret)
        return ret, dot

    @staticmethod
    def backward(ctx, grad_ret, grad_dot):
        if ctx.folded or grad_dot is None:
            relptrs, node_indices, eids, weights, inputs = ctx.saved_tensors[:5]
            grad_attn = None
            if grad_ret is None:
                return None, None, None, None, None, None, None
            gradout = grad_ret.contiguous()
        else:
            relptrs, node_indices, eids, weights, inputs, attn, ret = ctx.saved_tensors
            # gradient through dot = <ret, attn>: grad_ret += grad_dot (x) attn[r]; grad_attn[r] = SUM grad_dot * ret
            by_eid = {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": eids, "separate_coo_eids": eids}
            grad_attn = th.empty_like(attn)
            gd = grad_dot.contiguous()
            if grad_ret is None:
                # only the attention term was used: its gradient through the projection is rank one per head, so the
                # (relation, node) sums run over the [E,H] gradient (include/het_amd.h: ..._attn_dot_only)
                grad_weight = th.empty_like(weights, memory_format=th.contiguous_format)
                grad_input = th.empty_like(inputs, memory_format=th.contiguous_format)
                d = {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids}
                if _k.matmul_attn_dot_only_backward(d, th.transpose(weights, 2, 3).contiguous(), inputs, attn, gd,
                                                    grad_input, grad_weight):
                    _k.matmul_backward(by_eid, 0, attn.unsqueeze(2), ret, gd, None, grad_attn.unsqueeze(-1), False,
                                       accumulate=False)
                    return None, None, None, grad_weight, grad_input, grad_attn, None
                gradout = th.empty_like(ret)
                _k.matmul_backward(by_eid, 0, attn.unsqueeze(2), ret, gd, gradout, grad_attn.unsqueeze(-1), False,
                                   accumulate=False)
            else:
                gradout = grad_ret.contiguous().clone()
                grad_attn.zero_()
                _k.matmul_backward(by_eid, 0, attn.unsqueeze(2), ret, gd, gradout, grad_attn.unsqueeze(-1), False,
                                   accumulate=True)
        grad_weight = th.empty_like(weights, memory_format=th.contiguous_format)
        grad_input = th.empty_like(inputs, memory_format=th.contiguous_format)
        _k.matmul_backward(
            {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids},
            0, th.transpose(weights, 2, 3).contiguous(), inputs, gradout, grad_input, grad_weight, True, accumulate=False)
        return None, None, None, grad_weight, grad_input, grad_attn, None


@_consistent_plan
class _RgnnRelationalMatmulAttnDotOnly(th.autograd.Function):
    """dot[e, h] = <x[node(e)] . W[r, h], attn[r, h, :]> per edge WITHOUT the per-edge projection tensor: the S distinct
    (relation, node) rows are projected once and kept ([S,H,D]), only their [S,H] dots are duplicated to the edges.
    For RGAT's er under the default flags (RGAT/models.py:300-308), whose per-edge projection feeds nothing else.
    Same values as RgnnRelationalMatmul + the D_out = 1 RgnnRelationalMatmul."""

    @staticmethod
    def forward(ctx, relptrs, node_indices, eids, weights, inputs, attn):
        E, H = node_indices.numel(), weights.size(1)
        dot = th.empty((E, H), dtype=weights.dtype, device=weights.device)
        d = {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids}
        comp = _k.matmul_attn_dot(d, 0, weights, inputs, None, attn, dot)
        ctx.save_for_backward(relptrs, node_indices, eids, weights, inputs, attn, comp)
        return dot

    @staticmethod
    def backward(ctx, grad_dot):
        relptrs, node_indices, eids, weights, inputs, attn, comp = ctx.saved_tensors
        d = {"separate_coo_rel_ptrs": relptrs, "separate_coo_node_indices": node_indices, "separate_coo_eids": eids}
        grad_weight = th.empty_like(weights, memory_format=th.contiguous_format)
        grad_input = th.empty_like(inputs, memory_format=th.contiguous_format)
        grad_attn = th.empty_like(attn)
        ok = _k.matmul_attn_dot_only_backward(d, th.transpose(weights, 2, 3).contiguous(), inputs, attn, grad_dot.contiguous(),
                                              grad_input, grad_weight, comp_rows=comp, grad_dot_w=grad_attn)
        assert ok, "the grouping of the forward pass is gone"
        return None, None, None, grad_weight, grad_input, grad_attn


def rgnn_relational_matmul_attn_dot_only_ok(arg_tensor_dict, weights, inputs):
    return inputs.dim() == 2 and _k.matmul_attn_dot_only_ok(arg_tensor_dict, weights, inputs)


def rgnn_relational_matmul_attn_dot_only(arg_tensor_dict, weights, inputs, attn):
    return _RgnnRelationalMatmulAttnDotOnly.apply(
        arg_tensor_dict["separate_coo_rel_ptrs"], arg_tensor_dict["separate_coo_node_indices"],
        arg_tensor_dict["separate_coo_eids"], weights.contiguous(), inputs.contiguous(), attn.contiguous())


def rgnn_relational_matmul_with_attn_dot_ok(weights, inputs):
    return inputs.is_cuda and inputs.dim() == 2 and _k.matmul_attn_dot_ok(weights.size(1), weights.size(2), weights.size(3))


def rgnn_relational_matmul_with_attn_dot(arg_tensor_dict, weights, inputs, attn, folded=False):
    """(feat [E,H,D], dot [E,H]) for the separate COO lists of ``arg_tensor_dict`` (kind 0)."""
    return _RgnnRelationalMatmulWithAttnDot.apply(
        arg_tensor_dict["separate_coo_rel_ptrs"], arg_tensor_dict["separate_coo_node_indices"],
        arg_tensor_dict["separate_coo_eids"], weights.contiguous(), inputs.contiguous(), attn.contiguous(), folded)


@_consistent_plan
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


@_consistent_plan
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
            ctx.input_num_head_one_flag, accumulate=False,
            distinct_rows=True)  # the wrapper's contract (its argument names): a UNIQUE (relation, node) list
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
