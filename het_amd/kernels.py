"""``K = torch.ops.torch_hrt``: the reference's custom-op namespace, served by
libhet_amd.so.

The reference loads ``libtorch_hrt.so`` and calls its kernels as
``torch.ops.torch_hrt.<name>(...)`` (hrt/python/kernels/__init__.py:4-16; op list:
``m.def`` lines of hrt/include/DGLHackKernel/OpExport/*.inc.h).  This module
defines ops with the same names, argument order and argument meaning in the same
namespace and implements them by calling the C ABI of include/het_amd.h on the
current torch stream.  Tensor arguments are only unwrapped to device pointers
here; all arithmetic happens in the HIP library.  CPU tensors are rejected for
the compute ops (there is no CPU path); the five layout converters run the native
device-side builders (csrc/layouts.hip) on GPU tensors and torch index ops on CPU tensors.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from . import _lib, graph as _graph, plan as _plan

Tensor = torch.Tensor
NAMESPACE = "torch_hrt"
# HET_TORCH_HRT_LIB=<path to libtorch_hrt.so>: take the reference-named ops from the COMPILED registration object
# (csrc/torch_export.cpp, `make -C het_amd/csrc torch_hrt`; loaded as the reference loads its own library,
# hrt/python/kernels/__init__.py:4-16) instead of defining them here.  The functions below then only serve this package's
# own callers (layers, backend); K.<op> goes through the compiled dispatcher entry.
import os as _os
COMPILED_LIB = _os.environ.get("HET_TORCH_HRT_LIB") or None
if COMPILED_LIB:
    torch.ops.load_library(COMPILED_LIB)
    _libdef = None
else:
    _libdef = torch.library.Library(NAMESPACE, "DEF")
_registered: List[str] = []
# The library's own device memory (groupings, their construction scratch) comes from torch's caching allocator: inside
# torch.cuda.memory_allocated, returned to torch's pool on eviction, no hipFree on an op's path (include/het_amd.h:
# het_set_allocator).  HET_TORCH_ALLOCATOR=0: hipMalloc, as a caller of the bare C ABI gets.
if _os.environ.get("HET_TORCH_ALLOCATOR", "1") != "0" and torch.cuda.is_available():
    _lib.use_torch_allocator()


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: Tensor):
    """The current stream of the tensor's device as the raw hipStream_t (an int: the C entry points take it as void*).  The raw
    getter skips building a torch.cuda.Stream object per call -- host time per launch is what a rank's step at 8 ranks and a
    sampled block's step are made of (exp/host_profile_rank.py)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(name: str, floats=(), ints=()):
    for t in floats:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.HetError(f"{name}: expected contiguous float32 GPU tensors, got {t.dtype} on {t.device}"
                                f"{'' if t.is_contiguous() else ' (non-contiguous)'}")
    for t in ints:
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise _lib.HetError(f"{name}: expected contiguous int64 GPU tensors, got {t.dtype} on {t.device}")


def _written_positions(schema: str):
    """Positions of the arguments a schema marks as written (``Tensor(a!) name``)."""
    params = schema[schema.index("(") + 1:schema.rindex(") ->")]
    out, depth, cur, pos = [], 0, "", 0
    for ch in params + ",":
        if ch == "," and depth == 0:
            if "!" in cur:
                out.append(pos)
            cur, pos = "", pos + 1
            continue
        depth += ch == "("
        depth -= ch == ")"
        cur += ch
    return tuple(out)


def _op(schema: str):
    name = schema.split("(")[0]
    written = _written_positions(schema)

    def deco(fn):
        impl = fn
        if written:
            # The dispatcher does not bump the version counter of a custom op's ``(a!)`` arguments, and the library writes through
            # raw pointers: caches keyed by (data_ptr, numel, _version) -- scale_in_rank_order, the sorted exp stream of a4 / a5 --
            # would keep serving a copy of a buffer that another op of this library has refilled in place.  Bumped
            # BEFORE the call, so that what the op itself records about its outputs (the sorted stream's key) is the new version.
            def impl(*args, **kwargs):
                for i in written:
                    t = args[i] if i < len(args) else None
                    if isinstance(t, Tensor):
                        try:
                            torch.autograd.graph.increment_version(t)
                        except RuntimeError:  # (an inference tensor: no version counter to keep)
                            pass
                return fn(*args, **kwargs)
        if _libdef is not None:
            _libdef.define(schema)
            _libdef.impl(name, impl, "CompositeExplicitAutograd")
        elif not hasattr(getattr(torch.ops, NAMESPACE), name):
            raise _lib.HetError(f"{COMPILED_LIB} does not register torch_hrt.{name}")
        _registered.append(name)
        return fn

    return deco


# Optional per-entry-point device timing (bench.py): name -> list of (start, end, args) with events recorded on
# the stream the kernels are launched on.  Empty = no overhead.
event_timers: Dict[str, list] = {}


def _call(dev_tensor: Tensor, cname: str, *args):
    if not event_timers and dev_tensor.device.index == torch.cuda.current_device():
        _lib.call(cname, *args)  # the common case: the tensor's device is current and nobody times the entry points
        return
    with torch.cuda.device(dev_tensor.device):
        rec = event_timers.get(cname)
        if rec is None:
            rec = event_timers.get("*")  # wildcard: time every entry point (bench.py's per-op breakdown)
            if rec is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _lib.call(cname, *args)
                b.record()
                rec.append((a, b, cname))
                return
        if rec is None:
            _lib.call(cname, *args)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(cname, *args)
        b.record()
        rec.append((a, b, args))


# ------------------------------------------------------------------------------------
# info + layout converters
# ------------------------------------------------------------------------------------
@_op("build_debug_info() -> ()")
def build_debug_info():
    print(_lib.build_info())


@_op("transpose_csr(Tensor row_ptrs, Tensor col_indices, Tensor eids, Tensor rel_types) -> Tensor[]")
def transpose_csr(row_ptrs, col_indices, eids, rel_types):
    # DataConverters.inc.h:283-344: returns (row_ptrs, col_indices, eids, rel_types) of the transpose
    return list(_graph.transpose_csr(row_ptrs, col_indices, eids, rel_types))


@_op("convert_integrated_coo_to_separate_coo(Tensor row_indices, Tensor col_indices, Tensor rel_types, Tensor eids, "
     "int num_nodes, int num_rels) -> Tensor[]")
def convert_integrated_coo_to_separate_coo(row_indices, col_indices, rel_types, eids, num_nodes, num_rels):
    # DataConverters.inc.h:216-281 -> MyHyb.h:1047-1096 (bucket by relation; buckets kept in eid order)
    return list(_graph.integrated_coo_to_separate_coo(row_indices, col_indices, rel_types, eids, num_rels))


@_op("convert_integrated_csr_to_separate_coo(Tensor row_ptrs, Tensor col_indices, Tensor rel_types, Tensor eids) -> Tensor[]")
def convert_integrated_csr_to_separate_coo(row_ptrs, col_indices, rel_types, eids):
    # DataConverters.inc.h:10-77 -> MyHyb.h:1099-1150
    rows = _graph.csr_to_coo_rows(row_ptrs)
    num_rels = int(rel_types.max().item()) + 1 if rel_types.numel() else 1
    return list(_graph.integrated_coo_to_separate_coo(rows, col_indices, rel_types, eids, num_rels))


def _separate_csr(rows, cols, rels, eids, num_rows, num_rels):
    # a CSR over the composite row (relation, row): stable, so a row keeps its edges in input order
    row_ptrs, c, _, e = _graph.coo_to_csr(rels * num_rows + rows, cols, rels, eids, num_rows * num_rels)
    rel_ptrs = row_ptrs[::num_rows].contiguous()
    return [rel_ptrs, row_ptrs, c, e]


@_op("convert_integrated_csr_to_separate_csr(Tensor row_ptrs, Tensor col_indices, Tensor rel_types, Tensor eids) -> Tensor[]")
def convert_integrated_csr_to_separate_csr(row_ptrs, col_indices, rel_types, eids):
    # DataConverters.inc.h:79-145: one CSR per relation, row pointers concatenated ([R*N+1])
    rows = _graph.csr_to_coo_rows(row_ptrs)
    num_rels = int(rel_types.max().item()) + 1 if rel_types.numel() else 1
    return _separate_csr(rows, col_indices, rel_types, eids, row_ptrs.numel() - 1, num_rels)


@_op("convert_integrated_coo_to_separate_csr(Tensor row_indices, Tensor col_indices, Tensor rel_types, Tensor eids, "
     "int num_nodes, int num_rels) -> Tensor[]")
def convert_integrated_coo_to_separate_csr(row_indices, col_indices, rel_types, eids, num_nodes, num_rels):
    # DataConverters.inc.h:147-214
    return _separate_csr(row_indices, col_indices, rel_types, eids, num_nodes, num_rels)


# ------------------------------------------------------------------------------------
# segment GEMM ops
# ------------------------------------------------------------------------------------
def _matmul_lists(d: Dict[str, Tensor], kind: int):
    if kind == 0:
        return d["separate_coo_rel_ptrs"], d["separate_coo_node_indices"], d["separate_coo_eids"]
    if kind == 1:
        return d["unique_srcs_and_dests_rel_ptrs"], d["unique_srcs_and_dests_node_indices"], None
    raise _lib.HetError(f"rgnn_relational_matmul: CompactAsOfNodeKind {kind} not supported "
                        "(the reference asserts, RGNNOps.inc.h:292-294)")


@_op("rgnn_relational_matmul(Dict(str, Tensor) args_tensor_dict, int IntKind, Tensor weights, Tensor node_feat, "
     "Tensor(a!) ret, bool InputNumHeadOneFlag) -> ()")
def rgnn_relational_matmul(args_tensor_dict, IntKind, weights, node_feat, ret, InputNumHeadOneFlag):
    rp, g, s = _matmul_lists(args_tensor_dict, IntKind)
    _chk("rgnn_relational_matmul", (weights, node_feat, ret), tuple(t for t in (rp, g, s) if t is not None))
    R, H, K, D = weights.shape
    grp = ws = None
    X = H * D
    mfma = K in (32, 64, 128) and X in (32, 64, 128)
    # distinct (relation, node) rows + broadcast: the row-dot shape (D == 1), and projections the matrix-core kernel does
    # not take (e.g. an 8-wide output layer)
    if (IntKind == 0 and InputNumHeadOneFlag and ((D == 1 and H & (H - 1) == 0) or (D > 1 and not mfma and X & (X - 1) == 0 and X <= 256))
            and _plan.is_enabled() and ret.is_cuda and g.numel() > 0 and g.data_ptr() != s.data_ptr()):
        grp = _plan.get_grouping(rp, g, node_feat.shape[0], s, None)  # the grouping the backward uses as well
        if grp is not None:
            ws = torch.empty(max(1, grp.num_segments) * X, dtype=torch.float32, device=ret.device)
    _call(ret, "het_rgnn_relational_matmul", IntKind, _p(rp), R, _p(g), _p(s), g.numel(), _p(weights), _p(node_feat),
          _p(ret), H, K, D, int(InputNumHeadOneFlag), None if grp is None else grp.handle, _p(ws),
          0 if ws is None else ws.numel() * 4, _stream(ret))


def matmul_attn_dot_ok(H: int, K: int, D: int) -> bool:
    """Shapes het_rgnn_relational_matmul_attn_dot covers (the MFMA forward with the dot epilogue)."""
    return K in (32, 64, 128) and H * D in (32, 64, 128) and D >= 4 and D & (D - 1) == 0


def matmul_attn_dot_only_ok(args_tensor_dict, weights, node_feat) -> bool:
    """Whether the attention term can be formed without materialising the per-edge projection (kind 0 lists)."""
    R, H, K, D = weights.shape
    rp, g, s = _matmul_lists(args_tensor_dict, 0)
    return (_plan.is_enabled() and node_feat.is_cuda and matmul_attn_dot_ok(H, K, D) and H & (H - 1) == 0 and H <= H * D // 4
            and g.numel() > 0 and g.data_ptr() != s.data_ptr())


def matmul_attn_dot(args_tensor_dict, IntKind, weights, node_feat, ret, dot_w, dot_out, keep_rows=False, dot_rows=None):
    """rgnn_relational_matmul (one input head) that also writes dot_out[row, h] = <ret[row, h, :], dot_w[r, h, :]>.
    ret None: only dot_out (needs matmul_attn_dot_only_ok).  keep_rows: also return the [S, H, D] distinct projected
    rows of the (relation, node) grouping (None when that path is not taken).  dot_rows [E] int64: the dot of
    position i goes to dot_out[dot_rows[i]] instead of its separate_coo_eids row (grouping path only)."""
    rp, g, s = _matmul_lists(args_tensor_dict, IntKind)
    _chk("rgnn_relational_matmul_attn_dot", tuple(t for t in (weights, node_feat, ret, dot_w, dot_out) if t is not None),
         tuple(t for t in (rp, g, s) if t is not None))
    R, H, K, D = weights.shape
    grp = ws = comp = None
    if IntKind == 0 and _plan.is_enabled() and g.numel() > 0 and g.data_ptr() != s.data_ptr():
        grp = _plan.get_grouping(rp, g, node_feat.shape[0], s, None)  # the grouping the backward uses as well
        if grp is not None:
            S = max(1, grp.num_segments)
            if keep_rows or ret is None:
                comp = torch.empty((S, H, D), dtype=torch.float32, device=dot_out.device)
            ws = torch.empty(S * ((0 if comp is not None else H * D) + H), dtype=torch.float32, device=dot_out.device)
    if (ret is None or dot_rows is not None) and grp is None:
        raise _lib.HetError("rgnn_relational_matmul_attn_dot: ret=None / dot_rows need the (relation, node) grouping")
    dgrp = None if dot_rows is None else _plan.get_grouping(rp, g, node_feat.shape[0], dot_rows, None)
    _call(dot_out, "het_rgnn_relational_matmul_attn_dot", IntKind, _p(rp), R, _p(g), _p(s), g.numel(), _p(weights), _p(node_feat),
          _p(ret), _p(dot_w), _p(dot_out), H, K, D, None if grp is None else grp.handle, _p(ws),
          0 if ws is None else ws.numel() * 4, _p(comp), None if dgrp is None else dgrp.handle, _stream(dot_out))
    return comp


def matmul_attn_dot_rows(rel_ptrs, gather_idx, scatter_idx, weights, node_feat, ret, dot_w, dot_out):
    """ret[scatter_idx[i]] = node_feat[gather_idx[i]] . W[r(i)] and dot_out[scatter_idx[i], h] = <that row's head h, dot_w[r, h]> for
    a relation-bucketed list of DISTINCT rows -- no grouping is looked up or built (a piece of a unique (relation, node) list:
    the projection of the halo rows that have arrived, het_amd/backend/rgat_fused_layer.py)."""
    _chk("rgnn_relational_matmul_attn_dot", (weights, node_feat, ret, dot_w, dot_out), (rel_ptrs, gather_idx, scatter_idx))
    R, H, K, D = weights.shape
    if gather_idx.numel() == 0:
        return
    _call(dot_out, "het_rgnn_relational_matmul_attn_dot", 0, _p(rel_ptrs), R, _p(gather_idx), _p(scatter_idx), gather_idx.numel(),
          _p(weights), _p(node_feat), _p(ret), _p(dot_w), _p(dot_out), H, K, D, None, None, 0, None, None, _stream(dot_out))


def rows_add_bias(a, b=None, bias=None):
    """out = a (+ b) (+ bias broadcast over rows) in one pass (include/het_amd.h: het_rows_add_bias)."""
    _chk("rows_add_bias", tuple(t for t in (a, b, bias) if t is not None))
    out = torch.empty_like(a)
    _call(a, "het_rows_add_bias", _p(a), _p(b), _p(bias), _p(out), a.shape[0], a.numel() // max(1, a.shape[0]), _stream(a))
    return out


def rows_gather(x, idx):
    """out[i, :] = x[idx[i], :] (include/het_amd.h: het_rows_gather); torch.index_select off the GPU or for odd widths."""
    if not (x.is_cuda and x.dim() == 2 and x.shape[1] % 4 == 0 and x.dtype == torch.float32 and x.is_contiguous()):
        return x.index_select(0, idx)
    _chk("rows_gather", (x,), (idx,))
    out = torch.empty((idx.numel(), x.shape[1]), dtype=x.dtype, device=x.device)
    _call(x, "het_rows_gather", _p(x), _p(idx), idx.numel(), x.shape[1], _p(out), _stream(x))
    return out


def rows_scatter_add_(out, idx, src):
    """out[idx[i], :] += src[i, :] in place (het_rows_scatter_add); Tensor.index_add_ off the GPU."""
    if not (out.is_cuda and out.dim() == 2 and out.dtype == torch.float32 and out.is_contiguous() and src.is_contiguous()):
        return out.index_add_(0, idx, src)
    _chk("rows_scatter_add", (out, src), (idx,))
    X = out.shape[1]
    if _plan.is_enabled() and idx.numel() > 0 and 4 <= X <= 256 and X & (X - 1) == 0:
        # idx is a per-plan list (the halo send list): grouped by destination row once, then summed without atomics
        pos = _derived_get("positions", (idx,), lambda: torch.arange(idx.numel(), dtype=torch.int64, device=idx.device))
        g = _plan.get_grouping(None, idx, out.shape[0], pos, None)
        _call(out, "het_rows_scatter_add_grouped", g.handle, _p(src), X, _p(out), out.shape[0], _stream(out))
        return out
    _call(out, "het_rows_scatter_add", _p(src), _p(idx), idx.numel(), X, _p(out), _stream(out))
    return out


def matmul_attn_dot_only_backward(args_tensor_dict, weights_transposed, node_feat, dot_w, grad_dot, grad_node_feat, grad_weights,
                                  comp_rows=None, grad_dot_w=None, accumulate=False):
    """Backward of matmul_attn_dot when only dot_out was used (see include/het_amd.h); returns False when the fast
    path does not apply (no grouping / shape), leaving the outputs untouched."""
    rp, g, s = _matmul_lists(args_tensor_dict, 0)
    R, H, D, K = weights_transposed.shape
    if not (_plan.is_enabled() and H & (H - 1) == 0 and D % 4 == 0 and g.numel() > 0):
        return False
    grp = _plan.get_grouping(rp, g, node_feat.shape[0], s, None)
    if grp is None:
        return False
    _chk("backward_rgnn_relational_matmul_attn_dot_only", (weights_transposed, node_feat, dot_w, grad_dot, grad_node_feat, grad_weights), (rp, g, s))
    S = max(1, grp.num_segments)
    ws = torch.empty(((S * H + 3) // 4) * 4 + S * H * D, dtype=torch.float32, device=grad_dot.device)
    _call(grad_dot, "het_backward_rgnn_relational_matmul_attn_dot_only", _p(rp), R, _p(g), _p(s), g.numel(), node_feat.shape[0],
          _p(weights_transposed), _p(node_feat), _p(dot_w), _p(grad_dot), _p(grad_node_feat), _p(grad_weights), H, K, D,
          int(accumulate), grp.handle, _p(ws), ws.numel() * 4, _p(comp_rows), _p(grad_dot_w), _stream(grad_dot))
    return True


@_op("backward_rgnn_relational_matmul(Dict(str, Tensor) args_tensor_dict, int IntKind, Tensor weights_transposed, "
     "Tensor node_feat, Tensor gradout, Tensor(a!) grad_node_feat, Tensor(b!) grad_weights, bool InputNumHeadOneFlag) -> ()")
def backward_rgnn_relational_matmul(args_tensor_dict, IntKind, weights_transposed, node_feat, gradout, grad_node_feat,
                                    grad_weights, InputNumHeadOneFlag):
    matmul_backward(args_tensor_dict, IntKind, weights_transposed, node_feat, gradout, grad_node_feat, grad_weights,
                    InputNumHeadOneFlag, accumulate=True)


def matmul_backward(args_tensor_dict, IntKind, weights_transposed, node_feat, gradout, grad_node_feat, grad_weights,
                    InputNumHeadOneFlag, accumulate: bool, distinct_rows: bool = False):
    """backward_rgnn_relational_matmul with the choice of "+=" (the reference op's contract) or "=" outputs.
    distinct_rows (kind 1): the caller guarantees that the unique list holds a node at most once per relation -- the
    graph's own unique (relation, node) lists -- which lets the input gradients be added relation by relation with plain
    read-modify-write.  The reference-named op never sets it: any list, duplicates included, is then summed with float
    atomics as in the reference (include/het_amd.h, a2)."""
    rp, g, s = _matmul_lists(args_tensor_dict, IntKind)
    _chk("backward_rgnn_relational_matmul",
         tuple(t for t in (weights_transposed, node_feat, gradout, grad_node_feat, grad_weights) if t is not None),
         tuple(t for t in (rp, g, s) if t is not None))
    R, H, D, K = weights_transposed.shape
    grp, ws = None, None
    if IntKind == 0 and g.data_ptr() != s.data_ptr() and (InputNumHeadOneFlag or D > 1):
        grp = _plan.get_grouping(rp, g, node_feat.shape[0], s, None)
        if grp is not None:
            ws = torch.empty(max(1, grp.num_segments) * H * D, dtype=torch.float32, device=gradout.device)
    _call(gradout, "het_backward_rgnn_relational_matmul", IntKind, _p(rp), R, _p(g), _p(s), g.numel(),
          node_feat.shape[0], _p(weights_transposed), _p(node_feat), _p(gradout), _p(grad_node_feat), _p(grad_weights),
          H, K, D, int(InputNumHeadOneFlag), int(bool(accumulate)) | (2 if distinct_rows and IntKind == 1 else 0),
          None if grp is None else grp.handle, _p(ws),
          0 if ws is None else ws.numel() * 4, _stream(gradout))


@_op("rgnn_relational_matmul_no_scatter_gather_list(Tensor ntype_offset_ptrs, Tensor weights, Tensor inputs, "
     "Tensor(a!) ret) -> ()")
def rgnn_relational_matmul_no_scatter_gather_list(ntype_offset_ptrs, weights, inputs, ret):
    _chk("rgnn_relational_matmul_no_scatter_gather_list", (weights, inputs, ret), (ntype_offset_ptrs,))
    T, H, K, D = weights.shape
    n = inputs.shape[0]
    per_head = int(H > 1 and inputs.numel() == n * H * K)
    _call(ret, "het_rgnn_relational_matmul_no_scatter_gather_list", _p(ntype_offset_ptrs), T, n, _p(weights),
          _p(inputs), _p(ret), H, K, D, per_head, _stream(ret))


@_op("backward_rgnn_relational_matmul_no_scatter_gather_list(Tensor ntype_offset_ptrs, Tensor weights_transposed, "
     "Tensor inputs, Tensor gradout, Tensor(a!) grad_input, Tensor(b!) grad_weights) -> ()")
def backward_rgnn_relational_matmul_no_scatter_gather_list(ntype_offset_ptrs, weights_transposed, inputs, gradout,
                                                           grad_input, grad_weights):
    matmul_no_scatter_gather_backward(ntype_offset_ptrs, weights_transposed, inputs, gradout, grad_input, grad_weights,
                                      accumulate=True)


def matmul_no_scatter_gather_backward(ntype_offset_ptrs, weights_transposed, inputs, gradout, grad_input, grad_weights,
                                      accumulate: bool):
    _chk("backward_rgnn_relational_matmul_no_scatter_gather_list",
         tuple(t for t in (weights_transposed, inputs, gradout, grad_input, grad_weights) if t is not None), (ntype_offset_ptrs,))
    T, H, D, K = weights_transposed.shape
    n = inputs.shape[0]
    per_head = int(H > 1 and inputs.numel() == n * H * K)
    _call(gradout, "het_backward_rgnn_relational_matmul_no_scatter_gather_list", _p(ntype_offset_ptrs), T, n,
          _p(weights_transposed), _p(inputs), _p(gradout), _p(grad_input), _p(grad_weights), H, K, D, per_head,
          int(accumulate), _stream(gradout))


# ------------------------------------------------------------------------------------
# fused GAT (edge softmax + aggregation)
# ------------------------------------------------------------------------------------
def _gat_maps(kind: int, d: Dict[str, Tensor], backward: bool):
    """Flatten the op's dict into (row_a, row_b, col_a, col_b); key names as the launchers read them
    (RGATOps.inc.h:180-236 forward, :476-540 backward -- the backward spells the dual-list column
    pointer key 'unique_srcs_and_dests_rel_col')."""
    if kind == 0:
        return None, None, None, None
    if kind == 1:
        rp, nodes = d["unique_srcs_and_dests_rel_ptrs"], d["unique_srcs_and_dests_node_indices"]
        return rp, nodes, rp, nodes
    if kind == 3:
        rpc = d.get("unique_srcs_and_dests_rel_ptrs_col")
        if rpc is None:
            rpc = d["unique_srcs_and_dests_rel_col"]
        return (d["unique_srcs_and_dests_rel_ptrs"], d["unique_srcs_and_dests_node_indices_row"], rpc,
                d["unique_srcs_and_dests_node_indices_col"])
    if kind == 4:
        return d["edata_idx_to_inverse_idx_row"], None, d["edata_idx_to_inverse_idx_col"], None
    if kind == 2:  # one inverse index for both edge ends, as the reference's non-dual direct-indexing branch reads it
        m = d["edata_idx_to_inverse_idx"]
        return m, None, m, None
    raise _lib.HetError(f"relational_fused_gat: CompactAsOfNodeKind {kind} is not supported")


_derived: "OrderedDict[tuple, tuple]" = OrderedDict()
_DERIVED_MAX = 64
import threading as _threading
_derived_lock = _threading.RLock()


def _derived_get(tag, tensors, build):
    """Per-graph derived index tensors (row maps, relation of every position ...), built once and looked up by the
    identity of ALL the tensors they were derived from: (data_ptr, numel, _version) each, as het_amd/plan.py keys its
    groupings -- an in-place edit of any of them (a renumbering of the eids, a reordered block) misses the cache instead of
    returning stale rows.  The entry keeps the source tensors alive, which pins their storage (a data_ptr cannot be
    recycled for other contents while the entry lives).  Least recently used entries go first."""
    key = (tag,) + tuple(_plan._ident(t) for t in tensors)
    with _derived_lock:  # (model threads and autograd workers share the cache: lookup, build and eviction under one lock)
        hit = _derived.get(key)
        if hit is None:
            hit = (build(), tensors)
            _derived[key] = hit
            while len(_derived) > _DERIVED_MAX:
                _derived.popitem(last=False)
        else:
            _derived.move_to_end(key)
        return hit[0]


def _rows_by_search(rel_ptrs, nodes, ua, ub):
    """Row of every (relation of the position, node) pair in the unique list (ua = its relation pointers, ub = node ids)."""
    R = rel_ptrs.numel() - 1
    bound = int(max(int(nodes.max().item()), int(ub.max().item()))) + 1
    rel_e = torch.repeat_interleave(torch.arange(R, device=nodes.device), rel_ptrs[1:] - rel_ptrs[:-1])
    rel_u = torch.repeat_interleave(torch.arange(R, device=nodes.device), ua[1:] - ua[:-1])
    return torch.searchsorted(rel_u * bound + ub, rel_e * bound + nodes).contiguous()


def _gat_direct(kind, maps, rel_ptrs, row, col, eids):
    """(kind, maps) as handed to the C entry points: the binary-search kinds 1 / 3 are turned into the direct-index
    kind 4 once per graph (feat / er row of every edge id, cached) -- the maps the kernels read without searching."""
    if kind == 2:
        return 4, maps  # (maps[2] is maps[0]: see _gat_maps)
    if kind not in (1, 3) or not _plan.is_enabled() or eids.numel() == 0:
        return kind, maps

    def build():
        srow = _src_rows_by_position(kind, maps, rel_ptrs, row, eids)
        drow = _dst_rows_by_position(kind, maps, rel_ptrs, col, eids)
        n = int(eids.max().item()) + 1
        mr = torch.empty(n, dtype=torch.int64, device=eids.device)
        mc = torch.empty(n, dtype=torch.int64, device=eids.device)
        mr[eids] = srow
        mc[eids] = drow
        return (mr, None, mc, None)

    return 4, _derived_get(("gatmap", kind), (maps[0], maps[1], maps[2], maps[3], rel_ptrs, row, col, eids), build)


def _src_rows_by_position(kind, maps, rel_ptrs, row, eids):
    """feat row of every edge position for the compact kinds (cached per graph)."""
    ra, rb = maps[0], maps[1]
    if kind == 4:
        return _derived_get("srow4", (ra, eids), lambda: ra[eids].contiguous())
    return _derived_get("srow", (ra, rb, rel_ptrs, row), lambda: _rows_by_search(rel_ptrs, row, ra, rb))


def _dst_rows_by_position(kind, maps, rel_ptrs, col, eids):
    """er row of every edge position for the compact kinds (cached per graph)."""
    ca, cb = maps[2], maps[3]
    if kind == 4:
        return _derived_get("drow4", (ca, eids), lambda: ca[eids].contiguous())
    return _derived_get("drow", (ca, cb, rel_ptrs, col), lambda: _rows_by_search(rel_ptrs, col, ca, cb))


def _csr_compact_maps(row_ptrs, col, eids, reltypes, uniq_rel_ptrs, uniq_nodes, rows_are_dst: bool):
    """CompactAsOfNodeFlag of the CSR GAT pair (RGATOps.inc.h:251-277, 430-460): feat / el / er live on the rows of ONE
    unique (relation, node) list and every edge end is looked up by (relation of the edge, node).  Turned once per graph
    (cached) into the direct-index maps of kind 4 -- feat row and er row of every EDGE ID -- which the destination-grouped
    kernels read without searching: (rows of the CSR expanded per position, {edata idx -> row} dict)."""
    def build():
        E = eids.numel()
        rows = _graph.csr_to_coo_rows(row_ptrs).contiguous()
        src, dst = (col, rows) if rows_are_dst else (rows, col)
        R = uniq_rel_ptrs.numel() - 1
        bound = int(max(int(row_ptrs.numel()) - 1, int(uniq_nodes.max().item()) + 1 if uniq_nodes.numel() else 1))
        rel_u = torch.repeat_interleave(torch.arange(R, device=col.device), uniq_rel_ptrs[1:] - uniq_rel_ptrs[:-1])
        keys = rel_u * bound + uniq_nodes
        n = int(eids.max().item()) + 1 if E else 0
        mr = torch.empty(n, dtype=torch.int64, device=col.device)
        mc = torch.empty(n, dtype=torch.int64, device=col.device)
        mr[eids] = torch.searchsorted(keys, reltypes * bound + src)
        mc[eids] = torch.searchsorted(keys, reltypes * bound + dst)
        return rows, {"edata_idx_to_inverse_idx_row": mr, "edata_idx_to_inverse_idx_col": mc}
    return _derived_get(("csrcompact", rows_are_dst), (row_ptrs, col, eids, reltypes, uniq_rel_ptrs, uniq_nodes), build)


def _csr_expanded_rows(row_ptrs, num_edges):
    """(row id of every CSR position, a one-relation rel_ptrs [0, E]) -- cached per CSR."""
    return _derived_get(("csr", num_edges), (row_ptrs,),
                        lambda: (_graph.csr_to_coo_rows(row_ptrs).contiguous(),
                                 torch.tensor([0, num_edges], dtype=torch.int64, device=row_ptrs.device)))


def _rel_by_position(rel_ptrs, num_positions):
    """Relation of every edge position of the separate COO (cached per graph)."""
    R = rel_ptrs.numel() - 1
    return _derived_get(("rel", num_positions), (rel_ptrs,),
                        lambda: torch.repeat_interleave(torch.arange(R, device=rel_ptrs.device), rel_ptrs[1:] - rel_ptrs[:-1],
                                                        output_size=num_positions).contiguous())


def _by_dst(kind, maps, rel_ptrs, row, col, eids, num_nodes):
    """Positions grouped by destination; payload0 = edge id, payload1 = feat row (compact kinds) or the relation
    of the position (kind 0; read by the fold_attn_l backward)."""
    if not _plan.is_enabled():
        return None
    p1 = _rel_by_position(rel_ptrs, eids.numel()) if kind == 0 else _src_rows_by_position(kind, maps, rel_ptrs, row, eids)
    return _plan.get_grouping(None, col, num_nodes, eids, p1)


@_op("relational_fused_gat_separate_coo(Tensor separate_coo_eids, Tensor separate_coo_rel_ptrs, "
     "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, int IntKind, "
     "Dict(str, Tensor) args_tensor_dict, Tensor feat_src, Tensor el, Tensor er, Tensor(a!) sum, Tensor(b!) exp, "
     "Tensor(c!) ret, float slope) -> ()")
def relational_fused_gat_separate_coo(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices,
                                      separate_coo_col_indices, IntKind, args_tensor_dict, feat_src, el, er, sum, exp,
                                      ret, slope):
    # kind 0 on the destination-grouped kernels: the pass also leaves a copy of exp in the grouping's order (sequential 16-byte
    # stores: +338 MB written on ogbn-mag), remembered against the identity of (exp, el, er) -- the backward op streams it
    # when the same tensors come back (_sorted_stream_*)
    exs = None
    if (A5_SORTED_STREAM and IntKind == 0 and slope >= 0 and _plan.is_enabled() and separate_coo_eids.numel() > 0 and exp.dim() >= 2
            and gat_grouped_shape_ok(sum.shape[1], ret.numel() // max(1, ret.shape[0] * sum.shape[1]))):
        exs = torch.empty((separate_coo_eids.numel(), sum.shape[1]), dtype=exp.dtype, device=exp.device)
    wrote = fused_gat_forward(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                              IntKind, args_tensor_dict, feat_src, el, er, sum, exp, ret, slope, exs)
    _sorted_stream_put(exp, el, er, separate_coo_eids, separate_coo_col_indices, float(slope), exs if wrote else None)


def gat_grouped_shape_ok(H: int, D: int) -> bool:
    """Shapes the destination-grouped GAT kernels cover (fused_gat_grouped.hip: grouped_shape_ok)."""
    X = H * D
    return D >= 4 and D & (D - 1) == 0 and X & (X - 1) == 0 and X // 4 <= 64


def fused_gat_forward(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                      IntKind, args_tensor_dict, feat_src, el, er, sum, exp, ret, slope, exp_sorted, el_sorted=None,
                      er_sorted=None):
    """relational_fused_gat_separate_coo plus the optional ``exp_sorted`` output ([E,H], exp in
    destination-grouped order) that lets the backward stream instead of gather.  el_sorted / er_sorted (kind 0): the
    attention terms already in that order (gat_rank_of_position); el, er, exp may then be None."""
    name = "relational_fused_gat_separate_coo"
    maps = _gat_maps(IntKind, args_tensor_dict, False)
    _chk(name, tuple(t for t in (feat_src, el, er, sum, exp, ret, el_sorted, er_sorted) if t is not None),
         (separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices)
         + tuple(m for m in maps if m is not None))
    IntKind, maps = _gat_direct(IntKind, maps, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                                separate_coo_eids)
    E, N, H = separate_coo_eids.numel(), ret.shape[0], sum.shape[1]
    D = feat_src.numel() // (feat_src.shape[0] * H) if feat_src.numel() else ret.numel() // max(1, N * H)
    g = _by_dst(IntKind, maps, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                separate_coo_eids, N)
    _call(ret, "het_relational_fused_gat_separate_coo", _p(separate_coo_eids), _p(separate_coo_rel_ptrs),
          _p(separate_coo_row_indices), _p(separate_coo_col_indices), separate_coo_rel_ptrs.numel() - 1, E, N, IntKind,
          _p(maps[0]), _p(maps[1]), _p(maps[2]), _p(maps[3]), _p(feat_src), _p(el), _p(er), _p(sum), _p(exp), _p(ret),
          None if g is None else _p(exp_sorted), H, D, float(slope), None if g is None else g.handle, _p(el_sorted),
          _p(er_sorted), _stream(ret))
    return g is not None and exp_sorted is not None



# ---- the op-level a5 streams the sorted copy of exp its a4 left --------------------------------------------------------
# Called on their own (the reference's model code: RelationalFusedGatSeparateCOO, rgat_layers_and_funcs.py:262-330), a5 gathers
# exp / el / er by edge id in destination order: a 128-byte line per 16-byte record (0.41 of the HBM roofline, 21 GB of traffic
# for 13.9 GB of tensors).  The layer-private path avoids that with a destination-sorted copy of exp written by the forward;
# here the same copy is kept BETWEEN the two reference-named calls, keyed by the identity of the tensors the backward would
# otherwise read: (data_ptr, _version) of exp, el, er + the edge lists.  A weak reference per tensor makes a recycled address a
# miss (a dead tensor's storage may be handed out again); an in-place edit bumps _version: also a miss.  A miss = the gathers.
import weakref as _weakref
A5_SORTED_STREAM = _os.environ.get("HET_A5_SORTED_STREAM", "1") != "0"
_sorted_streams: "OrderedDict[tuple, tuple]" = OrderedDict()
_SORTED_STREAMS_MAX = 2  # (a [E,H] float tensor each: 338 MB on ogbn-mag)
sorted_stream_hits = 0   # (tests / bench: how often the backward op found its stream)


def _sorted_stream_key(exp, el, er, eids, col, slope):
    return (_plan._ident(exp)[:2], _plan._ident(el), _plan._ident(er), _plan._ident(eids), _plan._ident(col), slope, exp.device.index)


def _sorted_stream_put(exp, el, er, eids, col, slope, exs):
    key = _sorted_stream_key(exp, el, er, eids, col, slope)
    with _derived_lock:
        _sorted_streams.pop(key, None)
        if exs is not None:
            # exp's version is taken AFTER the op wrote it (the C side writes through the raw pointer: no bump of its own)
            _sorted_streams[key] = (exs, exp._version, tuple(_weakref.ref(t) for t in (exp, el, er)))
            while len(_sorted_streams) > _SORTED_STREAMS_MAX:
                _sorted_streams.popitem(last=False)


def _sorted_stream_get(exp, el, er, eids, col, slope):
    global sorted_stream_hits
    if not (A5_SORTED_STREAM and _plan.is_enabled()) or exp is None or el is None or er is None:
        return None
    with _derived_lock:
        hit = _sorted_streams.get(_sorted_stream_key(exp, el, er, eids, col, slope))
        if hit is None:
            return None
        exs, version, refs = hit
        if exp._version != version or any(r() is None for r in refs):
            return None
        sorted_stream_hits += 1
        return exs


@_op("backward_relational_fused_gat_separate_coo(Tensor separate_coo_eids, Tensor separate_coo_rel_ptrs, "
     "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, int IntKind, "
     "Dict(str, Tensor) args_tensor_dict, Tensor feat_src, Tensor el, Tensor er, Tensor sum, Tensor exp, Tensor ret, "
     "Tensor gradout, Tensor(a!) grad_feat_src, Tensor(b!) grad_el, Tensor(c!) grad_er, float slope) -> ()")
def backward_relational_fused_gat_separate_coo(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices,
                                               separate_coo_col_indices, IntKind, args_tensor_dict, feat_src, el, er,
                                               sum, exp, ret, gradout, grad_feat_src, grad_el, grad_er, slope):
    exs = _sorted_stream_get(exp, el, er, separate_coo_eids, separate_coo_col_indices, float(slope)) if IntKind == 0 else None
    fused_gat_backward(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                       IntKind, args_tensor_dict, feat_src, el, er, sum, exp, ret, gradout, grad_feat_src, grad_el,
                       grad_er, slope, exs)


def fused_gat_backward(separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                       IntKind, args_tensor_dict, feat_src, el, er, sum, exp, ret, gradout, grad_feat_src, grad_el,
                       grad_er, slope, exp_sorted, fold_attn_l=None, grad_fold_attn_l=None, fold_row_rel_ptrs=None,
                       grad_el_sorted=None):
    """fold_attn_l [R,H,D]: also add grad_el[e,h] * fold_attn_l[r,h,:] into grad_feat_src (see include/het_amd.h).
    grad_el_sorted [E,H] (kind 0): grad_el in destination-grouped order (gat_rank_of_position); grad_el / grad_er
    may then be None."""
    name = "backward_relational_fused_gat_separate_coo"
    maps = _gat_maps(IntKind, args_tensor_dict, True)
    _chk(name, tuple(t for t in (feat_src, el, er, sum, exp, ret, gradout, grad_feat_src, grad_el, grad_er, grad_el_sorted,
                                 exp_sorted) if t is not None),
         (separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices)
         + tuple(m for m in maps if m is not None))
    IntKind, maps = _gat_direct(IntKind, maps, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                                separate_coo_eids)
    E, N, H = separate_coo_eids.numel(), ret.shape[0], sum.shape[1]
    D = ret.numel() // max(1, N * H)
    g = _by_dst(0, maps, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices,
                separate_coo_eids, N) if IntKind == 0 else None
    gs = gd = ws = None
    if IntKind != 0 and _plan.is_enabled() and slope >= 0 and gat_grouped_shape_ok(H, D) and E > 0:
        srow = _src_rows_by_position(IntKind, maps, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_eids)
        drow = _dst_rows_by_position(IntKind, maps, separate_coo_rel_ptrs, separate_coo_col_indices, separate_coo_eids)
        gs = _plan.get_grouping(None, srow, feat_src.shape[0], separate_coo_eids, separate_coo_col_indices)
        gd = _plan.get_grouping(None, drow, er.shape[0], separate_coo_eids, None)
        ws = torch.empty(N * 2 * H + E * H, dtype=torch.float32, device=ret.device)
    if IntKind == 0 and g is not None and grad_fold_attn_l is not None:
        ws = torch.empty(64 * grad_fold_attn_l.numel(), dtype=torch.float32, device=ret.device)  # replicas of the weight gradient
    _call(ret, "het_backward_relational_fused_gat_separate_coo", _p(separate_coo_eids), _p(separate_coo_rel_ptrs),
          _p(separate_coo_row_indices), _p(separate_coo_col_indices), separate_coo_rel_ptrs.numel() - 1, E, N, IntKind,
          _p(maps[0]), _p(maps[1]), _p(maps[2]), _p(maps[3]), _p(feat_src), _p(el), _p(er), _p(sum), _p(exp), _p(ret),
          None if g is None else _p(exp_sorted), _p(gradout), _p(grad_feat_src), _p(grad_el), _p(grad_er), H, D,
          float(slope), None if g is None else g.handle, None if gs is None else gs.handle,
          None if gd is None else gd.handle, feat_src.shape[0], E if er is None else er.shape[0], _p(ws), 0 if ws is None else ws.numel() * 4,
          None if fold_attn_l is None else _p(fold_attn_l), None if grad_fold_attn_l is None else _p(grad_fold_attn_l),
          None if fold_row_rel_ptrs is None else _p(fold_row_rel_ptrs), _p(grad_el_sorted), _stream(ret))


def grouping_rank_of_position(g):
    """[E] int64: sorted rank of every position of grouping ``g`` (cached on the grouping object, built at plan time)."""
    r = getattr(g, "rank_of_position", None)
    if r is None:
        keys = g._keep[1]
        r = torch.empty(keys.numel(), dtype=torch.int64, device=keys.device)
        _call(r, "het_grouping_rank_of_position", g.handle, _p(r), _stream(r))
        g.rank_of_position = r  # lives (and is evicted) with the grouping
    return r


def rgat_runs_shape_ok(H: int, D: int) -> bool:
    """Shapes of the run-sum form of the compact RGAT pair (include/het_amd.h: het_rgat_aggregate_compact_runs): rows of 32 / 64 /
    128 floats, heads of at least 16."""
    X = H * D
    return X in (32, 64, 128) and D >= 16 and D & (D - 1) == 0 and X % D == 0


def rgat_compact_groupings(col, srow, drow, num_nodes, num_src_rows, num_dst_rows, rel_ptrs=None, drow_nodes=None,
                           drow_rel_ptrs=None):
    """The groupings of the compact RGAT passes (include/het_amd.h: het_rgat_aggregate_compact): (by destination, by feat row,
    by er row, None).  ``srow`` / ``drow`` [E] int64: feat row and er row of every edge position.
    With ``rel_ptrs`` (relation pointers of the positions; they are relation-major): the run-sum form
    (het_rgat_aggregate_compact_runs) -- (by destination, by feat row, None, by (destination, relation)).  That form needs the er
    rows to BE the distinct (relation, destination) pairs: ``drow_nodes`` [S_col] / ``drow_rel_ptrs`` [R+1] (the list the rows
    index) are checked against the edges once per list."""
    by_dst = _plan.get_grouping(None, col, num_nodes, srow, drow)
    by_srow = _plan.get_grouping(None, srow, num_src_rows, col, drow)
    if by_dst is None or by_srow is None:
        return None
    if rel_ptrs is not None and num_nodes * (rel_ptrs.numel() - 1) >= 2 ** 31:
        rel_ptrs = None  # (destination * R + relation does not fit the groupings' int32 keys: the per-edge form)
    if rel_ptrs is not None:
        R = rel_ptrs.numel() - 1

        def build():
            rel = torch.repeat_interleave(torch.arange(R, dtype=torch.int64, device=col.device), rel_ptrs[1:] - rel_ptrs[:-1])
            if drow_nodes is not None:
                ok = bool((drow_nodes[drow] == col).all()) and drow.numel() == rel.numel()
                if ok and drow_rel_ptrs is not None:
                    ok = bool(((drow >= drow_rel_ptrs[rel]) & (drow < drow_rel_ptrs[rel + 1])).all())
                if not ok:
                    raise _lib.HetError("rgat_compact_groupings: the run-sum form needs every er row to be the (relation, "
                                        "destination) pair of its edges")
            return (col * R + rel).contiguous()
        key = _derived_get("dst_rel_key", (col, rel_ptrs, drow) + ((drow_nodes,) if drow_nodes is not None else ()), build)
        return by_dst, by_srow, None, _plan.get_grouping(None, key, num_nodes * R, None, None)
    by_drow = _plan.get_grouping(None, drow, num_dst_rows, grouping_rank_of_position(by_srow), None)
    return by_dst, by_srow, by_drow, None


def rgat_aggregate_compact(groupings, feat_c, el_c, er_c, sum, ret, slope, h_inout=None, num_rels=None, attn_l=None,
                           feat_rel_ptrs=None):
    """h_inout [rows, H*D] (optional): ret's rows are also added into it in place (include/het_amd.h).  ``sum`` receives the
    log-sum-exp of every (destination, head) -- these two entry points subtract a running maximum (no overflow for any el + er).
    Groupings in the run-sum form (rgat_compact_groupings with rel_ptrs; ``num_rels`` required): returns (q_rows, q_sum, q_ref)
    for rgat_backward_compact.  attn_l [R,H,D] + feat_rel_ptrs [R+1] (relation pointers of the feat rows), run-sum form: el_c is
    <feat_c, attn_l[relation of the row]> and the pass may form it from the rows it gathers (include/het_amd.h)."""
    _chk("rgat_aggregate_compact", tuple(t for t in (feat_c, el_c, er_c, sum, ret, h_inout) if t is not None))
    N, H = sum.shape[0], sum.shape[1]
    D = ret.numel() // max(1, N * H)
    if groupings[3] is not None:
        S_col = er_c.shape[0]
        q_rows = torch.empty((S_col, H, D), dtype=ret.dtype, device=ret.device)
        q_sum, q_ref = torch.empty_like(er_c), torch.empty_like(er_c)
        with torch.cuda.device(ret.device):
            nbytes = int(_lib.lib().het_rgat_aggregate_compact_runs_workspace(groupings[0].handle, groupings[3].handle, int(num_rels), H, D,
                                                                              _stream(ret)))
        if nbytes < 0:
            raise _lib.HetError("het_rgat_aggregate_compact_runs_workspace: " + _lib.lib().het_last_error().decode())
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=ret.device) if nbytes else None
        host_ptrs = None
        if attn_l is not None and feat_rel_ptrs is not None:
            _chk("rgat_aggregate_compact", (attn_l,), (feat_rel_ptrs,))
            # (the relation boundaries as host integers: one device read per list, cached by its identity)
            lst = _derived_get("rel_ptrs_host", (feat_rel_ptrs,), lambda: feat_rel_ptrs.tolist())
            host_ptrs = (C.c_int64 * len(lst))(*lst)
        _call(ret, "het_rgat_aggregate_compact_runs", groupings[0].handle, groupings[3].handle, int(num_rels), _p(feat_c), _p(el_c),
              _p(er_c), _p(sum), _p(ret), N, H, D, float(slope), _p(h_inout), 0 if h_inout is None else h_inout.shape[0],
              _p(q_rows), _p(q_sum), _p(q_ref), S_col, _p(attn_l) if host_ptrs is not None else None, host_ptrs, _p(ws), nbytes,
              _stream(ret))
        return q_rows, q_sum, q_ref
    nbytes = int(_lib.lib().het_rgat_aggregate_compact_workspace(groupings[0].handle, H, D))  # (hub destinations only)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=ret.device) if nbytes else None
    _call(ret, "het_rgat_aggregate_compact", groupings[0].handle, _p(feat_c), _p(el_c), _p(er_c), _p(sum), _p(ret), N, H, D,
          float(slope), _p(h_inout), 0 if h_inout is None else h_inout.shape[0], _p(ws), nbytes, _stream(ret))
    return None


def rows_matmul_backward_split_ok(H: int, K: int, D: int) -> bool:
    """Shapes het_rows_matmul_backward_dx / _dw cover (one input head on the matrix cores)."""
    return K in (32, 64, 128) and H * D in (32, 64, 128)


def rows_matmul_backward_dx(rel_ptrs, gather_idx, weights_transposed, gradout, grad_x, atomic):
    """grad_x[gather_idx[i]] (+)= gradout[i] . Wt[r(i)] (include/het_amd.h: het_rows_matmul_backward_dx).  atomic: False "=",
    True "+=" with float atomics, 2 "+=" for lists whose rows are distinct inside every relation (unique (relation, node)
    lists): added relation by relation with plain read-modify-write."""
    _chk("rows_matmul_backward_dx", (weights_transposed, gradout, grad_x), (rel_ptrs,) + (() if gather_idx is None else (gather_idx,)))
    R, H, D, K = weights_transposed.shape
    _call(gradout, "het_rows_matmul_backward_dx", _p(rel_ptrs), R, _p(gather_idx), None, gradout.shape[0], _p(weights_transposed),
          _p(gradout), _p(grad_x), H, K, D, int(atomic), _stream(gradout))


def rows_matmul_backward_dw(rel_ptrs, gather_idx, x, gradout, grad_w, accumulate: bool, colsum: Optional[Tensor] = None):
    """grad_w[r(i)] (+)= x[gather_idx[i]]^T (x) gradout[i]; ``colsum`` [H*D] = SUM_i gradout[i] from the same pass
    (include/het_amd.h: het_rows_matmul_backward_dw, het_rows_matmul_backward_dw_colsum)."""
    _chk("rows_matmul_backward_dw", (x, gradout, grad_w) + (() if colsum is None else (colsum,)),
         (rel_ptrs,) + (() if gather_idx is None else (gather_idx,)))
    R, H, K, D = grad_w.shape
    if colsum is None:
        _call(gradout, "het_rows_matmul_backward_dw", _p(rel_ptrs), R, _p(gather_idx), None, gradout.shape[0], _p(x), _p(gradout),
              _p(grad_w), H, K, D, int(accumulate), _stream(gradout))
        return
    assert colsum.numel() == H * D, "rows_matmul_backward_dw: colsum must hold H*D floats"
    _call(gradout, "het_rows_matmul_backward_dw_colsum", _p(rel_ptrs), R, _p(gather_idx), None, gradout.shape[0], _p(x), _p(gradout),
          _p(grad_w), _p(colsum), H, K, D, int(accumulate), _stream(gradout))


def rgat_node_gemm_ok(R: int, H: int, K: int, D: int) -> bool:
    """Shapes of the node-major input-gradient pass (include/het_amd.h: het_rgat_node_backward_dx)."""
    return bool(_lib.lib().het_rgat_node_gemm_ok(R, H, K, D))


def node_row_map(rel_ptrs, nodes, num_nodes: int):
    """[R, num_nodes] int32: row of (relation, node) in a unique (relation, node) list, -1 where the node has none.  Built
    once per list on the device (het_node_row_map) and cached by the identity of the list."""
    def build():
        R = rel_ptrs.numel() - 1
        m = torch.empty((R, num_nodes), dtype=torch.int32, device=nodes.device)
        _call(nodes, "het_node_row_map", _p(rel_ptrs), R, _p(nodes), nodes.numel(), num_nodes, _p(m), _stream(nodes))
        return m
    _chk("node_row_map", (), (rel_ptrs, nodes))
    return _derived_get(("nodemap", num_nodes), (rel_ptrs, nodes), build)


def node_order_by_presence(row_map, dst_map=None, split=None):
    """[N] int32: the nodes sorted (stably) by which relations they have a row in -- tiles of 32 consecutive entries are then
    homogeneous and the node-major pass multiplies no zero rows (include/het_amd.h: node_order).  ``split``: nodes below it
    stay in front of the others (the owned / halo ranges of a partition are passed as two calls).  Cached per map."""
    def build():
        R, N = row_map.shape
        w = (1 << torch.arange(R, device=row_map.device, dtype=torch.int64)).view(R, 1)
        mask = ((row_map >= 0).to(torch.int64) * w).sum(0)
        if dst_map is not None:
            mask = mask | (((dst_map >= 0).to(torch.int64) * w).sum(0) << R)
        if split is not None:
            mask = mask | ((torch.arange(N, device=row_map.device) >= int(split)).to(torch.int64) << (2 * R))
        return torch.argsort(mask, stable=True).to(torch.int32).contiguous()
    return _derived_get(("node_order", None if split is None else int(split)), (row_map,) + (() if dst_map is None else (dst_map,)), build)


def rgat_node_backward_dx(n_begin, n_end, n_loop, grad_h, loop_wt, g_rows, weights_t, row_map, g_er, wa_t, dst_map, grad_x,
                          node_order=None):
    """grad_x rows [n_begin, n_end) of the one-node RGAT layer in one pass over the nodes (include/het_amd.h).  node_order
    (optional, [N] int32): the nodes of the call are the entries [n_begin, n_end) of this list."""
    _chk("rgat_node_backward_dx", tuple(t for t in (grad_h, loop_wt, g_rows, weights_t, g_er, wa_t, grad_x) if t is not None))
    R, H, D, K = weights_t.shape
    _call(grad_x, "het_rgat_node_backward_dx", int(n_begin), int(n_end), int(n_loop), grad_x.shape[0], R, _p(grad_h), _p(loop_wt),
          _p(g_rows), _p(weights_t), _p(row_map), _p(g_er), _p(wa_t), _p(dst_map), _p(grad_x), H, K, D, _p(node_order),
          _stream(grad_x))


def node_rows_matmul_sum_ok(num_sources: int, KS: int, XO: int) -> bool:
    return bool(_lib.lib().het_node_rows_matmul_sum_ok(int(num_sources), int(KS), int(XO)))


def node_rows_matmul_sum(n_begin, n_end, sources, out, node_order=None):
    """out[n] = SUM_s rows_s[map_s[n]] . wt_s for the nodes at positions [n_begin, n_end) of node_order (include/het_amd.h:
    het_node_rows_matmul_sum).  sources: (rows [n_rows, W] float tensor, first column, map [N] int32 or None (row = node id),
    wt [KS, XO]) each; a source reads the KS columns of its rows that start at ``first column``."""
    N, XO = out.shape
    S = len(sources)
    KS = sources[0][3].shape[0]
    _chk("node_rows_matmul_sum", tuple(t for src in sources for t in (src[0], src[3])) + (out,))
    ptrs = (C.c_void_p * S)(*[src[0].data_ptr() + 4 * int(src[1]) for src in sources])
    strides = (C.c_int64 * S)(*[src[0].shape[1] if src[0].dim() == 2 else src[0].numel() // src[0].shape[0] for src in sources])
    maps = (C.c_void_p * S)(*[None if src[2] is None else src[2].data_ptr() for src in sources])
    ident = (C.c_int64 * S)(*[min(N, src[0].shape[0]) if src[2] is None else 0 for src in sources])
    wts = (C.c_void_p * S)(*[src[3].data_ptr() for src in sources])
    for src in sources:
        assert src[3].shape == (KS, XO) and src[3].is_contiguous() and (src[2] is None or (src[2].dtype == torch.int32 and src[2].numel() == N))
    _call(out, "het_node_rows_matmul_sum_bias", int(n_begin), int(n_end), N, S, ptrs, strides, maps, ident, wts, None, _p(out), KS, XO,
          _p(node_order), _stream(out))


def rows_linear_bias_ok(K: int, X: int) -> bool:
    return K in (32, 64, 128) and X in (32, 64, 128)


def rows_linear_bias(offsets, x, w, bias, out=None):
    """x . w + bias for the rows [offsets[0], offsets[1]) of x (include/het_amd.h: het_rows_linear_bias).  ``out``: a tensor of the
    caller (e.g. allocated under another stream than the one the product is launched on)."""
    _chk("rows_linear_bias", tuple(t for t in (x, w, bias, out) if t is not None), (offsets,))
    if out is None:
        out = torch.empty((x.shape[0], w.shape[1]), dtype=x.dtype, device=x.device)
    _call(x, "het_rows_linear_bias", _p(offsets), _p(x), _p(w), _p(bias), _p(out), x.shape[0], w.shape[0], w.shape[1], _stream(x))
    return out


def rgat_backward_compact(groupings, feat_c, el_c, er_c, sum, ret, gradout, grad_feat_c, grad_el_c, grad_er_c, slope,
                          fold_attn_l=None, row_rel_ptrs=None, grad_bias=None, bias_rows=0, runs=None, drow_nodes=None,
                          grad_attn_l=None):
    """runs = (q_rows, q_sum, q_ref) of rgat_aggregate_compact in the run-sum form, drow_nodes [S_col] int64 the destination of
    every er row: grad_er_c from the run sums (het_rgat_backward_compact_runs).  grad_attn_l [R,H,D] (run-sum form, with
    fold_attn_l): also the weight gradient of el_c = <feat_c, attn_l[r]>, from the same pass."""
    _chk("rgat_backward_compact", tuple(t for t in (feat_c, el_c, er_c, sum, ret, gradout, grad_feat_c, grad_el_c, grad_er_c,
                                                    fold_attn_l, grad_bias, grad_attn_l) + (tuple(runs) if runs else ()) if t is not None),
         tuple(t for t in (row_rel_ptrs, drow_nodes) if t is not None))
    N, H = sum.shape[0], sum.shape[1]
    D = ret.numel() // max(1, N * H)
    if runs:
        with torch.cuda.device(ret.device):
            nbytes = int(_lib.lib().het_rgat_backward_compact_runs_workspace(groupings[1].handle, N, er_c.shape[0], H, D, int(grad_bias is not None),
                                                                             int(grad_attn_l is not None), _stream(ret)))
        if nbytes < 0:
            raise _lib.HetError("het_rgat_backward_compact_runs_workspace: " + _lib.lib().het_last_error().decode())
    else:
        assert grad_attn_l is None, "grad_attn_l comes with the run-sum form"
        nbytes = int(_lib.lib().het_rgat_backward_compact_workspace(N, groupings[1]._keep[1].numel(), H, D, int(grad_bias is not None)))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=ret.device)
    head = (_p(feat_c), _p(el_c), _p(er_c), _p(sum), _p(ret), _p(gradout), _p(grad_feat_c), _p(grad_el_c), _p(grad_er_c),
            _p(fold_attn_l), _p(row_rel_ptrs), 0 if row_rel_ptrs is None else row_rel_ptrs.numel() - 1, _p(grad_bias),
            int(bias_rows), N, feat_c.shape[0], er_c.shape[0], H, D, float(slope))
    tail = (_p(ws), ws.numel() * 4, _stream(ret))
    if runs:
        _call(ret, "het_rgat_backward_compact_runs", groupings[1].handle, _p(runs[0]), _p(runs[1]), _p(runs[2]), _p(drow_nodes), *head,
              _p(grad_attn_l), *tail)
    else:
        _call(ret, "het_rgat_backward_compact", groupings[1].handle, groupings[2].handle, *head, *tail)


def hgt_fold_source_weights(k_lin, v_lin, rel_att, rel_msg, rel_pri, src_type, transpose_att: bool):
    """w_kv [R,1,in,2*H*dk] (include/het_amd.h: het_hgt_fold_source_weights).  k_lin / v_lin [T,1,in,H*dk] contiguous fp32 on the GPU."""
    _chk("hgt_fold_source_weights", (k_lin, v_lin, rel_att, rel_msg, rel_pri), (src_type,))
    R, H, dk, _ = rel_att.shape
    T, K_in = k_lin.shape[0], k_lin.shape[2]
    w = torch.empty((R, 1, K_in, 2 * H * dk), dtype=k_lin.dtype, device=k_lin.device)
    _call(w, "het_hgt_fold_source_weights", _p(k_lin), _p(v_lin), _p(rel_att), _p(rel_msg), _p(rel_pri), _p(src_type), T, R, H, dk, K_in,
          int(transpose_att), _p(w), _stream(w))
    return w


def hgt_fold_source_weights_backward(grad_w, k_lin, v_lin, rel_att, rel_msg, rel_pri, src_type, transpose_att: bool):
    """(grad_k_lin, grad_v_lin, grad_att, grad_msg, grad_pri) of hgt_fold_source_weights."""
    _chk("hgt_fold_source_weights_backward", (grad_w, k_lin, v_lin, rel_att, rel_msg, rel_pri), (src_type,))
    R, H, dk, _ = rel_att.shape
    T, K_in = k_lin.shape[0], k_lin.shape[2]
    outs = tuple(torch.empty_like(t) for t in (k_lin, v_lin, rel_att, rel_msg, rel_pri))
    _call(grad_w, "het_hgt_fold_source_weights_backward", _p(grad_w), _p(k_lin), _p(v_lin), _p(rel_att), _p(rel_msg), _p(rel_pri),
          _p(src_type), T, R, H, dk, K_in, int(transpose_att), *(_p(o) for o in outs), _stream(grad_w))
    return outs


def hgt_compact_shape_ok(H: int, D: int) -> bool:
    return bool(_lib.lib().het_hgt_compact_shape_ok(int(H), int(D)))


def destination_lists(col, offsets):
    """(dst_nodes [S_dst] sorted, rank_of_edge [E], run_ptrs [len(offsets)]) for the destinations that HAVE in-edges: the
    distinct values of ``col``, the index of every edge's destination in that list, and the list split at the node-type
    offsets (rows of a node type are a contiguous piece of the sorted list).  Built once per graph (cached)."""
    def build():
        nodes, inv = torch.unique(col, return_inverse=True)
        ptrs = torch.searchsorted(nodes, offsets.to(nodes.dtype)).contiguous()
        return nodes.contiguous(), inv.contiguous(), ptrs
    return _derived_get("dst_lists", (col, offsets), build)


def hgt_compact_groupings(col, srow, num_nodes, num_src_rows):
    """The two groupings of the compact HGT passes (include/het_amd.h: het_hgt_aggregate_compact): by destination and by
    (relation, source) row.  ``srow`` [E] int64: that row of every edge position."""
    by_dst = _plan.get_grouping(None, col, num_nodes, srow, None)
    by_srow = _plan.get_grouping(None, srow, num_src_rows, col, None)
    return None if by_dst is None or by_srow is None else (by_dst, by_srow)


def hgt_aggregate_compact(groupings, kv_c, q, lsum, out):
    _chk("hgt_aggregate_compact", (kv_c, q, lsum, out))
    N, H = lsum.shape
    D = out.numel() // max(1, N * H)
    nbytes = int(_lib.lib().het_hgt_aggregate_compact_workspace(groupings[0].handle, H, D))  # (hub destinations only)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=out.device) if nbytes else None
    _call(out, "het_hgt_aggregate_compact", groupings[0].handle, _p(kv_c), _p(q), _p(lsum), _p(out), N, kv_c.shape[0], H, D,
          _p(ws), nbytes, _stream(out))


def hgt_backward_compact(groupings, kv_c, q, lsum, out, gradout, grad_kv_c, grad_q):
    _chk("hgt_backward_compact", (kv_c, q, lsum, out, gradout, grad_kv_c, grad_q))
    N, H = lsum.shape
    D = out.numel() // max(1, N * H)
    nbytes = int(_lib.lib().het_hgt_backward_compact_workspace(N, H))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=out.device)
    _call(out, "het_hgt_backward_compact", groupings[0].handle, groupings[1].handle, _p(kv_c), _p(q), _p(lsum), _p(out),
          _p(gradout), _p(grad_kv_c), _p(grad_q), N, kv_c.shape[0], H, D, _p(ws), ws.numel() * 4, _stream(out))


def gat_rank_of_position(rel_ptrs, row, col, eids, num_nodes):
    """[E] int64: the rank of every separate-COO position in the destination-grouped order of the kind-0 GAT kernels
    (row j of exp_sorted / grad_el_sorted belongs to the position whose rank is j).  None without groupings."""
    g = _by_dst(0, (None, None, None, None), rel_ptrs, row, col, eids, num_nodes)
    if g is None:
        return None
    r = getattr(g, "rank_of_position", None)
    if r is None:
        r = torch.empty(eids.numel(), dtype=torch.int64, device=eids.device)
        _call(r, "het_grouping_rank_of_position", g.handle, _p(r), _stream(r))
        g.rank_of_position = r  # lives (and is evicted) with the grouping
    return r


@_op("relational_fused_gat_csr(Tensor incsr_row_ptr, Tensor incsr_col_indices, Tensor incsr_eids, Tensor incsr_reltypes, "
     "Tensor unique_srcs_and_dests_rel_ptrs, Tensor unique_srcs_and_dests_node_indices, Tensor feat_src, Tensor el, "
     "Tensor er, Tensor(a!) sum, Tensor(b!) exp, Tensor(c!) ret, float slope, bool CompactAsOfNodeFlag=False) -> ()")
def relational_fused_gat_csr(incsr_row_ptr, incsr_col_indices, incsr_eids, incsr_reltypes,
                             unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, feat_src, el, er, sum,
                             exp, ret, slope, CompactAsOfNodeFlag=False):
    _chk("relational_fused_gat_csr", (feat_src, el, er, sum, exp, ret),
         (incsr_row_ptr, incsr_col_indices, incsr_eids, incsr_reltypes))
    N, E, H = incsr_row_ptr.numel() - 1, incsr_eids.numel(), el.shape[1]
    D = ret.numel() // max(1, N * H)
    if not CompactAsOfNodeFlag and _plan.is_enabled() and E > 0 and gat_grouped_shape_ok(H, D):
        # the in-CSR IS the edge list grouped by destination: same math as the separate-COO op on (eids, src, dst) in
        # CSR order, served by the destination-grouped kernels instead of E*H*D float atomics
        dst, rp1 = _csr_expanded_rows(incsr_row_ptr, E)
        fused_gat_forward(incsr_eids, rp1, incsr_col_indices, dst, 0, {}, feat_src, el, er, sum, exp, ret, slope, None)
        return
    if CompactAsOfNodeFlag and _plan.is_enabled() and E > 0 and gat_grouped_shape_ok(H, D):
        # compact rows: the same op as the separate-COO pair with CompactAsOfNodeKind 4 once every edge id knows its feat
        # row and er row (cached per graph) -- destination-grouped kernels instead of float atomics
        dst, maps = _csr_compact_maps(incsr_row_ptr, incsr_col_indices, incsr_eids, incsr_reltypes,
                                      unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, True)
        _, rp1 = _csr_expanded_rows(incsr_row_ptr, E)
        fused_gat_forward(incsr_eids, rp1, incsr_col_indices, dst, 4, maps, feat_src, el, er, sum, exp, ret, slope, None)
        return
    _call(ret, "het_relational_fused_gat_csr", _p(incsr_row_ptr), _p(incsr_col_indices), _p(incsr_eids),
          _p(incsr_reltypes), N, E, _p(unique_srcs_and_dests_rel_ptrs), _p(unique_srcs_and_dests_node_indices),
          max(0, unique_srcs_and_dests_rel_ptrs.numel() - 1), _p(feat_src), _p(el), _p(er), _p(sum), _p(exp), _p(ret),
          H, D, float(slope), int(CompactAsOfNodeFlag), _stream(ret))


@_op("backward_relational_fused_gat_csr(Tensor outcsr_row_ptr, Tensor outcsr_col_indices, Tensor outcsr_eids, "
     "Tensor outcsr_reltypes, Tensor unique_srcs_and_dests_rel_ptrs, Tensor unique_srcs_and_dests_node_indices, "
     "Tensor feat_src, Tensor el, Tensor er, Tensor sum, Tensor exp, Tensor ret, Tensor gradout, "
     "Tensor(a!) grad_feat_src, Tensor(b!) grad_el, Tensor(c!) grad_er, float slope, bool CompactAsOfNodeFlag=False) -> ()")
def backward_relational_fused_gat_csr(outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes,
                                      unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, feat_src, el,
                                      er, sum, exp, ret, gradout, grad_feat_src, grad_el, grad_er, slope,
                                      CompactAsOfNodeFlag=False):
    _chk("backward_relational_fused_gat_csr", (feat_src, el, er, sum, exp, ret, gradout, grad_feat_src, grad_el, grad_er),
         (outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes))
    N, E, H = outcsr_row_ptr.numel() - 1, outcsr_eids.numel(), el.shape[1]
    D = ret.numel() // max(1, N * H)
    if not CompactAsOfNodeFlag and _plan.is_enabled() and E > 0 and gat_grouped_shape_ok(H, D) and slope >= 0:
        # out-CSR rows are the sources: (eids, src, dst) in CSR order through the destination-grouped backward (every
        # gradient row of kind 0 belongs to one edge, so "+=" into the zero-filled buffers equals the stores it does)
        src, rp1 = _csr_expanded_rows(outcsr_row_ptr, E)
        fused_gat_backward(outcsr_eids, rp1, src, outcsr_col_indices, 0, {}, feat_src, el, er, sum, exp, ret, gradout,
                           grad_feat_src, grad_el, grad_er, slope, None)
        return
    if CompactAsOfNodeFlag and _plan.is_enabled() and E > 0 and gat_grouped_shape_ok(H, D) and slope >= 0:
        src, maps = _csr_compact_maps(outcsr_row_ptr, outcsr_col_indices, outcsr_eids, outcsr_reltypes,
                                      unique_srcs_and_dests_rel_ptrs, unique_srcs_and_dests_node_indices, False)
        _, rp1 = _csr_expanded_rows(outcsr_row_ptr, E)
        # "+=" contract of the reference-named op: the kind-4 kernels overwrite, so they run into temporaries that are added
        gf, gl, gr = torch.zeros_like(grad_feat_src), torch.zeros_like(grad_el), torch.zeros_like(grad_er)
        fused_gat_backward(outcsr_eids, rp1, src, outcsr_col_indices, 4, maps, feat_src, el, er, sum, exp, ret, gradout,
                           gf, gl, gr, slope, None)
        grad_feat_src += gf
        grad_el += gl
        grad_er += gr
        return
    _call(ret, "het_backward_relational_fused_gat_csr", _p(outcsr_row_ptr), _p(outcsr_col_indices), _p(outcsr_eids),
          _p(outcsr_reltypes), N, E, _p(unique_srcs_and_dests_rel_ptrs), _p(unique_srcs_and_dests_node_indices),
          max(0, unique_srcs_and_dests_rel_ptrs.numel() - 1), _p(feat_src), _p(el), _p(er), _p(sum), _p(exp), _p(ret),
          _p(gradout), _p(grad_feat_src), _p(grad_el), _p(grad_er), H, D, float(slope), int(CompactAsOfNodeFlag),
          _stream(ret))


# ------------------------------------------------------------------------------------
# RGCN
# ------------------------------------------------------------------------------------
@_op("rgcn_layer1_separate_coo(Tensor separate_coo_relptrs, Tensor separate_coo_eids, Tensor separate_coo_row_indices, "
     "Tensor separate_coo_col_indices, Tensor node_feat_input, Tensor weights, Tensor edge_norm, "
     "Tensor(a!) node_feat_output) -> ()")
def rgcn_layer1_separate_coo(separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices,
                             separate_coo_col_indices, node_feat_input, weights, edge_norm, node_feat_output):
    _chk("rgcn_layer1_separate_coo", (node_feat_input, weights, edge_norm, node_feat_output),
         (separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices))
    R, K, D = weights.shape
    N = node_feat_output.shape[0]
    g = _plan.get_grouping(separate_coo_relptrs, separate_coo_col_indices, N, separate_coo_row_indices, separate_coo_eids)
    ws = None if g is None else torch.empty(max(1, g.num_segments) * K, dtype=torch.float32, device=weights.device)
    _call(node_feat_output, "het_rgcn_layer1_separate_coo", _p(separate_coo_relptrs), _p(separate_coo_eids),
          _p(separate_coo_row_indices), _p(separate_coo_col_indices), R, separate_coo_eids.numel(), N,
          _p(node_feat_input), _p(weights), _p(edge_norm), _p(node_feat_output), K, D, None if g is None else g.handle,
          _p(ws), 0 if ws is None else ws.numel() * 4, _stream(node_feat_output))


@_op("backward_rgcn_layer1_separate_coo(Tensor separate_coo_relptrs, Tensor separate_coo_eids, "
     "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor node_feat_input, "
     "Tensor weights_transposed, Tensor edge_norm, Tensor(a!) grad_edge_norm, Tensor(b!) delta_node_feat_input, "
     "Tensor delta_node_feat_output, Tensor(c!) delta_weights) -> ()")
def backward_rgcn_layer1_separate_coo(separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices,
                                      separate_coo_col_indices, node_feat_input, weights_transposed, edge_norm,
                                      grad_edge_norm, delta_node_feat_input, delta_node_feat_output, delta_weights):
    _chk("backward_rgcn_layer1_separate_coo",
         (node_feat_input, weights_transposed, edge_norm, delta_node_feat_input, delta_node_feat_output, delta_weights),
         (separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices))
    R, D, K = weights_transposed.shape
    N = delta_node_feat_output.shape[0]
    g = _plan.get_grouping(separate_coo_relptrs, separate_coo_row_indices, node_feat_input.shape[0],
                           separate_coo_col_indices, separate_coo_eids)
    ws = None if g is None else torch.empty(max(1, g.num_segments) * D, dtype=torch.float32, device=delta_weights.device)
    _call(delta_weights, "het_backward_rgcn_layer1_separate_coo", _p(separate_coo_relptrs), _p(separate_coo_eids),
          _p(separate_coo_row_indices), _p(separate_coo_col_indices), R, separate_coo_eids.numel(), N,
          _p(node_feat_input), _p(weights_transposed), _p(edge_norm), _p(grad_edge_norm), _p(delta_node_feat_input),
          _p(delta_node_feat_output), _p(delta_weights), K, D, None if g is None else g.handle, _p(ws),
          0 if ws is None else ws.numel() * 4, _stream(delta_weights))


def rgcn_layer_ok(R: int, K: int, D: int) -> bool:
    """Shapes of the two-call RGCN layer (include/het_amd.h: het_rgcn_layer_forward)."""
    return bool(_lib.lib().het_rgcn_layer_ok(int(R), int(K), int(D)))


def _grouping_segment_map(g, rel_ptrs, keys, num_keys: int):
    """[R, num_keys] int32: segment of (relation, key) in the grouping ``g`` of ``keys`` by (relation, key), -1 = none (cached by
    the identity of the lists the grouping was built from)."""
    def build():
        R = rel_ptrs.numel() - 1
        m = torch.empty((R, num_keys), dtype=torch.int32, device=keys.device)
        _call(keys, "het_grouping_segment_map", g.handle, num_keys, _p(m), _stream(keys))
        return m
    return _derived_get(("segmap", num_keys), (rel_ptrs, keys), build)


def rgcn_layer_plan(rel_ptrs, eids, row, col, num_nodes: int):
    """What the two-call RGCN layer needs per graph (groupings from het_amd.plan, the rest cached by tensor identity):
    (by_rel_dst, by_rel_src, dst_map, dst_order, src_map, src_order), or None without groupings.  The maps come from the
    groupings' own segment lists, so a graph needs no unique (relation, node) lists for this layer."""
    gd = _plan.get_grouping(rel_ptrs, col, num_nodes, row, eids)
    gs = _plan.get_grouping(rel_ptrs, row, num_nodes, col, eids)
    if gd is None or gs is None:
        return None
    dst_map = _grouping_segment_map(gd, rel_ptrs, col, num_nodes)
    src_map = _grouping_segment_map(gs, rel_ptrs, row, num_nodes)
    return gd, gs, dst_map, node_order_by_presence(dst_map), src_map, node_order_by_presence(src_map)


SCALE_SORTED = _os.environ.get("HET_RGCN_NORM_SORTED", "1") != "0"  # A/B: 0 = the norm is gathered by edge id in every pass


def scale_in_rank_order(g, values):
    """``values`` ([E] or [E, H] by edge id) in the order of the grouping ``g`` (whose second payload is the edge id), or None.
    For a scale that is the same tensor step after step (an edge norm): built at its SECOND sighting -- a scale that changes every
    step never pays for it -- and kept with the grouping, one per grouping (a new tensor replaces it); the entry holds the source
    tensor, so its address cannot come back with other contents, and (data_ptr, numel, _version) catches in-place edits."""
    if not SCALE_SORTED or g is None or values is None:
        return None
    ident = _plan._ident(values)
    with _derived_lock:
        hit = getattr(g, "_scale_sorted", None)
        if hit is not None and hit[0] == ident:
            return hit[2]
        if getattr(g, "_scale_seen", None) != ident:
            g._scale_seen, g._scale_sorted = ident, None
            return None
        out = torch.empty_like(values, memory_format=torch.contiguous_format)
        H = values.numel() // max(1, values.shape[0])
        _call(values, "het_grouping_gather_payload1", g.handle, _p(values), H, _p(out), _stream(values))
        g._scale_sorted = (ident, values, out)
        return out


def rgcn_layer_forward(plan, x, weights, norm, bias):
    """(ret [N,D], ssum [S_col,K]) of het_rgcn_layer_forward: ret = bias + SUM_r (SUM_e norm x[src]) . W[r]."""
    gd, _, dst_map, dst_order, _, _ = plan
    _chk("rgcn_layer_forward", tuple(t for t in (x, weights, norm, bias) if t is not None))
    R, K, D = weights.shape
    N = dst_map.shape[1]
    ssum = torch.empty((max(1, gd.num_segments), K), dtype=torch.float32, device=x.device)
    ret = torch.empty((N, D), dtype=torch.float32, device=x.device)
    _call(ret, "het_rgcn_layer_forward", gd.handle, R, N, _p(x), _p(weights), _p(norm), _p(scale_in_rank_order(gd, norm)), _p(bias),
          _p(dst_map), _p(dst_order),
          _p(ssum), _p(ret), K, D, _stream(ret))
    return ret, ssum


def rgcn_layer_backward(plan, ssum, weights_t, norm, gradout, want_bias: bool, want_x: bool = True):
    """(grad_x [N,K] or None, grad_w [R,K,D], grad_bias [D] or None) of het_rgcn_layer_backward."""
    gd, gs, _, _, src_map, src_order = plan
    _chk("rgcn_layer_backward", (ssum, weights_t, norm, gradout))
    R, D, K = weights_t.shape
    N = src_map.shape[1]
    dev = gradout.device
    grad_x = torch.empty((N, K), dtype=torch.float32, device=dev) if want_x else None
    grad_w = torch.empty((R, K, D), dtype=torch.float32, device=dev)
    grad_bias = torch.empty((D,), dtype=torch.float32, device=dev) if want_bias else None
    nbytes = int(_lib.lib().het_rgcn_layer_backward_workspace(gs.num_segments, D))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    _call(gradout, "het_rgcn_layer_backward", gs.handle, gd.handle, R, N, gradout.shape[0], _p(ssum), _p(weights_t), _p(norm),
          _p(scale_in_rank_order(gs, norm) if want_x else None),
          _p(gradout), _p(src_map), _p(src_order), _p(grad_x), _p(grad_w), _p(grad_bias), K, D, _p(ws),
          ws.numel() * 4, _stream(gradout))
    return grad_x, grad_w, grad_bias


def _rgcn_maps(d: Dict[str, Tensor], direct: bool):
    if direct:
        return d["inverse_indices_row"], None
    return d["rel_ptrs_row"], d["node_indices_row"]


@_op("rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(Tensor separate_coo_eids, Tensor separate_coo_rel_ptrs, "
     "Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Dict(str, Tensor) args_tensor_dict, "
     "Tensor feat_src, Tensor enorm, Tensor(a!) ret, bool DirectIndexFlag) -> ()")
def rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(separate_coo_eids, separate_coo_rel_ptrs,
                                                               separate_coo_row_indices, separate_coo_col_indices,
                                                               args_tensor_dict, feat_src, enorm, ret, DirectIndexFlag):
    a, b = _rgcn_maps(args_tensor_dict, DirectIndexFlag)
    _chk("rgcn_node_mean_aggregation_compact_as_of_node_separate_coo", (feat_src, enorm, ret),
         (separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices, a)
         + ((b,) if b is not None else ()))
    N = ret.shape[0]
    g = None
    if _plan.is_enabled() and separate_coo_eids.numel() > 0:
        crow = _src_rows_by_position(4 if DirectIndexFlag else 3, (a, b, None, None), separate_coo_rel_ptrs,
                                     separate_coo_row_indices, separate_coo_eids)
        g = _plan.get_grouping(None, separate_coo_col_indices, N, crow, separate_coo_eids)
    _call(ret, "het_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo", _p(separate_coo_eids),
          _p(separate_coo_rel_ptrs), _p(separate_coo_row_indices), _p(separate_coo_col_indices),
          separate_coo_rel_ptrs.numel() - 1, separate_coo_eids.numel(), N, _p(a), _p(b), _p(feat_src), _p(enorm), _p(ret),
          ret.numel() // max(1, N), int(DirectIndexFlag), None if g is None else g.handle, _stream(ret))


@_op("backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(Tensor separate_coo_eids, "
     "Tensor separate_coo_rel_ptrs, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, "
     "Dict(str, Tensor) args_tensor_dict, Tensor feat_src, Tensor enorm, Tensor ret, Tensor gradout, "
     "Tensor(a!) grad_feat_src, bool DirectIndexFlag) -> ()")
def backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(separate_coo_eids, separate_coo_rel_ptrs,
                                                                        separate_coo_row_indices,
                                                                        separate_coo_col_indices, args_tensor_dict,
                                                                        feat_src, enorm, ret, gradout, grad_feat_src,
                                                                        DirectIndexFlag):
    a, b = _rgcn_maps(args_tensor_dict, DirectIndexFlag)
    _chk("backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo", (feat_src, enorm, ret, gradout, grad_feat_src),
         (separate_coo_eids, separate_coo_rel_ptrs, separate_coo_row_indices, separate_coo_col_indices, a)
         + ((b,) if b is not None else ()))
    N = ret.shape[0]
    g = None
    if _plan.is_enabled() and separate_coo_eids.numel() > 0:
        crow = _src_rows_by_position(4 if DirectIndexFlag else 3, (a, b, None, None), separate_coo_rel_ptrs,
                                     separate_coo_row_indices, separate_coo_eids)
        g = _plan.get_grouping(None, crow, grad_feat_src.shape[0], separate_coo_col_indices, separate_coo_eids)
    _call(ret, "het_backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo", _p(separate_coo_eids),
          _p(separate_coo_rel_ptrs), _p(separate_coo_row_indices), _p(separate_coo_col_indices),
          separate_coo_rel_ptrs.numel() - 1, separate_coo_eids.numel(), N, _p(a), _p(b), _p(feat_src), _p(enorm), _p(ret),
          _p(gradout), _p(grad_feat_src), ret.numel() // max(1, N), int(DirectIndexFlag),
          None if g is None else g.handle, grad_feat_src.shape[0], _stream(ret))


K = getattr(torch.ops, NAMESPACE)
REGISTERED_OPS = tuple(_registered)


# ------------------------------------------------------------------------------------
# HGT
# ------------------------------------------------------------------------------------
def _ip_maps(d: Dict[str, Tensor], kind: int):
    if kind == 0:
        return None, None
    if kind == 1:
        return d["unique_srcs_and_dests_rel_ptrs"], d["unique_srcs_and_dests_node_indices"]
    if kind == 2:
        return d["edata_idx_to_inverse_idx"], None
    raise _lib.HetError(f"rgnn_inner_product_right_node: CompactAsOfNodeKind {kind} not supported")


def _ip_direct(d: Dict[str, Tensor], kind: int, rel_ptrs, col, eids):
    """(kind, map_a, map_b) as handed to the C entry points: the binary-search kind 1 is turned into the direct-index
    kind 2 once per graph (left row of every edge id; cached), which is what the fast row kernels read."""
    a, b = _ip_maps(d, kind)
    if kind != 1 or not _plan.is_enabled() or eids.numel() == 0:
        return kind, a, b
    def build():
        lrow = _rows_by_search(rel_ptrs, col, a, b)
        m = torch.empty(int(eids.max().item()) + 1, dtype=torch.int64, device=col.device)
        m[eids] = lrow
        return m

    hit = (_derived_get("ipmap", (a, b, rel_ptrs, col, eids), build),)
    return 2, hit[0], None


@_op("rgnn_inner_product_right_node_separatecoo(Dict(str, Tensor) arg_tensor_dict, int IntKind, "
     "Tensor separate_coo_rel_ptrs, Tensor separate_coo_eids, Tensor separate_coo_row_indices, "
     "Tensor separate_coo_col_indices, Tensor left_side_data, Tensor right_node_vectors, "
     "Tensor(a!) edge_inner_product) -> ()")
def rgnn_inner_product_right_node_separatecoo(arg_tensor_dict, IntKind, separate_coo_rel_ptrs, separate_coo_eids,
                                              separate_coo_row_indices, separate_coo_col_indices, left_side_data,
                                              right_node_vectors, edge_inner_product):
    a, b = _ip_maps(arg_tensor_dict, IntKind)
    _chk("rgnn_inner_product_right_node_separatecoo", (left_side_data, right_node_vectors, edge_inner_product),
         (separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices)
         + tuple(t for t in (a, b) if t is not None))
    IntKind, a, b = _ip_direct(arg_tensor_dict, IntKind, separate_coo_rel_ptrs, separate_coo_col_indices, separate_coo_eids)
    H = edge_inner_product.shape[1]
    D = right_node_vectors.numel() // max(1, right_node_vectors.shape[0] * H)
    _call(edge_inner_product, "het_rgnn_inner_product_right_node_separatecoo", IntKind, _p(a), _p(b),
          _p(separate_coo_rel_ptrs), _p(separate_coo_eids), _p(separate_coo_row_indices), _p(separate_coo_col_indices),
          separate_coo_rel_ptrs.numel() - 1, separate_coo_eids.numel(), _p(left_side_data), _p(right_node_vectors),
          _p(edge_inner_product), H, D, _stream(edge_inner_product))


@_op("backward_inner_product_right_node_separatecoo(Dict(str, Tensor) arg_tensor_dict, int IntKind, "
     "Tensor separate_coo_rel_ptrs, Tensor separate_coo_eids, Tensor separate_coo_row_indices, "
     "Tensor separate_coo_col_indices, Tensor left_side_data, Tensor right_node_vectors, Tensor gradout, "
     "Tensor(a!) grad_left_side_data, Tensor(b!) grad_right_node_vectors) -> ()")
def backward_inner_product_right_node_separatecoo(arg_tensor_dict, IntKind, separate_coo_rel_ptrs, separate_coo_eids,
                                                  separate_coo_row_indices, separate_coo_col_indices, left_side_data,
                                                  right_node_vectors, gradout, grad_left_side_data,
                                                  grad_right_node_vectors):
    a, b = _ip_maps(arg_tensor_dict, IntKind)
    _chk("backward_inner_product_right_node_separatecoo",
         (left_side_data, right_node_vectors, gradout, grad_left_side_data, grad_right_node_vectors),
         (separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices)
         + tuple(t for t in (a, b) if t is not None))
    inner_product_backward(arg_tensor_dict, IntKind, separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices,
                           separate_coo_col_indices, left_side_data, right_node_vectors, gradout, grad_left_side_data,
                           grad_right_node_vectors, accumulate=True, checked=True)


def inner_product_backward(arg_tensor_dict, IntKind, separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices,
                           separate_coo_col_indices, left_side_data, right_node_vectors, gradout, grad_left_side_data,
                           grad_right_node_vectors, accumulate: bool, checked: bool = False):
    a, b = _ip_maps(arg_tensor_dict, IntKind)
    if not checked:
        _chk("backward_inner_product_right_node_separatecoo",
             (left_side_data, right_node_vectors, gradout, grad_left_side_data, grad_right_node_vectors),
             (separate_coo_rel_ptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices)
             + tuple(t for t in (a, b) if t is not None))
    H = gradout.shape[1]
    D = right_node_vectors.numel() // max(1, right_node_vectors.shape[0] * H)
    IntKind, a, b = _ip_direct(arg_tensor_dict, IntKind, separate_coo_rel_ptrs, separate_coo_col_indices, separate_coo_eids)
    g = gl = None
    if IntKind in (0, 2) and _plan.is_enabled():
        if IntKind == 0:
            lrow = separate_coo_eids
        else:
            lrow = _derived_get("iplrow", (a, separate_coo_eids), lambda: a[separate_coo_eids].contiguous())
            # compact left rows are shared by many edges: their gradient is a segmented sum over the edges of a row
            gl = _plan.get_grouping(None, lrow, left_side_data.shape[0], separate_coo_row_indices, separate_coo_eids)
        g = _plan.get_grouping(None, separate_coo_row_indices, right_node_vectors.shape[0], lrow, separate_coo_eids)
    _call(gradout, "het_backward_inner_product_right_node_separatecoo", IntKind, _p(a), _p(b),
          _p(separate_coo_rel_ptrs), _p(separate_coo_eids), _p(separate_coo_row_indices), _p(separate_coo_col_indices),
          separate_coo_rel_ptrs.numel() - 1, separate_coo_eids.numel(), _p(left_side_data), _p(right_node_vectors),
          _p(gradout), _p(grad_left_side_data), _p(grad_right_node_vectors), H, D, int(accumulate),
          None if g is None else g.handle, None if gl is None else gl.handle, left_side_data.shape[0],
          right_node_vectors.shape[0], _stream(gradout))


@_op("hgt_full_graph_edge_softmax_ops_separate_coo(Tensor row_indices, Tensor col_indices, Tensor eids, Tensor rel_ptrs, "
     "Tensor unnormalized_attn_score, Tensor mu, Tensor(a!) edgesoftmax_sum_per_node, "
     "Tensor(b!) mu_softmax_applied_unnormalized_attn_score, Tensor(c!) normalized_attn_score) -> ()")
def hgt_full_graph_edge_softmax_ops_separate_coo(row_indices, col_indices, eids, rel_ptrs, unnormalized_attn_score, mu,
                                                 edgesoftmax_sum_per_node, mu_softmax_applied_unnormalized_attn_score,
                                                 normalized_attn_score):
    _chk("hgt_full_graph_edge_softmax_ops_separate_coo",
         (unnormalized_attn_score, mu, edgesoftmax_sum_per_node, mu_softmax_applied_unnormalized_attn_score,
          normalized_attn_score), (row_indices, col_indices, eids, rel_ptrs))
    H = mu.shape[1]
    N = edgesoftmax_sum_per_node.shape[0]
    g = _by_dst(0, None, rel_ptrs, row_indices, col_indices, eids, N) if H % 4 == 0 and eids.numel() > 0 else None
    _call(mu, "het_hgt_full_graph_edge_softmax_ops_separate_coo", _p(row_indices), _p(col_indices), _p(eids),
          _p(rel_ptrs), rel_ptrs.numel() - 1, eids.numel(), N,
          _p(unnormalized_attn_score), _p(mu), _p(edgesoftmax_sum_per_node),
          _p(mu_softmax_applied_unnormalized_attn_score), _p(normalized_attn_score), H,
          None if g is None else g.handle, _stream(mu))


@_op("backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(Tensor row_indices, Tensor col_indices, "
     "Tensor eids, Tensor rel_ptrs, Tensor unnormalized_attn_score, Tensor normalized_attn_score, "
     "Tensor grad_normalized_attn_score, Tensor mu, Tensor(a!) grad_unnormalized_attn_score, Tensor(b!) grad_mu, "
     "Tensor(c!) sum_incoming_edges_product_softmax_score) -> ()")
def backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(
        row_indices, col_indices, eids, rel_ptrs, unnormalized_attn_score, normalized_attn_score,
        grad_normalized_attn_score, mu, grad_unnormalized_attn_score, grad_mu, sum_incoming_edges_product_softmax_score):
    _chk("backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo",
         (unnormalized_attn_score, normalized_attn_score, grad_normalized_attn_score, mu, grad_unnormalized_attn_score,
          grad_mu, sum_incoming_edges_product_softmax_score), (row_indices, col_indices, eids, rel_ptrs))
    H = mu.shape[1]
    N = sum_incoming_edges_product_softmax_score.shape[0]
    g = _by_dst(0, None, rel_ptrs, row_indices, col_indices, eids, N) if H % 4 == 0 and eids.numel() > 0 else None
    _call(mu, "het_backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo", _p(row_indices),
          _p(col_indices), _p(eids), _p(rel_ptrs), rel_ptrs.numel() - 1, eids.numel(),
          N, _p(unnormalized_attn_score), _p(normalized_attn_score),
          _p(grad_normalized_attn_score), _p(mu), _p(grad_unnormalized_attn_score), _p(grad_mu),
          _p(sum_incoming_edges_product_softmax_score), H, None if g is None else g.handle, _stream(mu))


@_op("hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(Tensor separate_coo_relptrs, "
     "Tensor separate_coo_eids, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor inputs, "
     "Tensor weights, Tensor edge_norm, Tensor(a!) new_h) -> ()")
def hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(separate_coo_relptrs, separate_coo_eids,
                                                                        separate_coo_row_indices,
                                                                        separate_coo_col_indices, inputs, weights,
                                                                        edge_norm, new_h):
    _chk("hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo", (inputs, weights, edge_norm, new_h),
         (separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices))
    R, H, dk, dout = weights.shape
    g = _plan.get_grouping(separate_coo_relptrs, separate_coo_col_indices, new_h.shape[0], separate_coo_row_indices,
                           separate_coo_eids)
    ws = None if g is None else torch.empty(max(1, g.num_segments) * H * dk, dtype=torch.float32, device=new_h.device)
    _call(new_h, "het_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo", _p(separate_coo_relptrs),
          _p(separate_coo_eids), _p(separate_coo_row_indices), _p(separate_coo_col_indices), R,
          separate_coo_eids.numel(), new_h.shape[0], _p(inputs), _p(weights), _p(edge_norm), _p(new_h), H, dk, dout,
          None if g is None else g.handle, _p(ws), 0 if ws is None else ws.numel() * 4, _stream(new_h))


@_op("backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(Tensor separate_coo_relptrs, "
     "Tensor separate_coo_eids, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, Tensor inputs, "
     "Tensor weights_transposed, Tensor edge_norm, Tensor new_h, Tensor(a!) grad_input, Tensor(b!) grad_weights, "
     "Tensor(c!) grad_edge_norm, Tensor gradout) -> ()")
def backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
        separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices, inputs,
        weights_transposed, edge_norm, new_h, grad_input, grad_weights, grad_edge_norm, gradout):
    _chk("backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo",
         (inputs, weights_transposed, edge_norm, new_h, grad_input, grad_weights, grad_edge_norm, gradout),
         (separate_coo_relptrs, separate_coo_eids, separate_coo_row_indices, separate_coo_col_indices))
    R, H, dout, dk = weights_transposed.shape
    g = _plan.get_grouping(separate_coo_relptrs, separate_coo_row_indices, inputs.shape[0], separate_coo_col_indices,
                           separate_coo_eids)
    # segment sums of the gradient; for wide heads also the per-(relation, source) message rows and the untransposed weight
    ws = None if g is None else torch.empty(2 * max(1, g.num_segments) * H * dout + weights_transposed.numel(),
                                            dtype=torch.float32, device=gradout.device)
    _call(gradout, "het_backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo",
          _p(separate_coo_relptrs), _p(separate_coo_eids), _p(separate_coo_row_indices), _p(separate_coo_col_indices),
          R, separate_coo_eids.numel(), new_h.shape[0], _p(inputs), _p(weights_transposed), _p(edge_norm), _p(new_h),
          _p(grad_input), _p(grad_weights), _p(grad_edge_norm), _p(gradout), H, dk, dout,
          None if g is None else g.handle, _p(ws), 0 if ws is None else ws.numel() * 4, _stream(gradout))


@_op("hgt_full_graph_hetero_attention_ops_coo(Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, "
     "Tensor separate_coo_eids, Tensor separate_coo_relptrs, Tensor applied_klinear_node_features, "
     "Tensor applied_qlinear_node_features, Tensor attn_score_weight, Tensor(a!) attn_score_inner_product, "
     "Tensor(b!) unnormalized_attn_score) -> ()")
def hgt_full_graph_hetero_attention_ops_coo(separate_coo_row_indices, separate_coo_col_indices, separate_coo_eids,
                                            separate_coo_relptrs, applied_klinear_node_features,
                                            applied_qlinear_node_features, attn_score_weight, attn_score_inner_product,
                                            unnormalized_attn_score):
    _chk("hgt_full_graph_hetero_attention_ops_coo",
         (applied_klinear_node_features, applied_qlinear_node_features, attn_score_weight, attn_score_inner_product,
          unnormalized_attn_score),
         (separate_coo_row_indices, separate_coo_col_indices, separate_coo_eids, separate_coo_relptrs))
    R, H, dk, dout = attn_score_weight.shape
    _call(unnormalized_attn_score, "het_hgt_full_graph_hetero_attention_ops_coo", _p(separate_coo_row_indices),
          _p(separate_coo_col_indices), _p(separate_coo_eids), _p(separate_coo_relptrs), R, separate_coo_eids.numel(),
          _p(applied_klinear_node_features), _p(applied_qlinear_node_features), _p(attn_score_weight),
          _p(attn_score_inner_product), _p(unnormalized_attn_score), H, dk, dout, _stream(unnormalized_attn_score))


@_op("backward_hgt_full_graph_hetero_attention_ops_coo(Tensor incsr_row_ptrs, Tensor incsr_col_indices, "
     "Tensor incsr_eids, Tensor incsr_reltypes, Tensor separate_coo_row_indices, Tensor separate_coo_col_indices, "
     "Tensor separate_coo_eids, Tensor separate_coo_relptrs, Tensor(a!) grad_attn_weight, "
     "Tensor attn_score_weight_transposed, Tensor applied_klinear_node_features, "
     "Tensor applied_qlinear_node_features, Tensor attn_score_inner_product, Tensor grad_unnorm_attn_score, "
     "Tensor(b!) grad_k, Tensor(c!) grad_q) -> ()")
def backward_hgt_full_graph_hetero_attention_ops_coo(incsr_row_ptrs, incsr_col_indices, incsr_eids, incsr_reltypes,
                                                     separate_coo_row_indices, separate_coo_col_indices,
                                                     separate_coo_eids, separate_coo_relptrs, grad_attn_weight,
                                                     attn_score_weight_transposed, applied_klinear_node_features,
                                                     applied_qlinear_node_features, attn_score_inner_product,
                                                     grad_unnorm_attn_score, grad_k, grad_q):
    _chk("backward_hgt_full_graph_hetero_attention_ops_coo",
         (grad_attn_weight, attn_score_weight_transposed, applied_klinear_node_features, applied_qlinear_node_features,
          attn_score_inner_product, grad_unnorm_attn_score, grad_k, grad_q),
         (separate_coo_row_indices, separate_coo_col_indices, separate_coo_eids, separate_coo_relptrs))
    R, H, dout, dk = attn_score_weight_transposed.shape
    nq = applied_qlinear_node_features.shape[0]
    gd = _plan.get_grouping(None, separate_coo_col_indices, nq, separate_coo_eids, None)
    gs = _plan.get_grouping(separate_coo_relptrs, separate_coo_row_indices, applied_klinear_node_features.shape[0],
                            separate_coo_col_indices, separate_coo_eids)
    ws = None if gs is None else torch.empty(max(1, gs.num_segments) * H * dout, dtype=torch.float32, device=grad_k.device)
    _call(grad_k, "het_backward_hgt_full_graph_hetero_attention_ops_coo", _p(separate_coo_row_indices),
          _p(separate_coo_col_indices), _p(separate_coo_eids), _p(separate_coo_relptrs), R, separate_coo_eids.numel(),
          _p(grad_attn_weight), _p(attn_score_weight_transposed), _p(applied_klinear_node_features),
          _p(applied_qlinear_node_features), _p(attn_score_inner_product), _p(grad_unnorm_attn_score), _p(grad_k),
          _p(grad_q), H, dk, dout, None if gd is None else gd.handle, None if gs is None else gs.handle, nq, _p(ws),
          0 if ws is None else ws.numel() * 4, _stream(grad_k))


REGISTERED_OPS = tuple(_registered)
