"""ctypes binding of libhet_amd.so (the C ABI declared in include/het_amd.h).

The library is built in-tree by ``het_amd/csrc/Makefile`` (``__graft_entry__.build``)
and must be present: there is no CPU or PyTorch fallback behind the ops.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HET_AMD_LIB") or os.path.join(_HERE, "libhet_amd.so")  # override: A/B of experiment builds

P, I64, INT, DBL = C.c_void_p, C.c_int64, C.c_int, C.c_double

# name -> argtypes, in the order of include/het_amd.h
_SIGNATURES = {
    "het_grouping_create": [P, I64, P, I64, I64, P, P, P, C.POINTER(P)],
    "het_grouping_rank_of_position": [P, P, P],
    "het_grouping_segment_map": [P, I64, P, P],
    "het_grouping_gather_payload1": [P, P, I64, P, P],
    "het_rows_add_bias": [P, P, P, P, I64, I64, P],
    "het_rows_gather": [P, P, I64, I64, P, P],
    "het_rows_scatter_add": [P, P, I64, I64, P, P],
    "het_rows_scatter_add_grouped": [P, P, I64, P, I64, P],
    "het_layout_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P],
    "het_layout_coo_to_csr": [P, P, P, P, I64, I64, P, P, P, P, P],
    "het_layout_transpose_csr": [P, P, P, P, I64, I64, I64, P, P, P, P, P],
    "het_layout_unique_rel_nodes": [P, I64, P, P, I64, I64, P, P, P, P, P],
    "het_rgnn_relational_matmul": [I64, P, I64, P, P, I64, P, P, P, I64, I64, I64, INT, P, P, I64, P],
    "het_backward_rgnn_relational_matmul_attn_dot_only": [P, I64, P, P, I64, I64, P, P, P, P, P, P, I64, I64, I64, INT, P, P, I64, P, P, P],
    "het_rgnn_relational_matmul_attn_dot": [I64, P, I64, P, P, I64, P, P, P, P, P, I64, I64, I64, P, P, I64, P, P, P],
    "het_backward_rgnn_relational_matmul": [I64, P, I64, P, P, I64, I64, P, P, P, P, P, I64, I64, I64, INT, INT, P, P, I64, P],
    "het_rgnn_relational_matmul_no_scatter_gather_list": [P, I64, I64, P, P, P, I64, I64, I64, INT, P],
    "het_backward_rgnn_relational_matmul_no_scatter_gather_list": [P, I64, I64, P, P, P, P, P, I64, I64, I64, INT, INT, P],
    "het_relational_fused_gat_separate_coo": [P, P, P, P, I64, I64, I64, I64, P, P, P, P, P, P, P, P, P, P, P, I64, I64, DBL, P, P, P, P],
    "het_backward_relational_fused_gat_separate_coo": [P, P, P, P, I64, I64, I64, I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, DBL, P, P, P, I64, I64, P, I64, P, P, P, P, P],
    "het_relational_fused_gat_csr": [P, P, P, P, I64, I64, P, P, I64, P, P, P, P, P, P, I64, I64, DBL, INT, P],
    "het_backward_relational_fused_gat_csr": [P, P, P, P, I64, I64, P, P, I64, P, P, P, P, P, P, P, P, P, P, I64, I64, DBL, INT, P],
    "het_hgt_fold_source_weights": [P, P, P, P, P, P, I64, I64, I64, I64, I64, INT, P, P],
    "het_hgt_fold_source_weights_backward": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, INT, P, P, P, P, P, P],
    "het_rgat_aggregate_compact": [P, P, P, P, P, P, I64, I64, I64, DBL, P, I64, P, I64, P],
    "het_rows_matmul_backward_dx": [P, I64, P, P, I64, P, P, P, I64, I64, I64, INT, P],
    "het_rows_matmul_backward_dw": [P, I64, P, P, I64, P, P, P, I64, I64, I64, INT, P],
    "het_rows_matmul_backward_dw_colsum": [P, I64, P, P, I64, P, P, P, P, I64, I64, I64, INT, P],
    "het_rows_linear_bias": [P, P, P, P, P, I64, I64, I64, P],
    "het_node_row_map": [P, I64, P, I64, I64, P, P],
    "het_rgat_node_backward_dx": [I64, I64, I64, I64, I64, P, P, P, P, P, P, P, P, P, I64, I64, I64, P, P],
    "het_node_rows_matmul_sum": [I64, I64, I64, I64, P, P, P, P, P, P, I64, I64, P, P],
    "het_node_rows_matmul_sum_bias": [I64, I64, I64, I64, P, P, P, P, P, P, P, I64, I64, P, P],
    "het_rgcn_layer_forward": [P, I64, I64, P, P, P, P, P, P, P, P, P, I64, I64, P],
    "het_rgcn_layer_backward": [P, P, I64, I64, I64, P, P, P, P, P, P, P, P, P, P, I64, I64, P, I64, P],
    "het_rgat_backward_compact": [P, P, P, P, P, P, P, P, P, P, P, P, P, I64, P, I64, I64, I64, I64, I64, I64, DBL, P, I64, P],
    "het_rgat_aggregate_compact_runs": [P, P, I64, P, P, P, P, P, I64, I64, I64, DBL, P, I64, P, P, P, I64, P, P, P, I64, P],
    "het_rgat_backward_compact_runs": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, P, I64, I64, I64, I64, I64, I64, DBL, P, P, I64, P],
    "het_hgt_aggregate_compact": [P, P, P, P, P, I64, I64, I64, I64, P, I64, P],
    "het_hgt_backward_compact": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, P, I64, P],
    "het_rgcn_layer1_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, I64, I64, P, P, I64, P],
    "het_backward_rgcn_layer1_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P, P, P, I64, I64, P, P, I64, P],
    "het_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P, I64, INT, P, P],
    "het_backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P, P, P, I64, INT, P, I64, P],
    "het_hgt_full_graph_edge_softmax_ops_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P, I64, P, P],
    "het_backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P, P, P, I64, P, P],
    "het_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, I64, I64, I64, P, P, I64, P],
    "het_backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo": [P, P, P, P, I64, I64, I64, P, P, P, P, P, P, P, P, I64, I64, I64, P, P, I64, P],
    "het_rgnn_inner_product_right_node_separatecoo": [I64, P, P, P, P, P, P, I64, I64, P, P, P, I64, I64, P],
    "het_backward_inner_product_right_node_separatecoo": [I64, P, P, P, P, P, P, I64, I64, P, P, P, P, P, I64, I64, INT, P, P, I64, I64, P],
    "het_hgt_full_graph_hetero_attention_ops_coo": [P, P, P, P, I64, I64, P, P, P, P, P, I64, I64, I64, P],
    "het_backward_hgt_full_graph_hetero_attention_ops_coo": [P, P, P, P, I64, I64, P, P, P, P, P, P, P, P, I64, I64, I64, P, P, I64, P, I64, P],
}


class HetError(RuntimeError):
    pass


class HetUnsupported(HetError, NotImplementedError):
    """A configuration of the reference's model code that this build names but does not implement (the message says which
    reference op it would need and what to use instead) -- raised where the reference's name is reached, instead of an
    AttributeError from a missing function."""


_lib = None


def lib() -> C.CDLL:
    """Load libhet_amd.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HetError(
                f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C het_amd/csrc). het_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.het_build_info.restype = C.c_char_p
        L.het_last_error.restype = C.c_char_p
        L.het_grouping_destroy.argtypes = [P]
        L.het_grouping_destroy.restype = None
        if hasattr(L, "het_grouping_note_stream"):  # (absent from a library of round 4 and before: exp/ A/B runs through HET_AMD_LIB)
            L.het_grouping_note_stream.argtypes = [P, P]
            L.het_grouping_note_stream.restype = None
        L.het_grouping_num_segments.argtypes = [P]
        L.het_grouping_num_segments.restype = I64
        L.het_grouping_bytes.argtypes = [P]
        L.het_grouping_bytes.restype = I64
        L.het_rgat_aggregate_compact_workspace.argtypes = [P, I64, I64]
        L.het_rgat_aggregate_compact_workspace.restype = I64
        L.het_rgat_aggregate_compact_runs_workspace.argtypes = [P, P, I64, I64, I64, P]
        L.het_rgat_aggregate_compact_runs_workspace.restype = I64
        L.het_rgat_backward_compact_runs_workspace.argtypes = [P, I64, I64, I64, I64, INT, INT, P]
        L.het_rgat_backward_compact_runs_workspace.restype = I64
        L.het_rgat_backward_compact_workspace.argtypes = [I64, I64, I64, I64, INT]
        L.het_rgat_backward_compact_workspace.restype = I64
        L.het_hgt_aggregate_compact_workspace.argtypes = [P, I64, I64]
        L.het_hgt_aggregate_compact_workspace.restype = I64
        L.het_hgt_backward_compact_workspace.argtypes = [I64, I64]
        L.het_hgt_backward_compact_workspace.restype = I64
        L.het_hgt_compact_shape_ok.argtypes = [I64, I64]
        L.het_hgt_compact_shape_ok.restype = INT
        L.het_node_rows_matmul_sum_ok.argtypes = [I64, I64, I64]
        L.het_node_rows_matmul_sum_ok.restype = INT
        L.het_rgcn_layer_ok.argtypes = [I64, I64, I64]
        L.het_rgcn_layer_ok.restype = INT
        L.het_rgcn_layer_backward_workspace.argtypes = [I64, I64]
        L.het_rgcn_layer_backward_workspace.restype = I64
        L.het_rgat_node_gemm_ok.argtypes = [I64, I64, I64, I64]
        L.het_rgat_node_gemm_ok.restype = INT
        L.het_set_allocator.argtypes = [P, P, P]
        L.het_set_allocator.restype = INT
        L.het_allocator_is_external.argtypes = []
        L.het_allocator_is_external.restype = INT
        L.het_kernel_timing_enable.argtypes = [INT]
        L.het_kernel_timing_enable.restype = INT
        L.het_kernel_timing_read.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(I64)]
        L.het_kernel_timing_read.restype = INT
        for name, args in _SIGNATURES.items():
            f = getattr(L, name, None)
            if f is None and name in _ROUND5_ENTRY_POINTS and os.environ.get("HET_AMD_LIB"):
                continue  # an OLDER library selected for an A/B run (exp/ab_r05.sh): the callers ask has() first
            if f is None:
                raise HetError(f"{LIB_PATH} does not export {name}: rebuild it (make -C het_amd/csrc)")
            f.argtypes = args
            f.restype = INT
        _lib = L
    return _lib


_ROUND5_ENTRY_POINTS = ("het_hgt_fold_source_weights", "het_hgt_fold_source_weights_backward", "het_rows_matmul_backward_dw_colsum")


def has(name: str) -> bool:
    """Whether the loaded library exports ``name`` (always true for the in-tree build; an older library selected with HET_AMD_LIB for
    a same-box A/B lacks the entry points of later rounds, and the callers then take the path that library knows)."""
    return hasattr(lib(), name)


_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
_FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_alloc_cbs = None  # the installed callbacks (kept alive: the library holds raw function pointers)


def use_torch_allocator(on: bool = True) -> bool:
    """Device memory the library keeps (groupings and their construction scratch: include/het_amd.h het_set_allocator) from
    torch's caching allocator instead of hipMalloc: it then shows up in torch.cuda.memory_allocated / max_memory_allocated, goes
    back to torch's pool when the plan cache evicts a grouping, and no hipFree (a device-wide synchronisation) happens on an
    op's path.  Blocks are bound to the stream the library first uses them on, like any tensor.  ``on=False``: back to
    hipMalloc (what a caller of the bare C ABI gets).  Returns whether an allocator is installed afterwards."""
    global _alloc_cbs
    L = lib()
    if not on:
        L.het_set_allocator(None, None, None)
        return False
    import torch

    def _alloc(nbytes, stream, _user):
        try:
            return int(torch.cuda.caching_allocator_alloc(int(nbytes), stream=int(stream or 0)))
        except BaseException:  # noqa: BLE001 -- out of memory: the C side reports it (NULL); an exception cannot cross the C frames
            return None

    def _free(ptr, _user):
        try:
            torch.cuda.caching_allocator_delete(int(ptr))
        except BaseException:  # noqa: BLE001
            pass

    cbs = (_ALLOC_FN(_alloc), _FREE_FN(_free))
    rc = L.het_set_allocator(C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p), None)
    if rc != 0:
        raise HetError(f"het_set_allocator failed (code {rc}): {L.het_last_error().decode()}")
    # (earlier callbacks stay referenced: pointers handed out before are still released through them)
    _alloc_cbs = (_alloc_cbs or ()) + (cbs,)
    return True


def allocator_is_external() -> bool:
    return bool(lib().het_allocator_is_external())


def call(name: str, *args) -> None:
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise HetError(f"{name} failed (code {rc}): {L.het_last_error().decode()}")


def kernel_timing(on: bool) -> None:
    """Start (clearing earlier records) / stop the library's per-kernel HIP-event timing (include/het_amd.h)."""
    rc = lib().het_kernel_timing_enable(int(bool(on)))
    if rc != 0:
        raise HetError(f"het_kernel_timing_enable failed (code {rc})")


def kernel_timing_read(prefix: str):
    """(total ms, launches) of the recorded kernels whose name starts with ``prefix``; synchronises their events."""
    L = lib()
    ms, n = C.c_double(0.0), C.c_int64(0)
    rc = L.het_kernel_timing_read(prefix.encode(), C.byref(ms), C.byref(n))
    if rc != 0:
        raise HetError(f"het_kernel_timing_read failed (code {rc}): {L.het_last_error().decode()}")
    return ms.value, n.value


def build_info() -> str:
    return lib().het_build_info().decode()
