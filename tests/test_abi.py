"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/het_amd.h declares, and the torch_hrt namespace carries the reference's op names."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUST_EXPORT_OPS = [  # the 20 compute + 5 layout + 1 info ops
    "build_debug_info", "transpose_csr", "convert_integrated_csr_to_separate_csr", "convert_integrated_csr_to_separate_coo",
    "convert_integrated_coo_to_separate_csr", "convert_integrated_coo_to_separate_coo",
    "rgnn_relational_matmul", "backward_rgnn_relational_matmul", "rgnn_relational_matmul_no_scatter_gather_list",
    "backward_rgnn_relational_matmul_no_scatter_gather_list", "rgcn_layer1_separate_coo", "backward_rgcn_layer1_separate_coo",
    "rgcn_node_mean_aggregation_compact_as_of_node_separate_coo",
    "backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo", "relational_fused_gat_separate_coo",
    "backward_relational_fused_gat_separate_coo", "relational_fused_gat_csr", "backward_relational_fused_gat_csr",
    "hgt_full_graph_edge_softmax_ops_separate_coo", "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo",
    "hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo",
    "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo",
    "hgt_full_graph_hetero_attention_ops_coo", "backward_hgt_full_graph_hetero_attention_ops_coo",
    "rgnn_inner_product_right_node_separatecoo", "backward_inner_product_right_node_separatecoo",
]


def _declared():
    src = open(os.path.join(ROOT, "include", "het_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(het_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from het_amd import _lib
    names = _declared()
    assert len(names) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # every declared compute entry point has a ctypes signature (so the Python side calls it typed)
    untyped = [n for n in names if n not in _lib._SIGNATURES and n not in
               ("het_build_info", "het_last_error", "het_grouping_destroy", "het_grouping_num_segments", "het_grouping_bytes",
                "het_kernel_timing_enable", "het_kernel_timing_read", "het_rgat_backward_compact_workspace",
                "het_hgt_backward_compact_workspace", "het_hgt_compact_shape_ok", "het_rgat_node_gemm_ok", "het_rgat_aggregate_compact_workspace", "het_hgt_aggregate_compact_workspace",
                "het_rgat_aggregate_compact_runs_workspace", "het_rgat_backward_compact_runs_workspace",
                "het_set_allocator", "het_allocator_is_external", "het_node_rows_matmul_sum_ok", "het_rgcn_layer_ok",
                "het_rgcn_layer_backward_workspace")]
    assert not untyped, untyped
    assert "gfx950" in _lib.build_info()


def test_allocator_hook_without_gpu():
    """het_set_allocator takes both functions or neither; the default is hipMalloc (nothing installed)."""
    import ctypes as C
    from het_amd import _lib
    L = _lib.lib()
    assert not _lib.allocator_is_external()
    cb = _lib._ALLOC_FN(lambda n, s, u: None)
    assert L.het_set_allocator(C.cast(cb, C.c_void_p), None, None) != 0 and b"both" in L.het_last_error()
    assert not _lib.allocator_is_external()


def test_kernel_timing_api_without_gpu():
    """The per-kernel timing switch works with no device: nothing recorded, zero launches."""
    from het_amd import _lib
    _lib.kernel_timing(True)
    ms, n = _lib.kernel_timing_read("HET_")
    assert ms == 0.0 and n == 0
    _lib.kernel_timing(False)


def test_torch_hrt_namespace_has_reference_op_names():
    import het_amd.kernels as k
    for name in MUST_EXPORT_OPS:
        assert hasattr(k.K, name), name
        assert name in k.REGISTERED_OPS


def test_argument_validation_without_gpu():
    """Host-side validation runs before any launch: bad sizes / null pointers give an error code and a
    message instead of a kernel fault (the reference only has compiled-out asserts)."""
    from het_amd import _lib
    L = _lib.lib()
    rc = L.het_rgnn_relational_matmul(0, None, 4, None, None, 10, None, None, None, 4, 64, 16, 1, None, None, 0, None)
    assert rc == 1 and b"null" in L.het_last_error()
    rc = L.het_rgnn_relational_matmul(7, None, 4, None, None, 10, None, None, None, 4, 64, 16, 1, None, None, 0, None)
    assert rc == 3
    rc = L.het_relational_fused_gat_separate_coo(None, None, None, None, 4, 10, 5, 2, None, None, None, None, None, None,
                                                 None, None, None, None, None, 4, 16, 0.2, None, None, None, None)
    assert rc != 0


def test_compute_ops_reject_cpu_tensors():
    import het_amd.kernels as k
    from het_amd._lib import HetError
    e = torch.zeros(1, dtype=torch.int64)
    with pytest.raises((HetError, RuntimeError)):
        k.K.rgcn_layer1_separate_coo(torch.tensor([0, 1]), e, e, e, torch.randn(2, 4), torch.randn(1, 4, 4), torch.rand(1),
                                     torch.zeros(2, 4))


def test_layout_ops_run_on_cpu(golden_toy):
    import het_amd.kernels as k
    g = golden_toy
    rp, r, c, e = k.K.convert_integrated_coo_to_separate_coo(g["row"], g["col"], g["rel"], g["eids"], 4, 2)
    assert torch.equal(rp, g["sep_rel_ptrs"]) and torch.equal(r, g["sep_row"]) and torch.equal(e, g["sep_eids"])
    ptr, col, eid, rel = k.K.transpose_csr(g["csr_row_ptrs"], g["csr_col"], g["csr_eids"], g["csr_rel"])
    assert torch.equal(ptr, g["tcsr_row_ptrs"])


def test_compiled_registration_object_lists_every_op():
    """libtorch_hrt.so (csrc/torch_export.cpp) loaded the reference's way -- torch.ops.load_library in an interpreter that never
    imports het_amd -- registers every op the Python registration defines (no GPU needed to load it)."""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "het_amd", "libtorch_hrt.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("libtorch_hrt.so not built (make -C het_amd/csrc torch_hrt)")
    import het_amd.kernels as k
    code = ("import sys, torch; torch.ops.load_library(%r); K = torch.ops.torch_hrt; assert 'het_amd' not in sys.modules; "
            "print(' '.join(n for n in %r if not hasattr(K, n)))" % (lib, list(k.REGISTERED_OPS)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == "", "ops missing from libtorch_hrt.so: " + r.stdout


def test_hgt_unfused_csr_path_is_a_named_error():
    """HGT/models.py:271 reaches B.hgt_full_graph_edge_softmax_and_message_mean_aggregation_csr when
    --fused_message_mean_aggregation_flag is off: het_amd.backend names it and says what to use (no AttributeError)."""
    import pytest
    import het_amd.backend as B
    from het_amd._lib import HetUnsupported
    from het_amd.layers import HET_HGTLayerHetero
    with pytest.raises(HetUnsupported, match="fused_message_mean_aggregation_flag"):
        B.hgt_full_graph_edge_softmax_and_message_mean_aggregation_csr(None, None, None, None)
    with pytest.raises(NotImplementedError):
        HET_HGTLayerHetero(2, 3, 16, 16, num_heads=2, fused_message_mean_aggregation_flag=False)
