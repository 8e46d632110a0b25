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
               ("het_build_info", "het_last_error", "het_grouping_destroy", "het_grouping_note_stream", "het_grouping_num_segments", "het_grouping_bytes",
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
    was_external = _lib.allocator_is_external()  # (importing het_amd.kernels on a box with a GPU installs torch's allocator)
    try:
        _lib.use_torch_allocator(False)
        assert not _lib.allocator_is_external()
        cb = _lib._ALLOC_FN(lambda n, s, u: None)
        assert L.het_set_allocator(C.cast(cb, C.c_void_p), None, None) != 0 and b"both" in L.het_last_error()
        assert not _lib.allocator_is_external()
    finally:
        if was_external:
            _lib.use_torch_allocator(True)


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


# Output tensors of the reference's launchers: the parameters it declares `at::Tensor&` AND writes, by position, per op
# (hrt/include/DGLHackKernel/OpExport/RGNNOps.inc.h:238-241, 946-953, 83-88, 744-753, 609-618, 1131-1142; RGATOps.inc.h:170-177,
# 465-475, 251-277, 430-460; RGCNOps.inc.h:84-92, 368-380, 24-33, 303-313; HGTOps.inc.h:23-31, 597-608; HGTOpsEdgeParallel.inc.h:33-41,
# 295-307, 95-104, 166-181)
WRITTEN_ARGS = {
    "rgnn_relational_matmul": (4,), "backward_rgnn_relational_matmul": (5, 6),
    "rgnn_relational_matmul_no_scatter_gather_list": (3,), "backward_rgnn_relational_matmul_no_scatter_gather_list": (4, 5),
    "relational_fused_gat_separate_coo": (9, 10, 11), "backward_relational_fused_gat_separate_coo": (13, 14, 15),
    "relational_fused_gat_csr": (9, 10, 11), "backward_relational_fused_gat_csr": (13, 14, 15),
    "rgcn_layer1_separate_coo": (7,), "backward_rgcn_layer1_separate_coo": (7, 8, 10),
    "rgcn_node_mean_aggregation_compact_as_of_node_separate_coo": (7,),
    "backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo": (9,),
    "rgnn_inner_product_right_node_separatecoo": (8,), "backward_inner_product_right_node_separatecoo": (9, 10),
    "hgt_full_graph_edge_softmax_ops_separate_coo": (6, 7, 8),
    "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo": (8, 9, 10),
    "hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo": (7,),
    "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo": (8, 9, 10),
    "hgt_full_graph_hetero_attention_ops_coo": (7, 8), "backward_hgt_full_graph_hetero_attention_ops_coo": (8, 14, 15),
}


def _schemas_in_a_fresh_interpreter(setup: str):
    """{op name: schema string} of torch.ops.torch_hrt after running `setup` in a new interpreter (the two registrations define
    the same names, so they cannot live in one process)."""
    import json
    import subprocess
    import sys
    code = (setup + "\nimport json, torch\nK = torch.ops.torch_hrt\n"
            "print('SCHEMAS' + json.dumps({n: str(getattr(K, n).default._schema) for n in %r}))" % (MUST_EXPORT_OPS,))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("HET_TORCH_HRT_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("SCHEMAS")][-1]
    return json.loads(line[len("SCHEMAS"):])


def test_compiled_and_python_schemas_are_identical():
    """One op name, one schema: libtorch_hrt.so (csrc/torch_export.cpp, loaded with torch.ops.load_library alone) and the Python
    registration (het_amd/kernels.py) expose the same schema string for each of the 26 ops, and every tensor the reference's
    launcher takes as `at::Tensor&` and writes carries an alias annotation (`Tensor(a!)`) -- what functionalization and
    torch.compile read to know that the op mutates its argument."""
    lib = os.path.join(ROOT, "het_amd", "libtorch_hrt.so")
    if not os.path.exists(lib):
        pytest.skip("libtorch_hrt.so not built (make -C het_amd/csrc torch_hrt)")
    compiled = _schemas_in_a_fresh_interpreter("import sys, torch\ntorch.ops.load_library(%r)\nassert 'het_amd' not in sys.modules" % lib)
    python = _schemas_in_a_fresh_interpreter("import het_amd.kernels")
    assert set(compiled) == set(python) == set(MUST_EXPORT_OPS)
    for n in MUST_EXPORT_OPS:
        assert compiled[n] == python[n], f"{n}:\n  compiled {compiled[n]}\n  python   {python[n]}"
    for n, written in WRITTEN_ARGS.items():
        args = torch._C.parse_schema(compiled[n]).arguments
        for i, a in enumerate(args):
            is_mut = a.alias_info is not None and a.alias_info.is_write
            assert is_mut == (i in written), f"{n}: argument {i} ({a.name}) mutable={is_mut}, the reference writes {written}"
    assert set(WRITTEN_ARGS) == set(MUST_EXPORT_OPS) - {"build_debug_info", "transpose_csr", "convert_integrated_csr_to_separate_csr",
                                                       "convert_integrated_csr_to_separate_coo", "convert_integrated_coo_to_separate_csr",
                                                       "convert_integrated_coo_to_separate_coo"}


def test_inferred_schema_of_a_tensor_ref_has_no_alias_info(tmp_path):
    """What a build of the REFERENCE registers: `m.def("name", fn)` with `at::Tensor&` parameters infers `Tensor _0` -- no alias
    annotation, positional placeholder names (OpExport/RGNNOps.inc.h:1188-1206 registers every op that way).  A ten-line extension
    compiled against this torch shows it; our explicit schemas differ from the reference's inferred ones exactly by the parameter
    names and the `(a!)` marks, neither of which a positional call sees."""
    import shutil
    import subprocess
    import sys
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import torch.utils.cpp_extension as ce
    src = tmp_path / "t.cpp"
    src.write_text('#include <torch/library.h>\n#include <ATen/ATen.h>\n'
                   'void f_ref(at::Tensor& a, at::Tensor& b, int64_t k, bool f, double s, torch::Dict<std::string, at::Tensor> d) { a.add_(1); }\n'
                   'TORCH_LIBRARY_FRAGMENT(het_schema_probe, m) { m.def("f_ref", f_ref); }\n')
    out = tmp_path / "libprobe.so"
    libdir = ce.library_paths()[0]
    cmd = (["g++", "-O0", "-std=c++17", "-fPIC", "-shared", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
           + ["-I" + p for p in ce.include_paths()] + [str(src), "-o", str(out), "-L" + libdir, "-lc10", "-ltorch_cpu", "-ltorch",
                                                       "-Wl,-rpath," + libdir])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        pytest.skip("probe extension did not compile: " + r.stderr[-300:])
    r = subprocess.run([sys.executable, "-c", "import torch; torch.ops.load_library(%r); "
                        "print(torch.ops.het_schema_probe.f_ref.default._schema)" % str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.strip() == "het_schema_probe::f_ref(Tensor _0, Tensor _1, int _2, bool _3, float _4, Dict(str, Tensor) _5) -> ()"
