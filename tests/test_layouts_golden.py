"""Layout builders (het_amd.graph) against golden vectors produced by the
reference's own importable Python (tests/golden/make_golden.py).  Exact."""
import pytest
import torch

from het_amd import graph as G
from het_amd.synth import IntegratedCOO


def _coo(gold):
    n, r = int(gold["num_nodes"]), int(gold["num_rels"])
    return IntegratedCOO(n, r, torch.tensor([0, n]), gold["row"], gold["col"], gold["rel"], gold["eids"])


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_separate_coo_and_unique_lists(which, golden_toy, golden_mag):
    gold = golden_toy if which == "toy" else golden_mag
    coo = _coo(gold)
    rp, r, c, e = G.integrated_coo_to_separate_coo(coo.row, coo.col, coo.rel, coo.eids, coo.num_rels)
    assert torch.equal(rp, gold["sep_rel_ptrs"])
    assert torch.equal(r, gold["sep_row"]) and torch.equal(c, gold["sep_col"]) and torch.equal(e, gold["sep_eids"])

    g = G.HetGraph.from_integrated_coo(coo)
    s = g.get_separate_coo_original()
    assert torch.equal(s["eids"], torch.arange(coo.num_edges))  # canonicalised
    assert torch.equal(s["row_indices"], gold["sep_row"]) and torch.equal(s["col_indices"], gold["sep_col"])
    ss = g.get_separate_unique_node_indices_single_sided()
    ssi = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    for k in ("node_indices_row", "rel_ptrs_row", "node_indices_col", "rel_ptrs_col"):
        assert torch.equal(ss[k], gold["ss_" + k]), k
    for k in ("inverse_indices_row", "inverse_indices_col"):
        assert torch.equal(ssi[k], gold["ss_" + k]), k
    ts = g.get_separate_unique_node_indices()
    tsi = g.get_separate_unique_node_indices_inverse_idx()
    assert torch.equal(ts["node_indices"], gold["ts_node_indices"])
    assert torch.equal(ts["rel_ptrs"], gold["ts_rel_ptrs"])
    assert torch.equal(tsi["inverse_indices"], gold["ts_inverse_indices"])


def _rows_as_sets(ptrs, *cols):
    out = []
    for i in range(ptrs.numel() - 1):
        a, b = int(ptrs[i]), int(ptrs[i + 1])
        out.append(sorted(zip(*[c[a:b].tolist() for c in cols])))
    return out


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_csr_and_transpose(which, golden_toy, golden_mag):
    gold = golden_toy if which == "toy" else golden_mag
    coo = _coo(gold)
    n = int(gold["csr_row_ptrs"].numel() - 1)
    ptr, c, r, e = G.coo_to_csr(coo.row, coo.col, coo.rel, coo.eids, n)
    assert torch.equal(ptr, gold["csr_row_ptrs"])
    # the reference's argsort is unstable: compare rows as multisets
    assert _rows_as_sets(ptr, c, r, e) == _rows_as_sets(gold["csr_row_ptrs"], gold["csr_col"], gold["csr_rel"], gold["csr_eids"])
    tptr, tc, te, tr = G.transpose_csr(gold["csr_row_ptrs"], gold["csr_col"], gold["csr_eids"], gold["csr_rel"])
    m = min(tptr.numel(), gold["tcsr_row_ptrs"].numel())
    assert torch.equal(tptr[:m], gold["tcsr_row_ptrs"][:m])
    assert _rows_as_sets(tptr[:m], tc, te, tr) == _rows_as_sets(gold["tcsr_row_ptrs"][:m], gold["tcsr_col"], gold["tcsr_eids"], gold["tcsr_rel"])
    # transposing twice gives the original row_ptrs (the reference's only unittest,
    # hrt/python/test/test_kernel_correctness.py:8-44)
    p2, _, _, _ = G.transpose_csr(tptr, tc, te, tr)
    assert torch.equal(p2[: ptr.numel()], ptr)
