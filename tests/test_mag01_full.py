"""The whole ogbn_mag_0.1 topology the reference ships (345 172 edges, 6 relations; hrt/data/ogbn_mag_0.1/*_coo_2.npy, loaded
whole by the reference's round-trip test hrt/src/test_hyb.cu.cc:26-36) -- SURVEY.md section 8c, fixture 2.

CPU side: the layout builders of het_amd.graph on CPU tensors against the reference builders' digests, and the oracle against
the reference's float outputs (fused GAT forward exp / sum, CompactAsOfNodeKind 0 and 4; grad_feat_src of its backward), all on
the full edge set.  The GPU side of the same fixture is tests/test_gpu_mag01_full.py."""
import json

import pytest
import torch

from het_amd import graph as G
from het_amd.synth import IntegratedCOO
from oracle import ops as O
from tests.golden import recipe


@pytest.fixture(scope="module")
def full(golden_mag_full):
    return mag01_full_case(golden_mag_full)


def mag01_full_case(gold):
    """Everything the tests derive from the fixture: shuffled integrated COO, digests, regenerated float inputs."""
    row, col, rel, eids, n, R = recipe.integrated_coo(gold["coo"])
    dig = json.loads(str(gold["digests_json"]))
    return {"row": row, "col": col, "rel": rel, "eids": eids, "n": n, "R": R, "dig": dig, "gold": gold}


def check_inputs(case, inp):
    for k, v in inp.items():
        assert recipe.digest(v) == case["dig"]["input_" + k], f"regenerated input {k} differs from the one the reference ran on"


def test_fixture_is_the_whole_shipped_topology(full):
    coo = full["gold"]["coo"]
    assert tuple(coo.shape) == (3, 345172) and coo.dtype == torch.int32
    counts = torch.bincount(coo[2].long(), minlength=6).tolist()
    assert counts == [52670, 52670, 64620, 64620, 55296, 55296]  # cited / citing / has / is-about / writing / written-by
    # the files hold every relation next to its reverse: "citing" is "cited" with the ends swapped, and so on
    for a, b in ((0, 1), (2, 3), (5, 4)):
        ea, eb = coo[:, coo[2] == a], coo[:, coo[2] == b]
        assert torch.equal(ea[0], eb[1]) and torch.equal(ea[1], eb[0])
    assert full["n"] == 73638 and full["R"] == 6


def test_layout_builders_match_reference_digests(full):
    c, dig, gold = full, full["dig"], full["gold"]
    rp, r, co, e = G.integrated_coo_to_separate_coo(c["row"], c["col"], c["rel"], c["eids"], c["R"])
    assert torch.equal(rp, gold["sep_rel_ptrs"])
    for name, t in (("sep_rel_ptrs", rp), ("sep_row", r), ("sep_col", co), ("sep_eids", e)):
        assert recipe.digest(t) == dig[name], name
    g = G.HetGraph.from_integrated_coo(IntegratedCOO(c["n"], c["R"], torch.tensor([0, c["n"]]), c["row"], c["col"], c["rel"], c["eids"]))
    ss = g.get_separate_unique_node_indices_single_sided()
    ssi = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    for k in ("node_indices_row", "rel_ptrs_row", "node_indices_col", "rel_ptrs_col"):
        assert recipe.digest(ss[k]) == dig["ss_" + k], k
    for k in ("inverse_indices_row", "inverse_indices_col"):
        assert recipe.digest(ssi[k]) == dig["ss_" + k], k
    assert torch.equal(ss["rel_ptrs_row"], gold["ss_rel_ptrs_row"]) and torch.equal(ss["rel_ptrs_col"], gold["ss_rel_ptrs_col"])
    ts, tsi = g.get_separate_unique_node_indices(), g.get_separate_unique_node_indices_inverse_idx()
    assert recipe.digest(ts["node_indices"]) == dig["ts_node_indices"]
    assert torch.equal(ts["rel_ptrs"], gold["ts_rel_ptrs"])
    assert recipe.digest(tsi["inverse_indices"]) == dig["ts_inverse_indices"]
    # CSR / transposed CSR: row pointers exactly, rows as multisets (the reference's argsort is not stable)
    ptr, cc, rr, ee = G.coo_to_csr(c["row"], c["col"], c["rel"], c["eids"], c["n"])
    assert torch.equal(ptr, gold["csr_row_ptrs"].long())
    for name, t in zip(("col", "rel", "eids"), recipe.canonical_csr(ptr, cc, rr, ee)):
        assert recipe.digest(t) == dig[f"csr_{name}_canonical"], name
    tptr, tc, te, tr = G.transpose_csr(ptr, cc, ee, rr)
    m = gold["tcsr_row_ptrs"].numel()
    assert torch.equal(tptr[:m], gold["tcsr_row_ptrs"].long())
    for name, t in zip(("col", "rel", "eids"), recipe.canonical_csr(tptr[:m], tc, tr, te)):
        assert recipe.digest(t) == dig[f"tcsr_{name}_canonical"], name
    # round trip, as the reference's own test of this data does with its hybrid format (test_hyb.cu.cc) and its unittest does
    # with transpose_csr (hrt/python/test/test_kernel_correctness.py:8-44)
    p2, c2, e2, r2 = G.transpose_csr(tptr, tc, te, tr)
    assert torch.equal(p2[: ptr.numel()], ptr)
    assert all(torch.equal(a, b) for a, b in zip(recipe.canonical_csr(p2[: ptr.numel()], c2, r2, e2), recipe.canonical_csr(ptr, cc, rr, ee)))


def mag01_full_layouts(case):
    g = G.HetGraph.from_integrated_coo(IntegratedCOO(case["n"], case["R"], torch.tensor([0, case["n"]]), case["row"], case["col"],
                                                     case["rel"], case["eids"]))
    return g, g.get_separate_coo_original(), g.get_separate_unique_node_indices_single_sided(), \
        g.get_separate_unique_node_indices_single_sided_inverse_idx()


def test_oracle_gat_forward_backward_on_the_full_edge_set(full):
    gold = full["gold"]
    g, s, ss, ssi = mag01_full_layouts(full)
    n, E, H, D = full["n"], full["row"].numel(), recipe.H, recipe.D
    inp = recipe.gat_inputs(E, n, ss["node_indices_row"].numel(), ss["node_indices_col"].numel())
    check_inputs(full, inp)
    ar = torch.arange(E)
    rp, row, col = s["rel_ptrs"], s["row_indices"], s["col_indices"]
    sm, ex, ret = torch.empty(n, H), torch.empty(E, H), torch.empty(n, H, D)
    O.relational_fused_gat_separate_coo(ar, rp, row, col, 0, {}, torch.zeros(E, H, D), inp["gat_el"], inp["gat_er"], sm, ex, ret, recipe.SLOPE)
    torch.testing.assert_close(ex, gold["gat_exp"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(sm, gold["gat_sum"], rtol=1e-5, atol=1e-6)
    gf, gl, gr = torch.zeros(E, H, D), torch.zeros(E, H), torch.zeros(E, H)
    O.backward_relational_fused_gat_separate_coo(ar, rp, row, col, 0, {}, torch.zeros(E, H, D), inp["gat_el"], inp["gat_er"],
                                                 gold["gat_sum"], gold["gat_exp"], torch.zeros(n, H, D), inp["gatb_gradout"], gf, gl, gr,
                                                 recipe.SLOPE)
    per_node = torch.zeros(n, H, D, dtype=torch.float64).index_add_(0, row, gf.double()).float()
    torch.testing.assert_close(per_node, gold["gatb_grad_feat_src"], rtol=5e-5, atol=5e-6)  # (fp32 sums of hub sources in the reference)
    d = {"edata_idx_to_inverse_idx_row": ssi["inverse_indices_row"], "edata_idx_to_inverse_idx_col": ssi["inverse_indices_col"]}
    sm2, ex2 = torch.empty(n, H), torch.empty(E, H)
    O.relational_fused_gat_separate_coo(ar, rp, row, col, 4, d, torch.zeros(inp["gatc_el"].shape[0], H, D), inp["gatc_el"], inp["gatc_er"],
                                        sm2, ex2, ret, recipe.SLOPE)
    torch.testing.assert_close(ex2, gold["gatc_exp"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(sm2, gold["gatc_sum"], rtol=1e-5, atol=1e-6)


def round5_lists_and_inputs(case, g):
    """(lists, inputs) of tests/util.py::check_round5_gat_pins on the full topology: our builders' layouts (equal to the reference's:
    the digest test above) and the float inputs regenerated from their seeds."""
    s, ss = g.get_separate_coo_original(), g.get_separate_unique_node_indices_single_sided()
    ssi, ts = g.get_separate_unique_node_indices_single_sided_inverse_idx(), g.get_separate_unique_node_indices_inverse_idx()
    tsn = g.get_separate_unique_node_indices()
    lists = {"sep_rel_ptrs": s["rel_ptrs"], "sep_row": s["row_indices"], "sep_col": s["col_indices"], "ts_rel_ptrs": ts["rel_ptrs"],
             "ts_node_indices": tsn["node_indices"], "ts_inverse_indices": ts["inverse_indices"],
             "ss_inverse_indices_row": ssi["inverse_indices_row"], "ss_inverse_indices_col": ssi["inverse_indices_col"],
             "ss_node_indices_row": ss["node_indices_row"]}
    E, n = case["row"].numel(), case["n"]
    inp = recipe.gat_inputs(E, n, ss["node_indices_row"].numel(), ss["node_indices_col"].numel())
    inp5 = recipe.gat_inputs_round5(tsn["node_indices"].numel())
    check_inputs(case, inp)
    check_inputs(case, inp5)
    inp.update(inp5)
    return lists, inp


def test_oracle_round5_pins_on_the_full_edge_set(full):
    """Kinds 1 / 2 forward, kinds 4 / 1 backward grad_feat of the fused GAT pair (round-5 fixtures) on all 345 172 edges."""
    from tests.util import check_round5_gat_pins
    g = mag01_full_layouts(full)[0]
    lists, inp = round5_lists_and_inputs(full, g)
    check_round5_gat_pins(O, "cpu", full["gold"], lists, inp, recipe.SLOPE)
