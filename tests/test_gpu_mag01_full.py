"""GPU side of the whole-ogbn_mag_0.1 fixture (tests/test_mag01_full.py is the CPU side; SURVEY.md section 8c, fixture 2): the
native layout builders bit-exact against the reference builders' digests, the HIP fused GAT pair against the reference's own
float outputs on all 345 172 edges, and the three layers against the fp64 oracle on the only real topology the reference ships
(real degree skew, every relation beside its reverse)."""
import pytest
import torch

from het_amd import graph as G
from het_amd.synth import IntegratedCOO
from tests.golden import recipe
from tests.test_mag01_full import check_inputs, mag01_full_case
from tests.util import cpu

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def full(golden_mag_full):
    return mag01_full_case(golden_mag_full)


@pytest.fixture(scope="module")
def K():
    import het_amd.kernels as k
    return k.K


@pytest.fixture(params=[True, False], ids=["grouped", "atomics"])
def plan_mode(request):
    import het_amd.plan as plan
    old = plan.enabled
    plan.enabled = request.param
    plan.clear()
    yield request.param
    plan.enabled = old
    plan.clear()


def _graph(case, dev):
    return G.HetGraph.from_integrated_coo(IntegratedCOO(case["n"], case["R"], torch.tensor([0, case["n"]]).to(dev), case["row"].to(dev),
                                                        case["col"].to(dev), case["rel"].to(dev), case["eids"].to(dev)))


def test_native_builders_match_reference_digests(full):
    c, dig, gold = full, full["dig"], full["gold"]
    row, col, rel, eids = (c[k].to(DEV) for k in ("row", "col", "rel", "eids"))
    rp, r, co, e = G.integrated_coo_to_separate_coo(row, col, rel, eids, c["R"])
    assert r.is_cuda
    for name, t in (("sep_rel_ptrs", rp), ("sep_row", r), ("sep_col", co), ("sep_eids", e)):
        assert recipe.digest(t) == dig[name], name
    g = _graph(c, DEV)
    ss = g.get_separate_unique_node_indices_single_sided()
    ssi = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    for k in ("node_indices_row", "rel_ptrs_row", "node_indices_col", "rel_ptrs_col"):
        assert ss[k].is_cuda and recipe.digest(ss[k]) == dig["ss_" + k], k
    for k in ("inverse_indices_row", "inverse_indices_col"):
        assert recipe.digest(ssi[k]) == dig["ss_" + k], k
    ts, tsi = g.get_separate_unique_node_indices(), g.get_separate_unique_node_indices_inverse_idx()
    assert recipe.digest(ts["node_indices"]) == dig["ts_node_indices"] and recipe.digest(ts["rel_ptrs"]) == dig["ts_rel_ptrs"]
    assert recipe.digest(tsi["inverse_indices"]) == dig["ts_inverse_indices"]
    ptr, cc, rr, ee = G.coo_to_csr(row, col, rel, eids, c["n"])
    assert ptr.is_cuda and recipe.digest(ptr) == dig["csr_row_ptrs"]
    for name, t in zip(("col", "rel", "eids"), recipe.canonical_csr(cpu(ptr), cpu(cc), cpu(rr), cpu(ee))):
        assert recipe.digest(t) == dig[f"csr_{name}_canonical"], name
    tptr, tc, te, tr = G.transpose_csr(ptr, cc, ee, rr)
    m = gold["tcsr_row_ptrs"].numel()
    assert recipe.digest(tptr[:m]) == dig["tcsr_row_ptrs"]
    for name, t in zip(("col", "rel", "eids"), recipe.canonical_csr(cpu(tptr[:m]), cpu(tc), cpu(tr), cpu(te))):
        assert recipe.digest(t) == dig[f"tcsr_{name}_canonical"], name


def test_gat_forward_backward_against_the_reference_outputs(K, plan_mode, full):
    """a4 (CompactAsOfNodeKind 0 and 4) exp / sum and a5 grad_feat_src on the full edge set against what the reference's
    ref_rgat.py produced on the same (regenerated) inputs."""
    gold = full["gold"]
    g = _graph(full, "cpu")
    s, ss = g.get_separate_coo_original(), g.get_separate_unique_node_indices_single_sided()
    ssi = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    n, E, H, D = full["n"], full["row"].numel(), recipe.H, recipe.D
    inp = recipe.gat_inputs(E, n, ss["node_indices_row"].numel(), ss["node_indices_col"].numel())
    check_inputs(full, inp)
    ar = torch.arange(E, device=DEV)
    rp, row, col = s["rel_ptrs"].to(DEV), s["row_indices"].to(DEV), s["col_indices"].to(DEV)
    feat = torch.randn(E, H, D, device=DEV)
    sm, ex, ret = torch.empty(n, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(n, H, D, device=DEV)
    K.relational_fused_gat_separate_coo(ar, rp, row, col, 0, {}, feat, inp["gat_el"].to(DEV), inp["gat_er"].to(DEV), sm, ex, ret, recipe.SLOPE)
    torch.testing.assert_close(cpu(ex), gold["gat_exp"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cpu(sm), gold["gat_sum"], rtol=1e-5, atol=1e-5)
    gf, gl, gr = torch.zeros(E, H, D, device=DEV), torch.zeros(E, H, device=DEV), torch.zeros(E, H, device=DEV)
    K.backward_relational_fused_gat_separate_coo(ar, rp, row, col, 0, {}, feat, inp["gat_el"].to(DEV), inp["gat_er"].to(DEV),
                                                 gold["gat_sum"].to(DEV), gold["gat_exp"].to(DEV), ret, inp["gatb_gradout"].to(DEV), gf, gl, gr,
                                                 recipe.SLOPE)
    per_node = torch.zeros(n, H, D, dtype=torch.float64).index_add_(0, s["row_indices"], cpu(gf).double()).float()
    torch.testing.assert_close(per_node, gold["gatb_grad_feat_src"], rtol=5e-5, atol=5e-6)
    d = {"edata_idx_to_inverse_idx_row": ssi["inverse_indices_row"].to(DEV), "edata_idx_to_inverse_idx_col": ssi["inverse_indices_col"].to(DEV)}
    featc = torch.randn(inp["gatc_el"].shape[0], H, D, device=DEV)
    sm2, ex2 = torch.empty(n, H, device=DEV), torch.empty(E, H, device=DEV)
    K.relational_fused_gat_separate_coo(ar, rp, row, col, 4, d, featc, inp["gatc_el"].to(DEV), inp["gatc_er"].to(DEV), sm2, ex2, ret, recipe.SLOPE)
    torch.testing.assert_close(cpu(ex2), gold["gatc_exp"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cpu(sm2), gold["gatc_sum"], rtol=1e-5, atol=1e-5)


def test_gat_round5_pins_against_the_reference_outputs(K, plan_mode, full):
    """Kinds 1 / 2 forward and kinds 4 / 1 backward grad_feat on the full edge set against the reference's ref_rgat.py (round 5)."""
    from tests.test_mag01_full import round5_lists_and_inputs
    from tests.util import check_round5_gat_pins
    lists, inp = round5_lists_and_inputs(full, _graph(full, "cpu"))
    check_round5_gat_pins(K, DEV, full["gold"], lists, inp, recipe.SLOPE, rtol_exp=1e-5, atol_exp=1e-6, rtol_sum=1e-5, atol_sum=1e-5)


def _typed_graph(full):
    row, col, rel, off = recipe.typed_coo(full["gold"]["coo"])
    return G.HetGraph.from_integrated_coo(IntegratedCOO(int(off[-1]), 6, off, row, col, rel, torch.arange(row.numel())))


@pytest.mark.parametrize("compact,mulfirst", [(False, False), (True, False), (True, True)])
def test_rgat_layer_on_the_shipped_topology(full, compact, mulfirst):
    """HET_RGATLayer (feat 64, 4 heads; the one-node dataflow with hubs, run sums and the node-major input gradient) against the
    fp64 oracle layer on the shipped graph, on its one-id-space form (as test_hyb.cu.cc builds it) and typed."""
    from tests.test_gpu_layers import _run_rgat
    _run_rgat(_graph(full, "cpu"), H=4, K=64, X=64, compact=compact, direct=compact, mulfirst=mulfirst, oracle_dev=DEV)
    _run_rgat(_typed_graph(full), H=4, K=64, X=64, compact=compact, direct=compact, mulfirst=mulfirst, seed=3, oracle_dev=DEV)


@pytest.mark.parametrize("compact", [False, True])
def test_rgcn_layer_on_the_shipped_topology(full, compact):
    from tests.test_gpu_layers import _run_rgcn
    _run_rgcn(_graph(full, "cpu"), compact, compact, 64, 64, 6, oracle_dev=DEV)


@pytest.mark.parametrize("H,fused_attn", [(8, False), (1, True)])
def test_hgt_layer_on_the_shipped_topology(full, H, fused_attn, monkeypatch):
    """HET_HGTLayerHetero (feat 64; BASELINE.json configs[3]'s 8 heads and the reference sweep's 1) on the typed view -- canonical
    edge types, three node types -- through the distinct-row kernels, against the fp64 oracle."""
    from tests.test_gpu_layers import _run_hgt_fused
    _run_hgt_fused(fused_attn, True, H, 64, 64, monkeypatch, g=_typed_graph(full), oracle_dev=DEV)
