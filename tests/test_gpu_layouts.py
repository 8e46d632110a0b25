"""Native device-side layout builders (het_amd/csrc/layouts.hip through het_amd.graph on GPU tensors): bit-exact
against the golden vectors of the reference's importable Python builders (tests/golden) and against the CPU
restatement of the same module on random graphs, including the five torch_hrt layout ops."""
import pytest
import torch

from het_amd import graph as G
from het_amd.synth import IntegratedCOO, make_mag_like, make_random

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _coo(gold):
    n, r = int(gold["num_nodes"]), int(gold["num_rels"])
    return IntegratedCOO(n, r, torch.tensor([0, n]), gold["row"], gold["col"], gold["rel"], gold["eids"])


def _to(coo, dev):
    return IntegratedCOO(coo.num_nodes, coo.num_rels, coo.node_type_offsets.to(dev), coo.row.to(dev), coo.col.to(dev),
                         coo.rel.to(dev), coo.eids.to(dev))


def _flat(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "/"))
        elif torch.is_tensor(v):
            out[prefix + k] = v
    return out


def _shuffled(coo, seed):
    p = torch.randperm(coo.num_edges, generator=torch.Generator().manual_seed(seed))
    return IntegratedCOO(coo.num_nodes, coo.num_rels, coo.node_type_offsets, coo.row[p], coo.col[p], coo.rel[p],
                         torch.randperm(coo.num_edges, generator=torch.Generator().manual_seed(seed + 1)))


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_native_builders_match_reference_golden(which, golden_toy, golden_mag):
    gold = golden_toy if which == "toy" else golden_mag
    coo = _to(_coo(gold), DEV)
    rp, r, c, e = G.integrated_coo_to_separate_coo(coo.row, coo.col, coo.rel, coo.eids, coo.num_rels)
    for name, t in (("sep_rel_ptrs", rp), ("sep_row", r), ("sep_col", c), ("sep_eids", e)):
        assert torch.equal(t.cpu(), gold[name]), name
    g = G.HetGraph.from_integrated_coo(coo)
    ss = g.get_separate_unique_node_indices_single_sided()
    ssi = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    for k in ("node_indices_row", "rel_ptrs_row", "node_indices_col", "rel_ptrs_col"):
        assert torch.equal(ss[k].cpu(), gold["ss_" + k]), k
    for k in ("inverse_indices_row", "inverse_indices_col"):
        assert torch.equal(ssi[k].cpu(), gold["ss_" + k]), k
    ts, tsi = g.get_separate_unique_node_indices(), g.get_separate_unique_node_indices_inverse_idx()
    assert torch.equal(ts["node_indices"].cpu(), gold["ts_node_indices"])
    assert torch.equal(ts["rel_ptrs"].cpu(), gold["ts_rel_ptrs"])
    assert torch.equal(tsi["inverse_indices"].cpu(), gold["ts_inverse_indices"])
    n = int(gold["csr_row_ptrs"].numel() - 1)
    ptr, cc, rr, ee = G.coo_to_csr(coo.row, coo.col, coo.rel, coo.eids, n)
    assert torch.equal(ptr.cpu(), gold["csr_row_ptrs"])


@pytest.mark.parametrize("case", ["random", "random_empty_rel", "mag_small", "no_edges"])
def test_native_builders_match_cpu_restatement(case):
    if case == "random":
        coo = _shuffled(make_random(300, 5, 4000, seed=3), 7)
    elif case == "random_empty_rel":
        coo = make_random(200, 6, 1500, seed=4)
        coo.rel[coo.rel == 2] = 3
        coo = _shuffled(coo, 8)
    elif case == "mag_small":
        coo = make_mag_like(scale=2e-3)
    else:
        z = torch.zeros(0, dtype=torch.int64)
        coo = IntegratedCOO(10, 3, torch.tensor([0, 10]), z, z.clone(), z.clone(), z.clone())
    g_cpu = G.HetGraph.from_integrated_coo(coo)
    g_gpu = G.HetGraph.from_integrated_coo(_to(coo, DEV))
    a, b = _flat(g_cpu.graph_data), _flat(g_gpu.graph_data)
    assert sorted(a) == sorted(b)
    for k in a:
        assert torch.equal(a[k], b[k].cpu()), k


def test_layout_ops_on_gpu_tensors():
    import het_amd.kernels as k
    coo = _shuffled(make_random(150, 4, 2000, seed=5), 9)
    d = _to(coo, DEV)
    ref = k.K.convert_integrated_coo_to_separate_coo(coo.row, coo.col, coo.rel, coo.eids, coo.num_nodes, coo.num_rels)
    got = k.K.convert_integrated_coo_to_separate_coo(d.row, d.col, d.rel, d.eids, coo.num_nodes, coo.num_rels)
    for x, y in zip(ref, got):
        assert y.is_cuda and torch.equal(x, y.cpu())
    ptr, c, r, e = G.coo_to_csr(coo.row, coo.col, coo.rel, coo.eids, coo.num_nodes)
    ref = k.K.transpose_csr(ptr, c, e, r)
    got = k.K.transpose_csr(ptr.to(DEV), c.to(DEV), e.to(DEV), r.to(DEV))
    for x, y in zip(ref, got):
        assert y.is_cuda and torch.equal(x, y.cpu())
    for op in ("convert_integrated_csr_to_separate_coo", "convert_integrated_csr_to_separate_csr",
               "convert_integrated_coo_to_separate_csr"):
        args = (ptr, c, r, e) if "csr_to" in op else (coo.row, coo.col, coo.rel, coo.eids)
        tail = () if "csr_to" in op else (coo.num_nodes, coo.num_rels)
        ref = getattr(k.K, op)(*args, *tail)
        got = getattr(k.K, op)(*(t.to(DEV) for t in args), *tail)
        for x, y in zip(ref, got):
            assert torch.equal(x, y.cpu()), op
