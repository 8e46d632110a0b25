"""Mini-batch path: the neighbour sampler's block structure (CPU) and, on the GPU, layers on sampled blocks against
the full-graph layers (with fan-out >= every in-degree the block holds all in-edges of its destinations, so the
destination rows must match the full-graph result)."""
import pytest
import torch

from het_amd.graph import HetGraph
from het_amd.sampling import NeighborSampler, run_blocks
from het_amd.synth import make_random


def _graph(dev="cpu"):
    coo = make_random(400, 4, 6000, seed=13)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    return coo, HetGraph.from_integrated_coo(coo, full=True)


def test_blocks_structure_cpu():
    coo, g = _graph()
    s = NeighborSampler(g, [3, 5], seed=1)
    seeds = torch.tensor([5, 17, 99, 250, 3])
    blocks = s.sample_blocks(seeds)
    assert len(blocks) == 2 and torch.equal(blocks[-1].nodes[: blocks[-1].num_dst], seeds)
    assert torch.equal(blocks[0].nodes[: blocks[0].num_dst], blocks[1].nodes)
    true_edges = set(zip(coo.row.tolist(), coo.col.tolist(), coo.rel.tolist()))
    for b, fan in zip(blocks, [3, 5]):
        sc = b.graph.get_separate_coo_original()
        assert torch.equal(sc["eids"], torch.arange(sc["eids"].numel()))
        src, dst = b.nodes[sc["row_indices"]], b.nodes[sc["col_indices"]]
        rel = torch.repeat_interleave(torch.arange(4), sc["rel_ptrs"][1:] - sc["rel_ptrs"][:-1])
        assert all((a, c, r) in true_edges for a, c, r in zip(src.tolist(), dst.tolist(), rel.tolist()))
        assert int(sc["col_indices"].max()) < b.num_dst                      # edges end in destination nodes
        indeg_block = torch.bincount(sc["col_indices"], minlength=b.num_dst)
        indeg_full = torch.bincount(coo.col, minlength=coo.num_nodes)[b.nodes[: b.num_dst]]
        assert torch.equal(indeg_block, torch.minimum(indeg_full, torch.tensor(fan)))  # min(deg, fanout) per destination
        # global edge ids point at the same (src, dst) pairs
        assert torch.equal(coo.row[b.edge_ids], src) and torch.equal(coo.col[b.edge_ids], dst)
    again = NeighborSampler(g, [3, 5], seed=1).sample_blocks(seeds)
    assert all(torch.equal(x.edge_ids, y.edge_ids) for x, y in zip(blocks, again))  # seeded


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True])
def test_rgat_on_full_fanout_blocks_matches_full_graph(compact):
    from het_amd.layers import HET_RGATLayer
    coo, g = _graph("cuda")
    torch.manual_seed(3)
    flags = dict(compact_as_of_node_flag=compact, compact_direct_indexing_flag=compact, self_loop=True, dropout=0.0)
    layers = torch.nn.ModuleList([HET_RGATLayer(64, 64, 4, 4, activation=torch.relu, **flags),
                                  HET_RGATLayer(64, 64, 4, 1, **flags)]).cuda()
    x = torch.randn(coo.num_nodes, 64, device="cuda", requires_grad=True)
    full = x
    for layer in layers:
        full = layer(g, full)
    seeds = torch.tensor([7, 300, 42, 9, 111, 250], device="cuda")
    go = torch.randn(seeds.numel(), 64, device="cuda")
    full[seeds].backward(go)
    gx_full, gw_full = x.grad.clone(), layers[0].conv_weights.grad.clone()
    x.grad = None
    layers.zero_grad()
    blocks = NeighborSampler(g, [-1, -1]).sample_blocks(seeds)
    out = run_blocks(layers, blocks, x[blocks[0].nodes])
    out.backward(go)
    torch.testing.assert_close(out, full[seeds].detach(), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(x.grad, gx_full, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(layers[0].conv_weights.grad, gw_full, rtol=2e-4, atol=1e-4)


def _oracle_rgat_on_blocks(layers, blocks, x64):
    """oracle/layers.py evaluated on every block's own sub-graph (fp64): layer l on block l, destination rows kept -- what
    the reference's mini-batch loop computes block by block (hrt/python/RGNNUtils/RGNNUtils.py:164-196 feeds DGL blocks
    converted by hrt/python/utils/mydglgraph_converters.py:18-71 to the same layer code)."""
    from oracle import layers as OL
    params = []
    h = x64
    for layer, b in zip(layers, blocks):
        p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in layer.named_parameters()}
        params.append(p)
        s = {k: v.cpu() for k, v in b.graph.get_separate_coo_original().items()}
        n = b.graph.get_num_nodes()
        # the self-loop and the bias apply to destination rows; the oracle computes them for every row and the first
        # num_dst rows are kept (rows of source-only nodes aggregate nothing and are dropped)
        out = OL.rgat_layer(h, p["conv_weights"], p["attn_l"], p["attn_r"], s["rel_ptrs"], s["row_indices"], s["col_indices"],
                            n, layer.leaky_relu_slope, p.get("loop_weight"), p.get("h_bias"))[: b.num_dst]
        h = layer.activation(out) if layer.activation else out
    return h, params


@pytest.mark.gpu
@pytest.mark.parametrize("one_shot", [False, True])
@pytest.mark.parametrize("compact", [False, True])
def test_rgat_on_sampled_blocks_matches_the_oracle_on_the_blocks(compact, one_shot):
    """Fan-out-limited blocks (the sampled neighbourhood is NOT the full one): the HIP layers on the blocks against the fp64
    oracle on the same block sub-graphs -- values, input gradient, weight gradients of both layers -- with the ops'
    groupings (plans) and, as a training step on small blocks runs, without them (sampling.one_shot_graphs)."""
    import contextlib
    from het_amd.layers import HET_RGATLayer
    from het_amd.sampling import one_shot_graphs
    from tests.util import assert_close
    coo, g = _graph("cuda")
    torch.manual_seed(6)
    flags = dict(compact_as_of_node_flag=compact, compact_direct_indexing_flag=compact, self_loop=True, dropout=0.0)
    layers = torch.nn.ModuleList([HET_RGATLayer(64, 64, 4, 4, activation=torch.relu, **flags),
                                  HET_RGATLayer(64, 64, 4, 1, **flags)]).cuda()
    x = torch.randn(coo.num_nodes, 64, device="cuda", requires_grad=True)
    seeds = torch.tensor([7, 300, 42, 9, 111, 250, 18, 77], device="cuda")
    blocks = NeighborSampler(g, [4, 6], seed=3).sample_blocks(seeds)
    indeg = torch.bincount(coo.col, minlength=coo.num_nodes)
    assert int(indeg[blocks[-1].nodes[: blocks[-1].num_dst]].max()) > 6  # the fan-out really truncates neighbourhoods
    go = torch.randn(seeds.numel(), 64, device="cuda")
    with (one_shot_graphs(blocks) if one_shot else contextlib.nullcontext()):
        out = run_blocks(layers, blocks, x[blocks[0].nodes])
        out.backward(go)
    x64 = x.detach().double().cpu()[blocks[0].nodes.cpu()].requires_grad_(True)
    ref, params = _oracle_rgat_on_blocks(layers, blocks, x64)
    ref.backward(go.double().cpu())
    assert_close(out, ref, what="out")
    gx = torch.zeros(coo.num_nodes, 64, dtype=torch.float64).index_add_(0, blocks[0].nodes.cpu(), x64.grad)
    assert_close(x.grad, gx, what="grad_x")
    for layer, p in zip(layers, params):
        for name in ("conv_weights", "attn_l", "attn_r", "loop_weight", "h_bias"):
            assert_close(getattr(layer, name).grad, p[name].grad, what="grad_" + name)


@pytest.mark.gpu
def test_rgcn_on_sampled_blocks_and_driver(tmp_path):
    from het_amd import train
    from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
    coo, g = _graph("cuda")
    torch.manual_seed(4)
    layer = HET_EglRelGraphConv_EdgeParallel(64, 64, 4).cuda()
    x, norm = torch.randn(coo.num_nodes, 64, device="cuda"), torch.rand(coo.num_edges, 1, device="cuda")
    full = layer(g, x, norm)
    seeds = torch.tensor([1, 2, 3, 399], device="cuda")
    b = NeighborSampler(g, [-1]).sample_blocks(seeds)
    out = run_blocks([layer], b, x[b[0].nodes], norm)
    torch.testing.assert_close(out, full[seeds], rtol=2e-4, atol=2e-5)
    res = train.main(["--model", "rgat", "-d", "mag", "--scale", "0.01", "--n_infeat", "64", "--num_classes", "64",
                      "--num_heads", "4", "--num_layers", "2", "--fanout", "5", "10", "--batch_size", "256",
                      "--n_epochs", "6", "--dropout", "0.0"])
    assert res["minibatch_sample_and_layout_ms"] is not None and res["final_loss"] < 4.3


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True])
def test_hgt_on_type_sorted_blocks_matches_full_graph(compact):
    """HGT's per-node-type linears on sampled blocks: by_type=True orders the seeds by node type and makes every block's
    nodes a few type-sorted runs; two layers on full-fan-out blocks reproduce the full-graph result (values, gradients of
    the input and of a per-type weight) on the seeds."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_HGTLayerHetero
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=4e-4)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).cuda())
    g = HetGraph.from_integrated_coo(coo, full=True)
    T, R, N = g.get_num_ntypes(), g.get_num_rels(), coo.num_nodes
    assert T == 4
    torch.manual_seed(5)
    flags = dict(num_heads=4, dropout=0.0, compact_as_of_node_flag=compact, compact_direct_indexing_flag=compact)
    layers = torch.nn.ModuleList([HET_HGTLayerHetero(T, R, 64, 64, **flags), HET_HGTLayerHetero(T, R, 64, 32, **flags)]).cuda()
    x = torch.randn(N, 64, device="cuda", requires_grad=True)
    full = x
    for layer in layers:
        full = layer(g, full)
    gen = torch.Generator(device="cuda").manual_seed(1)
    seeds = torch.randperm(N, device="cuda", generator=gen)[:40]  # all four node types, unordered
    s = NeighborSampler(g, [-1, -1], by_type=True)
    blocks = s.sample_blocks(seeds)
    order = blocks[-1].nodes[: blocks[-1].num_dst]
    assert torch.equal(torch.sort(order).values, torch.sort(seeds).values)
    types = torch.searchsorted(coo.node_type_offsets[1:].contiguous(), order, right=True)
    assert bool((types[1:] >= types[:-1]).all())
    for b in blocks:  # every run of a block holds nodes of its type only
        rt, ro = b.runs
        nt = torch.searchsorted(coo.node_type_offsets[1:].contiguous(), b.nodes, right=True)
        for t, lo, hi in zip(rt.tolist(), ro[:-1].tolist(), ro[1:].tolist()):
            assert bool((nt[lo:hi] == t).all())
    go = torch.randn(order.numel(), 32, device="cuda")
    full[order].backward(go)
    gx_full, gk_full = x.grad.clone(), layers[0].k_linears.grad.clone()
    x.grad = None
    layers.zero_grad()
    out = run_blocks(layers, blocks, x[blocks[0].nodes])
    out.backward(go)
    torch.testing.assert_close(out, full[order].detach(), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(x.grad, gx_full, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(layers[0].k_linears.grad, gk_full, rtol=2e-4, atol=1e-4)
