"""GPU parity at layer level: the HET layer modules (op compositions of the reference's
model scripts) against the plain-PyTorch fp64 oracle, outputs and all gradients.
Layers run on graphs with canonical eids, as HetGraph and the reference build them; the op tests in
test_gpu_ops.py renumber the eids with a random permutation."""
import pytest
import torch

from oracle import layers as OL
from tests.util import assert_close, mag_graph, random_graph, rgat_min_abs_preactivation

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run_rgat(g, H, K, X, compact, direct, mulfirst, edge_parallel=True, seed=0, attn_scale=1.0, oracle_dev="cpu", **layer_kw):
    """oracle_dev: where the fp64 oracle layer is evaluated (it is plain torch: "cuda" for graphs of 1e5+ edges, where the CPU
    takes most of a minute per case)."""
    from het_amd.layers import HET_RGATLayer
    torch.manual_seed(seed)
    R, N = g.get_num_rels(), g.get_num_nodes()
    layer = HET_RGATLayer(K, X, R, H, bias=True, self_loop=True, compact_as_of_node_flag=compact,
                          compact_direct_indexing_flag=direct, multiply_among_weights_first_flag=mulfirst,
                          gat_edge_parallel_flag=edge_parallel, dropout=0.0, **layer_kw)
    with torch.no_grad():
        layer.h_bias.uniform_(-0.1, 0.1)
        layer.attn_l.mul_(attn_scale)
        layer.attn_r.mul_(attn_scale)
    x = torch.randn(N, K) * 0.5
    go = torch.randn(N, X)
    # oracle (fp64, autograd)
    s = g.get_separate_coo_original()
    if oracle_dev != "cpu":  # (1e5+ edges: the per-node nudge of tests/util.py, on the device)
        from tests.util import rgat_nudge_off_kink
        xd_, zmin = rgat_nudge_off_kink(x.to(oracle_dev), layer.conv_weights.to(oracle_dev), layer.attn_l.to(oracle_dev),
                                        layer.attn_r.to(oracle_dev), s)
        assert zmin >= 2e-6, zmin
        x = xd_.cpu()
    for _ in range(64 if oracle_dev == "cpu" else 0):  # no (edge, head) on the leaky-ReLU kink (see tests/util.py); fp32 rounding of el + er is ~3e-7
        if rgat_min_abs_preactivation(x, layer.conv_weights, layer.attn_l, layer.attn_r, s) >= 2e-6:
            break
        x = x + 1e-3 * torch.randn(N, K)
    p = {n: t.detach().double().to(oracle_dev).requires_grad_(True) for n, t in layer.named_parameters()}
    x64 = x.double().to(oracle_dev).requires_grad_(True)
    ref = OL.rgat_layer(x64, p["conv_weights"], p["attn_l"], p["attn_r"], s["rel_ptrs"].to(oracle_dev), s["row_indices"].to(oracle_dev),
                        s["col_indices"].to(oracle_dev), N, 0.2, p["loop_weight"], p["h_bias"])
    names = ["conv_weights", "attn_l", "attn_r", "loop_weight", "h_bias"]
    grads_ref = [t.cpu() for t in torch.autograd.grad(ref, [x64] + [p[n] for n in names], go.double().to(oracle_dev))]
    ref = ref.detach().cpu()
    # device
    g.to_(DEV)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = layer(g, xd)
    out.backward(go.to(DEV))
    g.cpu_()
    assert_close(out, ref, what="out")
    assert_close(xd.grad, grads_ref[0], what="grad_x")
    for n, gr in zip(names, grads_ref[1:]):
        assert_close(dict(layer.named_parameters())[n].grad, gr, what="grad_" + n)


@pytest.mark.parametrize("compact,direct,mulfirst", [(False, False, False), (False, False, True), (True, False, False),
                                                     (True, True, False), (True, True, True), (True, False, True)])
def test_rgat_layer_variants(compact, direct, mulfirst):
    _run_rgat(random_graph(seed=41, n=400, r=4, e=6000, shuffle=False), H=4, K=64, X=64, compact=compact, direct=direct, mulfirst=mulfirst)


@pytest.mark.parametrize("compact", [False, True])
def test_rgat_layer_large_preactivations_stay_finite(compact):
    """Pre-activations el + er of +-100 and more: exp() of them overflows fp32 (the reference's kernels, and the
    reference-named a4 here, would return inf / inf); the one-node layer subtracts a running maximum per (destination, head)
    and must agree with the fp64 oracle (whose raw exp is finite up to 709).  Graphs with and without destinations of more than
    256 in-edges (several work items that have to agree on one maximum)."""
    from tests.util import rgat_min_abs_preactivation
    for g in (mag_graph(scale=5e-3), random_graph(seed=44, n=300, r=4, e=5000, shuffle=False)):
        s = g.get_separate_coo_original()
        _run_rgat(g, H=4, K=64, X=64, compact=compact, direct=compact, mulfirst=False, attn_scale=300.0, seed=5)
    # the scale really produces such values
    from het_amd.layers import HET_RGATLayer
    torch.manual_seed(5)
    layer = HET_RGATLayer(64, 64, g.get_num_rels(), 4, self_loop=True, dropout=0.0)
    x = torch.randn(g.get_num_nodes(), 64) * 0.5
    R = layer.conv_weights.shape[0]
    rel = torch.repeat_interleave(torch.arange(R), s["rel_ptrs"][1:] - s["rel_ptrs"][:-1])
    wl = torch.einsum("rhkd,rhd->rhk", layer.conv_weights.detach(), layer.attn_l.detach() * 300)
    wr = torch.einsum("rhkd,rhd->rhk", layer.conv_weights.detach(), layer.attn_r.detach() * 300)
    z = torch.einsum("ek,ehk->eh", x[s["row_indices"]], wl[rel]) + torch.einsum("ek,ehk->eh", x[s["col_indices"]], wr[rel])
    assert float(z.max()) > 90.0, float(z.max())


@pytest.mark.parametrize("literal_er", [False, True])
@pytest.mark.parametrize("mulfirst", [False, True])
def test_rgat_layer_per_edge_dataflow(mulfirst, literal_er, monkeypatch):
    """Default flags on the per-edge (kind 0) dataflow of round 1 (HET_RGAT_PER_EDGE=1), with er as the reference's
    literal (x . W) . attn_r and as x . (W . attn_r): same layer, same oracle."""
    from het_amd.backend import rgat_fused_layer as FL
    monkeypatch.setattr(FL, "PER_EDGE", True)
    monkeypatch.setattr(FL, "LITERAL_ER", literal_er)
    _run_rgat(random_graph(seed=43, n=400, r=4, e=6000, shuffle=False), H=4, K=64, X=64, compact=False, direct=False, mulfirst=mulfirst)


def test_rgat_layer_default_flags_literal_er(monkeypatch):
    """Default flags on the distinct-row dataflow with er = (x . W) . attn_r as two products (HET_RGAT_LITERAL_ER=1)."""
    from het_amd.backend import rgat_fused_layer as FL
    monkeypatch.setattr(FL, "LITERAL_ER", True)
    _run_rgat(random_graph(seed=44, n=400, r=4, e=6000, shuffle=False), H=4, K=64, X=64, compact=False, direct=False, mulfirst=False)


@pytest.mark.parametrize("mulfirst", [False, True])
def test_rgat_reference_op_sequence_calls_only_reference_ops(mulfirst, monkeypatch):
    """The drop-in path (het_amd/backend/reference_protocol.py = RGAT/models.py:265-385 on the reference's wrappers):
    same values as the oracle, and every C entry point it reaches belongs to a reference-named torch_hrt op (plus the
    one-time grouping construction those ops do internally)."""
    from het_amd import _lib
    import het_amd.kernels as k
    called = []
    real = _lib.call

    def spy(name, *a):
        called.append(name)
        return real(name, *a)

    monkeypatch.setattr(_lib, "call", spy)
    _run_rgat(random_graph(seed=45, n=300, r=4, e=5000, shuffle=False), H=4, K=64, X=64, compact=False, direct=False,
              mulfirst=mulfirst, reference_op_sequence=True)
    allowed = {"het_" + n for n in k.REGISTERED_OPS} | {"het_grouping_create", "het_grouping_rank_of_position"}
    assert called and set(called) <= allowed, sorted(set(called) - allowed)
    assert {"het_rgnn_relational_matmul", "het_backward_rgnn_relational_matmul", "het_relational_fused_gat_separate_coo",
            "het_backward_relational_fused_gat_separate_coo"} <= set(called)


def test_rgat_layer_csr_path():
    _run_rgat(random_graph(seed=42, n=200, r=3, e=2000, empty_rel=False, shuffle=False), H=2, K=16, X=16, compact=False, direct=False,
              mulfirst=False, edge_parallel=False)


def test_rgat_layer_mag_like_small():
    """ogbn-mag-shaped graph at 0.2 % scale: typed node ranges, 4 relations, skewed degrees."""
    _run_rgat(mag_graph(2e-3), H=4, K=64, X=64, compact=False, direct=False, mulfirst=False)


@pytest.mark.parametrize("H", [1, 2])
@pytest.mark.parametrize("compact,mulfirst", [(False, False), (True, False), (False, True), (True, True)])
def test_rgat_layer_one_and_two_heads(H, compact, mulfirst):
    """--num_heads 1 is the reference's CLI default (RGAT/train_dgl.py): [E,H] tensors of 1 or 2 floats per row."""
    _run_rgat(random_graph(seed=44, n=320, r=4, e=5000, shuffle=False), H=H, K=64, X=64, compact=compact, direct=compact,
              mulfirst=mulfirst)


@pytest.mark.parametrize("H,K,X", [(1, 64, 8), (2, 64, 16), (1, 16, 16), (4, 64, 16), (1, 64, 4), (8, 64, 64), (4, 32, 32),
                                   (8, 128, 8),   # the reference's experiments/run_het_rgat.sh: 128 -> 8 classes, 8 heads of ONE float
                                   (4, 64, 8), (2, 64, 10), (2, 32, 6), (1, 64, 5),  # heads of 2 / 5 / 3 / 5 floats
                                   (4, 100, 64), (8, 16, 64), (2, 48, 24)])  # input widths outside 32 / 64 / 128: zero-padded columns
@pytest.mark.parametrize("compact,mulfirst,pad", [(False, False, True), (True, True, True), (False, False, False)])
def test_rgat_layer_small_and_odd_widths(H, K, X, compact, mulfirst, pad, monkeypatch):
    """Output widths outside the matrix-core shapes -- 64 -> 8 is the layer of the reference CLI's defaults (--n_infeat 64
    --num_classes 8 --num_heads 1), 128 -> 8 over 8 heads the reference's RGAT experiment -- and heads of 1 - 8 floats or of a
    width that is not a power of two.  pad: the layer zero-pads every head to the row kernels' widths (het_amd/layers.py
    _padded_head) and runs the one-node layer on the distinct-row dataflow; a spy checks that the row kernels ran, not the
    op-by-op composition.  Without the padding (HET_RGAT_PAD_HEADS=0): any-shape projection + row-dot and the non-cooperative
    gather kernels where the shapes allow, the op-by-op composition elsewhere.  n = 40: rows with hundreds of edges."""
    import het_amd.kernels as k
    import het_amd.layers as L
    monkeypatch.setattr(L, "PAD_HEADS", pad)
    D = X // H
    # (without padding the one-node layer takes power-of-two heads of 4+ floats and, for er from the folded weight, a power-of-two K >= 4 H)
    fused_without_padding = D >= 4 and D & (D - 1) == 0 and K & (K - 1) == 0 and K >= 4 * H
    calls = []
    real = k.rgat_aggregate_compact
    monkeypatch.setattr(k, "rgat_aggregate_compact", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    for n in (320, 40):
        _run_rgat(random_graph(seed=45, n=n, r=4, e=5000, shuffle=False), H=H, K=K, X=X, compact=compact, direct=compact,
                  mulfirst=mulfirst)
    assert len(calls) == (2 if (pad or fused_without_padding) else 0)


def test_rgat_layer_heads1_feat128():
    _run_rgat(random_graph(seed=43, n=300, r=5, e=4000, shuffle=False), H=1, K=128, X=128, compact=False, direct=False, mulfirst=False)


@pytest.mark.parametrize("compact,direct", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("K,D,R,pad", [(16, 16, 4, False), (16, 16, 4, True), (64, 64, 7, True),
                                       (128, 8, 4, True), (32, 16, 4, True), (64, 40, 3, True),  # (hrt/experiments/run_het_rgcn.sh: 128 | 32 -> 16 | 8)
                                       (100, 64, 4, True), (100, 64, 4, False), (48, 8, 3, True)])  # input widths outside 32 / 64 / 128
def test_rgcn_layer(compact, direct, K, D, R, pad, monkeypatch):
    """pad: output widths outside 32 / 64 / 128 run zero-padded on the matrix-core kernels (het_amd/layers.py PAD_WIDTHS);
    without it the any-shape kernels."""
    import het_amd.layers as L
    monkeypatch.setattr(L, "PAD_WIDTHS", pad)
    _run_rgcn(random_graph(seed=44, n=350, r=R, e=5000, shuffle=False), compact, direct, K, D, R)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("R", [4, 104])
def test_rgcn_layer_on_the_aifb_shaped_graph(R, compact):
    """BASELINE.json configs[0] (C1): one RGCN layer, feat 16, on the AIFB-shaped graph (N = 8 285, E = 58 086) with 4
    relations (as BASELINE.json states) and 104 (what the reference observes for DGL's AIFB,
    hrt/python/test/test_graphiler_load_data.py:18) against oracle/layers.py -- output, input and weight gradients."""
    from het_amd.graph import HetGraph
    from het_amd.synth import make_aifb_like
    g = HetGraph.from_integrated_coo(make_aifb_like(R))
    assert g.get_num_nodes() == 8285 and g.get_num_edges() == 58086 and g.get_num_rels() == R
    _run_rgcn(g, compact, compact, 16, 16, R)


@pytest.mark.parametrize("compact", [False, True])
def test_rgcn_layer_many_seeds(compact):
    """Eight random graphs per flag set: sizes, relation counts (one relation empty, one tiny), hub patterns, shuffled eids
    (the edge norm is indexed by eid), widths 16 / 32 / 64 / 128 in and out."""
    for seed in range(8):
        R = 3 + seed % 5
        g = random_graph(seed=300 + seed, n=90 + 53 * seed, r=R, e=1200 + 900 * seed, shuffle=seed % 2 == 1)
        K, D = ((16, 16), (64, 64), (32, 64), (128, 32))[seed % 4]
        _run_rgcn(g, compact, compact and seed % 3 != 0, K, D, R)


def _run_rgcn(g, compact, direct, K, D, R, oracle_dev="cpu"):
    from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
    torch.manual_seed(1)
    N, E = g.get_num_nodes(), g.get_num_edges()
    layer = HET_EglRelGraphConv_EdgeParallel(K, D, R, bias=True, compact_as_of_node_flag=compact,
                                             compact_direct_indexing_flag=direct)
    x, norm, go = torch.randn(N, K), torch.rand(E, 1), torch.randn(N, D)
    s = g.get_separate_coo_original()
    w64 = layer.weight.detach().double().to(oracle_dev).requires_grad_(True)
    b64 = layer.h_bias.detach().double().to(oracle_dev).requires_grad_(True)
    x64 = x.double().to(oracle_dev).requires_grad_(True)
    # (the layer reads the norm of an edge by its eid; the oracle takes it in separate-COO position order)
    ref = OL.rgcn_layer(x64, w64, norm.double()[s["eids"]].to(oracle_dev), s["rel_ptrs"].to(oracle_dev), s["row_indices"].to(oracle_dev),
                        s["col_indices"].to(oracle_dev), N, b64)
    gx_r, gw_r, gb_r = (t.cpu() for t in torch.autograd.grad(ref, [x64, w64, b64], go.double().to(oracle_dev)))
    ref = ref.detach().cpu()
    g.to_(DEV)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = layer(g, xd, norm.to(DEV))
    out.backward(go.to(DEV))
    g.cpu_()
    assert_close(out, ref, what="out")
    assert_close(xd.grad, gx_r, what="grad_x")
    assert_close(layer.weight.grad, gw_r, what="grad_W")
    assert_close(layer.h_bias.grad, gb_r, what="grad_bias")


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("K,D,R", [(64, 64, 4), (64, 64, 7), (32, 64, 3), (64, 32, 5), (32, 32, 1), (64, 64, 8)])
def test_rgcn_layer_two_call_form(K, D, R, fused, monkeypatch):
    """The RGCN layer as two library calls (het_rgcn_layer_forward / _backward: node-major output and input gradient, weight
    gradient from the kept (relation, destination) sums, bias gradient in the library) against the fp64 oracle, and the a7 / a8
    pair on the same graphs (HET_RGCN_FUSED=0): graphs with an empty relation, nodes without edges, shuffled eids, a hub.
    R = 8 at 64 x 64 exceeds the LDS the node pass keeps its weights in: the layer falls back to the pair."""
    import het_amd.backend.rgcn_layers_and_funcs as B
    import het_amd.kernels as k
    monkeypatch.setattr(B, "FUSED", fused)
    calls = []
    real = k.rgcn_layer_forward
    monkeypatch.setattr(k, "rgcn_layer_forward", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    for seed, n, e, shuffle in ((500, 97, 900, True), (501, 1500, 30000, False), (502, 4000, 2500, True)):
        _run_rgcn(random_graph(seed=seed, n=n, r=R, e=e, shuffle=shuffle, empty_rel=R > 2), False, False, K, D, R)
    assert len(calls) == (3 if fused and k.rgcn_layer_ok(R, K, D) else 0)
    assert k.rgcn_layer_ok(R, K, D) == (R < 8)


def test_rgcn_layer_norm_in_rank_order():
    """An edge norm that is the same tensor step after step is brought into the order of each gather grouping at its second
    sighting (kernels.scale_in_rank_order: the passes read it as a stream): same results as the by-edge-id gather (shuffled
    eids), an in-place edit of the norm is seen, a new tensor starts over."""
    import het_amd.kernels as k
    from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
    g = random_graph(seed=520, n=900, r=4, e=20000, shuffle=True)
    torch.manual_seed(4)
    N, E = g.get_num_nodes(), g.get_num_edges()
    layer = HET_EglRelGraphConv_EdgeParallel(64, 64, 4, bias=True).to(DEV)
    x, go = torch.randn(N, 64, device=DEV), torch.randn(N, 64, device=DEV)
    norm = torch.rand(E, 1, device=DEV)
    g.to_(DEV)
    s = g.get_separate_coo_original()

    def step(nrm):
        layer.zero_grad()
        xd = x.clone().requires_grad_(True)
        out = layer(g, xd, nrm)
        out.backward(go)
        return out.detach().clone(), xd.grad.clone(), layer.weight.grad.clone()

    def sorted_now():
        gd = k._plan.get_grouping(s["rel_ptrs"], s["col_indices"], N, s["row_indices"], s["eids"])
        gs = k._plan.get_grouping(s["rel_ptrs"], s["row_indices"], N, s["col_indices"], s["eids"])
        return [getattr(q, "_scale_sorted", None) is not None for q in (gd, gs)]

    first = step(norm)
    assert sorted_now() == [False, False]
    second = step(norm)
    assert sorted_now() == [True, True]
    third = step(norm)
    for a, b, c in zip(first, second, third):
        torch.testing.assert_close(a, b, rtol=0, atol=0)  # (the same sums of the same products in the same order)
        torch.testing.assert_close(a, c, rtol=0, atol=0)
    norm.mul_(2.0)  # in place: the version counter moves, the sorted copy is stale
    edited = step(norm)
    assert sorted_now() == [False, False]
    fresh = step(norm.clone())
    for a, b in zip(edited, fresh):
        torch.testing.assert_close(a, b, rtol=0, atol=0)
    torch.testing.assert_close(edited[1], 2.0 * first[1], rtol=1e-5, atol=1e-5)
    g.cpu_()


def test_rgcn_layer_fixed_input_features():
    """A layer input that needs no gradient (fixed features): the two-call form skips its gather pass and node pass in the backward
    (grad_x NULL) -- weight and bias gradients as with the gradient."""
    from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
    g = random_graph(seed=510, n=700, r=4, e=9000, shuffle=True)
    torch.manual_seed(3)
    N, E = g.get_num_nodes(), g.get_num_edges()
    layer = HET_EglRelGraphConv_EdgeParallel(64, 64, 4, bias=True).to(DEV)
    x, norm, go = torch.randn(N, 64, device=DEV), torch.rand(E, 1, device=DEV), torch.randn(N, 64, device=DEV)
    g.to_(DEV)
    grads = []
    for needs in (True, False):
        layer.zero_grad()
        xd = x.clone().requires_grad_(needs)
        layer(g, xd, norm).backward(go)
        assert (xd.grad is not None) == needs
        grads.append((layer.weight.grad.clone(), layer.h_bias.grad.clone()))
    g.cpu_()
    torch.testing.assert_close(grads[0][0], grads[1][0], rtol=1e-5, atol=1e-5)  # (dW runs beside another launch or alone: same sums)
    torch.testing.assert_close(grads[0][1], grads[1][1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("fused_attn,compact,direct", [(False, False, False), (True, False, False), (False, True, False),
                                                       (False, True, True)])
@pytest.mark.parametrize("H,in_dim,out_dim", [(8, 64, 64), (2, 12, 8), (1, 64, 64), (2, 64, 64), (4, 256, 256), (1, 64, 8), (4, 256, 64)])
def test_hgt_layer(fused_attn, compact, direct, H, in_dim, out_dim):
    """HGT layer (BASELINE.json configs[3]: feat 64, heads 8) against the plain-PyTorch fp64 oracle."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_HGTLayerHetero
    from het_amd.synth import make_random
    coo = make_random(300, 4, 4000, seed=51, num_ntypes=3)
    g = HetGraph.from_integrated_coo(coo)
    torch.manual_seed(2)
    N, R, T = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes()
    layer = HET_HGTLayerHetero(T, R, in_dim, out_dim, num_heads=H, dropout=0.0, hgt_fused_attn_score_flag=fused_attn,
                               compact_as_of_node_flag=compact, compact_direct_indexing_flag=direct)
    with torch.no_grad():
        layer.relation_pri.uniform_(0.5, 1.5)
        layer.skip.uniform_(-1, 1)
    h, go = torch.randn(N, in_dim) * 0.5, torch.randn(N, out_dim)
    s = g.get_separate_coo_original()
    names = ["k_linears", "q_linears", "v_linears", "a_linears", "relation_att", "relation_msg", "relation_pri", "skip"]
    p = {n: getattr(layer, n).detach().double().requires_grad_(True) for n in names}
    h64 = h.double().requires_grad_(True)
    ref = OL.hgt_layer(h64, g.get_original_node_type_offsets(), s["rel_ptrs"], s["row_indices"], s["col_indices"], N,
                       p["k_linears"], p["q_linears"], p["v_linears"], p["a_linears"], p["relation_att"], p["relation_msg"],
                       p["relation_pri"], p["skip"], H, fused_attn=fused_attn)
    grads_ref = torch.autograd.grad(ref, [h64] + [p[n] for n in names], go.double())
    g.to_(DEV)
    layer = layer.to(DEV)
    hd = h.to(DEV).requires_grad_(True)
    out = layer(g, hd)
    out.backward(go.to(DEV))
    g.cpu_()
    assert_close(out, ref, what="out")
    assert_close(hd.grad, grads_ref[0], what="grad_h")
    for n, gr in zip(names, grads_ref[1:]):
        assert_close(getattr(layer, n).grad, gr, what="grad_" + n)


@pytest.mark.parametrize("fused_attn,compact_dst", [(False, True), (True, True), (False, False)])
@pytest.mark.parametrize("H,in_dim,out_dim", [(8, 64, 64), (1, 64, 64), (4, 64, 64), (2, 32, 64), (1, 32, 32), (4, 128, 128), (2, 64, 16),
                                              (1, 64, 8),   # 64 -> 8, one head: the layer of the reference CLI's defaults
                                              (4, 64, 8), (2, 64, 10), (1, 64, 4),  # heads of 2 / 5 / 4 floats: zero-padded to 8
                                              (4, 100, 64), (2, 16, 32),  # input widths outside 32 / 64 / 128: zero-padded columns
                                              (8, 64, 256), (2, 32, 256)])  # rows wider than the row kernels take: the heads in two passes
def test_hgt_layer_fused(fused_attn, compact_dst, H, in_dim, out_dim, monkeypatch):
    _run_hgt_fused(fused_attn, compact_dst, H, in_dim, out_dim, monkeypatch)


def test_hgt_layer_large_scores_stay_finite(monkeypatch):
    """Attention scores of +-100 and more (relation_pri scaled up): exp() of them overflows fp32; the row kernels subtract a
    running maximum per (destination, head) and must agree with the fp64 oracle (raw exp, finite up to 709)."""
    _run_hgt_fused(False, True, 8, 64, 64, monkeypatch, pri=(150.0, 300.0))
    _run_hgt_fused(True, False, 4, 64, 64, monkeypatch, pri=(150.0, 300.0))


def _run_hgt_fused(fused_attn, compact_dst, H, in_dim, out_dim, monkeypatch, pri=(0.5, 1.5), g=None, oracle_dev="cpu"):
    """The HGT layer with attention + aggregation as one node on the distinct (relation, source) rows
    (het_amd/backend/hgt_fused_layer.py, csrc/hgt_compact.hip) -- what a full graph with canonical relations runs by default
    (BASELINE.json configs[3]: feat 64, heads 8) -- against the fp64 oracle: output and the gradients of the input and of all
    eight parameters.  A spy checks that the row kernels are what ran.  compact_dst: q, new_h and the output projection on the
    destinations that have in-edges only (the other output rows are zero by construction) / on all nodes."""
    import het_amd.kernels as k
    from het_amd.backend import hgt_fused_layer
    from het_amd.layers import HET_HGTLayerHetero
    monkeypatch.setattr(hgt_fused_layer, "COMPACT_DST_BELOW", 2.0 if compact_dst else 0.0)
    g = mag_graph(1.5e-3) if g is None else g
    torch.manual_seed(4)
    N, R, T = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes()
    layer = HET_HGTLayerHetero(T, R, in_dim, out_dim, num_heads=H, dropout=0.0, hgt_fused_attn_score_flag=fused_attn)
    with torch.no_grad():
        layer.relation_pri.uniform_(*pri)
        layer.skip.uniform_(-1, 1)
    h, go = torch.randn(N, in_dim) * 0.5, torch.randn(N, out_dim)
    s = g.get_separate_coo_original()
    names = ["k_linears", "q_linears", "v_linears", "a_linears", "relation_att", "relation_msg", "relation_pri", "skip"]
    p = {n: getattr(layer, n).detach().double().to(oracle_dev).requires_grad_(True) for n in names}
    h64 = h.double().to(oracle_dev).requires_grad_(True)
    ref = OL.hgt_layer(h64, g.get_original_node_type_offsets(), s["rel_ptrs"].to(oracle_dev), s["row_indices"].to(oracle_dev),
                       s["col_indices"].to(oracle_dev), N,
                       p["k_linears"], p["q_linears"], p["v_linears"], p["a_linears"], p["relation_att"], p["relation_msg"],
                       p["relation_pri"], p["skip"], H, fused_attn=fused_attn)
    grads_ref = [t.cpu() for t in torch.autograd.grad(ref, [h64] + [p[n] for n in names], go.double().to(oracle_dev))]
    ref = ref.detach().cpu()
    assert bool(torch.isfinite(ref).all())
    calls = []
    real_f, real_b = k.hgt_aggregate_compact, k.hgt_backward_compact
    monkeypatch.setattr(k, "hgt_aggregate_compact", lambda *a, **kw: (calls.append("fwd"), real_f(*a, **kw))[1])
    monkeypatch.setattr(k, "hgt_backward_compact", lambda *a, **kw: (calls.append("bwd"), real_b(*a, **kw))[1])
    g.to_(DEV)
    layer = layer.to(DEV)
    hd = h.to(DEV).requires_grad_(True)
    out = layer(g, hd)
    out.backward(go.to(DEV))
    g.cpu_()
    passes = hgt_fused_layer._head_groups(H, hgt_fused_layer._padded_head(out_dim // H))  # (rows above 128 floats: the heads in groups)
    assert calls == ["fwd"] * passes + ["bwd"] * passes
    assert_close(out, ref, what="out")
    try:
        assert_close(hd.grad, grads_ref[0], what="grad_h")
    except AssertionError as ex:  # which node types / parameters are off: says where in the backward chain the rows went wrong
        offs = g.get_original_node_type_offsets().tolist()
        bad = ~torch.isclose(hd.grad.detach().cpu().double(), grads_ref[0], rtol=1e-3, atol=1e-4 * float(grads_ref[0].abs().max())).all(1)
        zero = (hd.grad.detach().cpu() == 0).all(1)
        per_type = [(int(bad[a:b].sum()), int((bad & zero)[a:b].sum()), b - a) for a, b in zip(offs[:-1], offs[1:])]
        params = []
        for n, gr in zip(names, grads_ref[1:]):
            try:
                assert_close(getattr(layer, n).grad, gr, what="grad_" + n)
            except AssertionError:
                params.append(n)
        raise AssertionError(f"{ex}\n(bad rows, of them all-zero, nodes) per node type: {per_type}; parameter gradients off: {params}") from None
    for n, gr in zip(names, grads_ref[1:]):
        assert_close(getattr(layer, n).grad, gr, what="grad_" + n)


@pytest.mark.parametrize("rels_per_type", [1, 3, 4])
def test_hgt_layer_fused_node_major_input_gradient_shapes(rels_per_type, monkeypatch):
    """The node-major input gradient of the HGT layer (csrc/node_sum.hip) on typed graphs with 1, 3 and 4 relations leaving the
    same node type: 3 relations = 7 sources in one launch (4 waves per workgroup), 4 = 9 sources, more than the weights the
    pass keeps in LDS -- the layer falls back to the per-relation launches; a node type that is only a destination and one that is
    only a source are in the graph too.  Output and all gradients against the fp64 oracle either way."""
    import het_amd.kernels as k
    from het_amd.graph import HetGraph
    from het_amd.synth import make_hetero_graph
    # types: 0 source-only, 1 both, 2 destination-only; every relation leaves type 0 or 1
    rels = [(0, 2, 900)] + [(1, 1 + (i % 2), 700 + 50 * i) for i in range(rels_per_type)]
    g = HetGraph.from_integrated_coo(make_hetero_graph([150, 200, 120], rels, seed=5))
    calls = []
    real = k.node_rows_matmul_sum
    monkeypatch.setattr(k, "node_rows_matmul_sum", lambda *a, **kw: (calls.append(len(a[2])), real(*a, **kw))[1])
    _run_hgt_fused(False, True, 4, 64, 64, monkeypatch, g=g)
    from het_amd.backend import hgt_fused_layer
    if not hgt_fused_layer.NODE_DX:  # (HET_HGT_NODE_DX=0: the per-relation launches everywhere; parity was checked above)
        assert not calls
    elif rels_per_type <= 3:
        assert calls and max(calls) == 1 + 2 * rels_per_type, calls  # the destination term + two halves per relation (type 1)
    else:
        assert not calls  # 9 sources: outside the pass


def test_hgt_layer_fused_many_typed_graphs(monkeypatch):
    """Twelve random typed graphs through the fused HGT layer: 1-4 node types of uneven sizes (one of them may neither send nor
    receive), 1-6 relations with repeated (source type, destination type) pairs and relations inside one type, hubs, heads
    1 / 2 / 4 / 8, q on the destinations with in-edges or on all nodes -- the node-major input gradient's tiles, sources per
    launch and zero rows differ in every case.  Output and all gradients against the fp64 oracle."""
    import numpy as np
    from het_amd.graph import HetGraph
    from het_amd.synth import make_hetero_graph
    rng = np.random.Generator(np.random.PCG64(77))
    for case in range(12):
        T = int(rng.integers(1, 5))
        counts = [int(rng.integers(3, 260)) for _ in range(T)]
        live = list(range(T)) if (T == 1 or case % 3) else list(range(T - 1))  # every third case: the last type has no edges
        R = int(rng.integers(1, 7))
        rels = [(int(rng.choice(live)), int(rng.choice(live)), int(rng.integers(1, 2500))) for _ in range(R)]
        g = HetGraph.from_integrated_coo(make_hetero_graph(counts, rels, seed=200 + case))
        H = (1, 2, 4, 8)[case % 4]
        in_dim, out_dim = ((64, 64), (32, 64), (64, 32), (32, 32))[(case // 4) % 4]
        _run_hgt_fused(bool(case % 2), case % 5 != 0, H, in_dim, out_dim, monkeypatch, g=g)


@pytest.mark.parametrize("fused_attn", [False, True])
@pytest.mark.parametrize("H,in_dim,out_dim", [(1, 64, 64), (2, 32, 64), (1, 32, 64)])
def test_hgt_layer_multiply_among_weights_first(fused_attn, H, in_dim, out_dim, monkeypatch):
    """--multiply_among_weights_first_flag of HGT (HGT/models.py:124-151; heads = 1 in the reference's sweep): the typed
    K / Q / V projections folded into the relation weights.  Same function as the layer without the flag, so it is
    checked against the same fp64 oracle: output and the gradients of the input and of all eight parameters."""
    from het_amd.layers import HET_HGTLayerHetero
    from het_amd.backend import hgt_fused_layer
    monkeypatch.setattr(hgt_fused_layer, "FUSED", False)  # the op-by-op composition of the flag, not the one-node attention
    g = mag_graph(1.5e-3)  # typed node ranges, relations = canonical edge types
    torch.manual_seed(3)
    N, R, T = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes()
    layer = HET_HGTLayerHetero(T, R, in_dim, out_dim, num_heads=H, dropout=0.0, hgt_fused_attn_score_flag=fused_attn,
                               multiply_among_weights_first_flag=True)
    with torch.no_grad():
        layer.relation_pri.uniform_(0.5, 1.5)
        layer.skip.uniform_(-1, 1)
    h, go = torch.randn(N, in_dim) * 0.5, torch.randn(N, out_dim)
    s = g.get_separate_coo_original()
    names = ["k_linears", "q_linears", "v_linears", "a_linears", "relation_att", "relation_msg", "relation_pri", "skip"]
    p = {n: getattr(layer, n).detach().double().requires_grad_(True) for n in names}
    h64 = h.double().requires_grad_(True)
    ref = OL.hgt_layer(h64, g.get_original_node_type_offsets(), s["rel_ptrs"], s["row_indices"], s["col_indices"], N,
                       p["k_linears"], p["q_linears"], p["v_linears"], p["a_linears"], p["relation_att"], p["relation_msg"],
                       p["relation_pri"], p["skip"], H, fused_attn=fused_attn)
    grads_ref = torch.autograd.grad(ref, [h64] + [p[n] for n in names], go.double())
    g.to_(DEV)
    layer = layer.to(DEV)
    hd = h.to(DEV).requires_grad_(True)
    out = layer(g, hd)
    out.backward(go.to(DEV))
    g.cpu_()
    assert_close(out, ref, what="out")
    assert_close(hd.grad, grads_ref[0], what="grad_h")
    for n, gr in zip(names, grads_ref[1:]):
        assert_close(getattr(layer, n).grad, gr, what="grad_" + n)


@pytest.mark.parametrize("model,flags", [("rgat", ["--num_heads", "4"]),
                                         ("rgat", ["--num_heads", "2", "--num_layers", "2", "--compact_as_of_node_flag",
                                                   "--compact_direct_indexing_flag"]),
                                         ("rgcn", []), ("hgt", ["--num_heads", "4"])])
def test_train_driver_reference_flags(model, flags, tmp_path):
    """The benchmark driver with the reference's flag names: runs, loss decreases, JSON log written."""
    from het_amd import train
    log = tmp_path / "log.json"
    res = train.main(["--model", model, "-d", "mag", "--scale", "0.002", "--full_graph_training", "--n_infeat", "64",
                      "--num_classes", "64", "--n_epochs", "8", "--dropout", "0.0", "--lr", "0.01", "--logfile_enabled",
                      "--logfilename", str(log)] + flags)
    assert res["num_edges"] > 0 and res["mean_forward_ms"] > 0 and res["mean_backward_ms"] > 0
    assert res["final_loss"] < 4.3  # ln(64) = 4.16 at initialisation: finite and not diverging
    # (the edge softmax has no max-subtraction, as the reference's: large learning rates overflow exp -- SURVEY Q2)
    assert log.exists() and '"model"' in log.read_text()


@pytest.mark.parametrize("compact", [False, True])
def test_rgat_layer_on_degenerate_graphs(compact):
    """No edges at all; one relation only; more than 8 relations (the fused weight gradient keeps 8 in registers)."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import IntegratedCOO, make_random
    z = torch.zeros(0, dtype=torch.int64)
    cases = [IntegratedCOO(50, 3, torch.tensor([0, 50]), z, z.clone(), z.clone(), z.clone()),
             make_random(120, 1, 900, seed=5), make_random(150, 11, 2500, seed=6)]
    for coo in cases:
        g = HetGraph.from_integrated_coo(coo)
        torch.manual_seed(2)
        layer = HET_RGATLayer(64, 64, coo.num_rels, 4, self_loop=True, dropout=0.0, compact_as_of_node_flag=compact,
                              compact_direct_indexing_flag=compact)
        x, go = torch.randn(coo.num_nodes, 64), torch.randn(coo.num_nodes, 64)
        s = g.get_separate_coo_original()
        p64 = {k: v.detach().double().requires_grad_(True) for k, v in layer.named_parameters()}
        x64 = x.double().requires_grad_(True)
        ref = OL.rgat_layer(x64, p64["conv_weights"], p64["attn_l"], p64["attn_r"], s["rel_ptrs"], s["row_indices"],
                            s["col_indices"], coo.num_nodes, 0.2, p64["loop_weight"], p64["h_bias"])
        ref.backward(go.double())
        g.to_(DEV)
        layer = layer.to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        out = layer(g, xd)
        out.backward(go.to(DEV))
        assert_close(out, ref, what=f"out R={coo.num_rels} E={coo.num_edges}")
        assert_close(xd.grad, x64.grad, what="grad_x")
        for n, prm in layer.named_parameters():
            if prm.grad is not None:
                assert_close(prm.grad, p64[n].grad if p64[n].grad is not None else torch.zeros_like(p64[n]), what="grad_" + n)


def test_layer_step_is_hip_graph_capturable():
    """Forward + backward of the layer through the C-ABI kernels inside a HIP graph (no host synchronisation, no
    allocation outside torch's graph pool once the groupings exist): replay reproduces the eager gradients."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=2e-3)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(DEV))
    g = HetGraph.from_integrated_coo(coo, full=True)
    torch.manual_seed(5)
    layer = HET_RGATLayer(64, 64, 4, 4, self_loop=True, dropout=0.0).to(DEV)
    x = torch.nn.Parameter(torch.randn(coo.num_nodes, 64, device=DEV) * 0.1)
    go = torch.randn(coo.num_nodes, 64, device=DEV)
    params = [x] + list(layer.parameters())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on a side stream, as torch's graph capture requires
        for _ in range(3):
            for p in params:
                p.grad = None
            layer(g, x).backward(go)
    torch.cuda.current_stream().wait_stream(side)
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        layer(g, x).backward(go)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for p, r in zip(params, ref):
        torch.testing.assert_close(p.grad, r, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("mode", ["atomics", "op_by_op"])
@pytest.mark.parametrize("compact", [False, True])
def test_rgat_layer_fallback_paths(mode, compact, monkeypatch):
    """The layer without its single-node fast path: (atomics) groupings disabled -- every op on its atomics kernels, the
    reference's op sequence; (op_by_op) groupings on, single node off -- the fused ops of het_amd/layers.py."""
    import het_amd.plan as plan
    from het_amd.backend import rgat_fused_layer as FL
    old = plan.enabled
    try:
        if mode == "atomics":
            plan.enabled = False
            plan.clear()
        else:
            monkeypatch.setattr(FL, "rgat_layer_fused_ok", lambda *a, **k: False)
        _run_rgat(random_graph(seed=47, n=300, r=4, e=5000, shuffle=False), H=4, K=64, X=64, compact=compact, direct=compact,
                  mulfirst=False)
    finally:
        plan.enabled = old
        plan.clear()


@pytest.mark.parametrize("compact", [False, True])
def test_rgat_mulfirst_op_by_op_and_heads8(compact, monkeypatch):
    """--multiply_among_weights_first_flag outside the single-node layer (op-by-op composition: the row-dot forward on the
    distinct (relation, destination) rows) and, inside it, with 8 heads."""
    from het_amd.backend import rgat_fused_layer as FL
    g = random_graph(seed=49, n=280, r=4, e=4500, shuffle=False)
    _run_rgat(g, H=8, K=64, X=64, compact=compact, direct=compact, mulfirst=True)
    monkeypatch.setattr(FL, "rgat_layer_fused_ok", lambda *a, **k: False)
    _run_rgat(g, H=4, K=64, X=64, compact=compact, direct=compact, mulfirst=True)


@pytest.mark.parametrize("H,K,X,compact", [(4, 128, 128, False), (8, 64, 64, False), (8, 32, 128, True), (16, 64, 64, True),
                                           (4, 256, 256, False), (4, 256, 256, True), (1, 64, 8, False), (2, 64, 16, True),
                                           (4, 256, 64, True), (1, 256, 32, False), (2, 256, 128, True)])
def test_rgat_layer_other_shapes_single_node(H, K, X, compact):
    """feat = 128 (BASELINE.json configs[4]), 8 and 16 heads, K != X: the single-node layer on its other shapes."""
    _run_rgat(random_graph(seed=48, n=260, r=3, e=4000, shuffle=False), H=H, K=K, X=X, compact=compact, direct=compact,
              mulfirst=False)


@pytest.mark.parametrize("compact", [False, True])
def test_rgat_layer_many_seeds(compact):
    """Eight random graphs / parameter draws per flag set (different sizes, relation counts, hub patterns)."""
    for seed in range(8):
        g = random_graph(seed=100 + seed, n=120 + 37 * seed, r=2 + seed % 4, e=1500 + 700 * seed, shuffle=False)
        _run_rgat(g, H=4, K=64, X=64, compact=compact, direct=compact and seed % 2 == 0, mulfirst=False, seed=seed)


@pytest.mark.parametrize("fused_attn", [False, True])
@pytest.mark.parametrize("T,R,H,dk,in_dim", [(3, 5, 8, 8, 64), (4, 4, 1, 64, 64), (2, 7, 2, 4, 6), (3, 5, 4, 16, 100)])
def test_hgt_fold_kernel_matches_the_torch_composition(T, R, H, dk, in_dim, fused_attn, monkeypatch):
    """csrc/hgt_fold.hip (the HGT layer's per-step parameter folding as one launch, its backward as two) against the torch
    composition it replaces (hgt_fused_layer.fold_source_weights with HET_HGT_FOLD_KERNEL off) and that composition's autograd, in
    fp64: w_kv and all five parameter gradients."""
    import het_amd.backend.hgt_fused_layer as F
    gen = torch.Generator().manual_seed(13)
    X = H * dk
    mk = lambda *shape: torch.randn(*shape, generator=gen)
    k_lin, v_lin, att, msg = mk(T, 1, in_dim, X), mk(T, 1, in_dim, X), mk(R, H, dk, dk), mk(R, H, dk, dk)
    pri = torch.rand(R, H, generator=gen) + 0.5
    st = torch.randint(0, T, (R,), generator=gen)
    gw = mk(R, 1, in_dim, 2 * X)
    monkeypatch.setattr(F, "FOLD_KERNEL", False)
    ref_in = [t.double().requires_grad_(True) for t in (k_lin, v_lin, att, msg, pri)]
    w_ref = F.fold_source_weights(*ref_in, st, H, fused_attn)
    g_ref = torch.autograd.grad(w_ref, ref_in, gw.double())
    monkeypatch.setattr(F, "FOLD_KERNEL", True)
    dev_in = [t.to(DEV).requires_grad_(True) for t in (k_lin, v_lin, att, msg, pri)]
    w = F.fold_source_weights(*dev_in, st.to(DEV), H, fused_attn)
    assert w.shape == w_ref.shape and w.grad_fn is not None and "FoldSourceWeights" in type(w.grad_fn).__name__
    g = torch.autograd.grad(w, dev_in, gw.to(DEV))
    assert_close(w, w_ref.detach(), what="w_kv")
    for name, a, b in zip(("grad_k_lin", "grad_v_lin", "grad_att", "grad_msg", "grad_pri"), g, g_ref):
        assert_close(a, b, what=name)
