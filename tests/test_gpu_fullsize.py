"""Full-size checks (BASELINE.json sizes: the ogbn-mag-shaped graph, E = 21.1 M, N = 1.94 M, feat 64) through
size-independent properties: the CPU oracle would take minutes here, so each test pins the HIP path against a
closed-form consequence of the op's definition, computed with plain torch ops on the GPU."""
import pytest
import torch

from tests.util import rgat_nudge_off_kink

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def mag():
    from het_amd.graph import HetGraph
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=1.0)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(DEV))
    g = HetGraph.from_integrated_coo(coo, full=False)
    s = g.get_separate_coo_original()
    assert g.get_num_edges() == 21_111_007 and g.get_num_nodes() == 1_939_743 and g.get_num_rels() == 4
    assert torch.equal(s["eids"], torch.arange(g.get_num_edges(), device=DEV))
    return g


def test_projection_with_identity_weights_is_an_exact_gather(mag):
    """W[r,h] = columns h*D..(h+1)*D of the identity: the segment GEMM must reproduce x[row[i]] bit for bit
    (fp32 MFMA is an exact fma chain; every other product term is 0)."""
    import het_amd.backend as B
    s = mag.get_separate_coo_original()
    N, E, R, H, K = mag.get_num_nodes(), mag.get_num_edges(), 4, 4, 64
    D = K // H
    x = torch.randn(N, K, device=DEV)
    eye = torch.eye(K, device=DEV).view(K, H, D).permute(1, 0, 2).contiguous()  # [H, K, D]
    W = eye.unsqueeze(0).repeat(R, 1, 1, 1).contiguous()
    d = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"], "separate_coo_eids": s["eids"]}
    out = B.rgnn_relational_matmul(d, W, x, True, 0)
    # compare in 4 slabs to bound memory
    for a in range(0, E, 6_000_000):
        b = min(E, a + 6_000_000)
        assert torch.equal(out[a:b].view(b - a, K), x.index_select(0, s["row_indices"][a:b]))


def test_projection_backward_with_identity_weights_is_a_scatter_add(mag):
    """With identity weights grad_x[n] = sum of gradout rows gathered at n; grad_W[r] = x[row]^T gradout summed per relation
    is checked through its trace-like checksum sum_i <x[row_i], gradout_i>."""
    import het_amd.kernels as HK
    s = mag.get_separate_coo_original()
    N, E, R, H, K = mag.get_num_nodes(), mag.get_num_edges(), 4, 4, 64
    D = K // H
    x = torch.randn(N, K, device=DEV)
    go = torch.randn(E, H, D, device=DEV)
    eye = torch.eye(K, device=DEV).view(K, H, D).permute(1, 0, 2).contiguous()
    Wt = eye.unsqueeze(0).repeat(R, 1, 1, 1).transpose(2, 3).contiguous()
    d = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"], "separate_coo_eids": s["eids"]}
    gx, gW = torch.empty(N, K, device=DEV), torch.empty(R, H, K, D, device=DEV)
    HK.matmul_backward(d, 0, Wt, x, go, gx, gW, True, accumulate=False)
    ref = torch.zeros(N, K, device=DEV).index_add_(0, s["col_indices"], go.view(E, K))
    # both sides sum up to ~1e5 fp32 terms per hub node in different orders
    torch.testing.assert_close(gx, ref, rtol=2e-3, atol=5e-3)
    # sum over relations of trace(x[col]^T gradout) restricted to matching (k, (h,d)) pairs == <x[col], gradout>
    diag = sum(float((gW[r].permute(1, 0, 2).reshape(K, K)).diagonal().double().sum()) for r in range(R))
    want = float((x.index_select(0, s["col_indices"][:E]).double() * go.view(E, K).double()).sum())
    assert abs(diag - want) <= 1e-5 * max(1.0, abs(want)) + 50.0


def test_gat_attention_rows_sum_to_one_and_uniform_case_is_a_mean(mag):
    """(i) exp / sum[dst] sums to 1 over the in-edges of every destination; (ii) with el = er = 0 the attention is
    uniform, so ret[dst] is the mean of the feat rows of its in-edges and sum[dst] its in-degree."""
    import het_amd.backend as B
    s = mag.get_separate_coo_original()
    N, E, H, D = mag.get_num_nodes(), mag.get_num_edges(), 4, 16
    feat = torch.randn(E, H, D, device=DEV)
    el, er = torch.randn(E, H, device=DEV) * 0.5, torch.randn(E, H, device=DEV) * 0.5
    exp = el.new_empty(E, H); sm = el.new_empty(N, H); ret = feat.new_empty(N, H, D)
    import het_amd.kernels as HK
    HK.fused_gat_forward(s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], 0, {}, feat, el, er, sm, exp, ret, 0.2, None)
    a = exp / sm[s["col_indices"]]
    tot = torch.zeros(N, H, device=DEV).index_add_(0, s["col_indices"], a)
    indeg = torch.bincount(s["col_indices"], minlength=N)
    has = indeg > 0
    torch.testing.assert_close(tot[has], torch.ones_like(tot[has]), rtol=1e-4, atol=1e-4)
    assert float(ret[~has].abs().max()) == 0.0 and float(sm[~has].abs().max()) == 0.0
    zero = torch.zeros(E, H, device=DEV)
    HK.fused_gat_forward(s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], 0, {}, feat, zero, zero, sm, exp, ret, 0.2, None)
    torch.testing.assert_close(sm[:, 0], indeg.float(), rtol=1e-5, atol=0.5)
    mean = torch.zeros(N, H * D, device=DEV).index_add_(0, s["col_indices"], feat.view(E, H * D)) / indeg.clamp(min=1).unsqueeze(1)
    torch.testing.assert_close(ret.view(N, H * D), mean, rtol=1e-3, atol=1e-4)


def test_rgcn_layer_is_linear_in_the_edge_norm(mag):
    """ret(2*norm) == 2*ret(norm) and ret(norm1 + norm2) == ret(norm1) + ret(norm2) (the op is linear in norm)."""
    import het_amd.backend as B
    N, E, K = mag.get_num_nodes(), mag.get_num_edges(), 64
    x = torch.randn(N, K, device=DEV) * 0.3
    W = torch.randn(4, K, K, device=DEV) * 0.2
    n1, n2 = torch.rand(E, 1, device=DEV), torch.rand(E, 1, device=DEV)
    r1, r2, r12 = (B.rgcn_layer1_separate_coo(mag, x, W, n) for n in (n1, n2, n1 + n2))
    torch.testing.assert_close(r12, r1 + r2, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(B.rgcn_layer1_separate_coo(mag, x, W, 2 * n1), 2 * r1, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("heads", [4, 1])
def test_rgat_layer_dataflows_agree_at_full_size(heads):
    """The RGAT layer at BASELINE.json's size through its independent dataflows -- default flags (per-edge projections,
    kind-0 GAT kernels), compact + direct indexing (projections on unique (relation, node) rows, kind-4 kernels), and
    --multiply_among_weights_first_flag -- must produce the same output and the same gradients (same parameters, same
    input, same upstream gradient)."""
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=1.0)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(DEV))
    g = HetGraph.from_integrated_coo(coo, full=True)
    N = g.get_num_nodes()
    gen = torch.Generator(device=DEV).manual_seed(11)
    x0 = torch.randn(N, 64, device=DEV, generator=gen) * 0.3
    go = torch.randn(N, 64, device=DEV, generator=gen)
    # no (edge, head) pre-activation within 2e-6 of the leaky-ReLU kink (tests/util.py): the dataflows round el + er
    # differently (~3e-7), and an edge that lands on different sides of the kink changes a gradient by a finite amount
    torch.manual_seed(0)
    probe = HET_RGATLayer(64, 64, g.get_num_rels(), heads, self_loop=True, dropout=0.0)
    x0, zmin = rgat_nudge_off_kink(x0, probe.conv_weights.to(DEV), probe.attn_l.to(DEV), probe.attn_r.to(DEV),
                                   g.get_separate_coo_original())
    assert zmin >= 2e-6, zmin
    del probe
    results = []
    for flags in ({}, {"compact_as_of_node_flag": True, "compact_direct_indexing_flag": True},
                  {"multiply_among_weights_first_flag": True}):
        torch.manual_seed(0)
        layer = HET_RGATLayer(64, 64, g.get_num_rels(), heads, self_loop=True, dropout=0.0, **flags).to(DEV)
        x = x0.clone().requires_grad_(True)
        out = layer(g, x)
        out.backward(go)
        results.append((out.detach(), x.grad, layer.conv_weights.grad, layer.attn_l.grad, layer.attn_r.grad, layer.loop_weight.grad))
        del layer, out, x
    names = ("out", "grad_x", "grad_W", "grad_attn_l", "grad_attn_r", "grad_loop_weight")
    # hub nodes sum 1e5+ fp32 terms in a different order per dataflow: compare in norm and against the tensor's scale
    for other in results[1:]:
        for name, a, b in zip(names, results[0], other):
            a64, d64 = a.double(), (b.double() - a.double())
            rel_l2 = float(d64.norm() / a64.norm().clamp(min=1e-30))
            worst = float(d64.abs().max() / a64.abs().max().clamp(min=1e-30))
            # per-node tensors agree to ~1e-7, parameter gradients to ~1e-6 (sums over 2 M .. 21 M rows in different orders)
            tol_l2, tol_max = (2e-5, 1e-3)
            assert rel_l2 < tol_l2 and worst < tol_max, f"{name}: relative L2 error {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}"


@pytest.mark.parametrize("heads,flags,slope", [
    (4, {}, 0.2), (1, {}, 0.2),
    (4, {"compact_as_of_node_flag": True, "compact_direct_indexing_flag": True, "multiply_among_weights_first_flag": True}, 0.2),
    (4, {"compact_as_of_node_flag": True, "compact_direct_indexing_flag": True}, 0.2),
    (4, {"multiply_among_weights_first_flag": True}, 0.2),
    (4, {}, 1.0)])  # slope 1: no kink at all
def test_rgat_layer_matches_the_fp64_oracle_at_full_size(heads, flags, slope, feat=64, nudge=True):
    """BASELINE.json's RGAT configuration (ogbn-mag shape, feat 64) against oracle/layers.py evaluated in fp64 -- the
    oracle is plain torch, so at this size it runs on the GPU (minutes on the CPU): output, input gradient and every
    parameter gradient of the HIP layer, all held to relative L2 2e-5 / max 1e-3.  With slope < 1 the input is first moved
    off the leaky-ReLU kink (no |el + er| below 2e-6 in fp64: tests/util.py::rgat_nudge_off_kink); nudge=False is the
    explicitly named kink-tolerance test below."""
    from oracle import layers as OL
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=1.0)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(DEV))
    g = HetGraph.from_integrated_coo(coo, full=True)
    s = g.get_separate_coo_original()
    N = g.get_num_nodes()
    gen = torch.Generator(device=DEV).manual_seed(12)
    x0 = torch.randn(N, feat, device=DEV, generator=gen) * 0.3
    go = torch.randn(N, feat, device=DEV, generator=gen)
    torch.manual_seed(0)
    layer = HET_RGATLayer(feat, feat, g.get_num_rels(), heads, self_loop=True, dropout=0.0, leaky_relu_slope=slope, **flags).to(DEV)
    if nudge and slope != 1.0:
        x0, zmin = rgat_nudge_off_kink(x0, layer.conv_weights, layer.attn_l, layer.attn_r, s)
        assert zmin >= 2e-6, zmin
    x = x0.clone().requires_grad_(True)
    out = layer(g, x)
    out.backward(go)
    got = {"out": out.detach(), "grad_x": x.grad, "grad_W": layer.conv_weights.grad, "grad_attn_l": layer.attn_l.grad,
           "grad_attn_r": layer.attn_r.grad, "grad_loop_weight": layer.loop_weight.grad, "grad_h_bias": layer.h_bias.grad}
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in layer.named_parameters()}
    x64 = x0.double().requires_grad_(True)
    ref = OL.rgat_layer(x64, p64["conv_weights"], p64["attn_l"], p64["attn_r"], s["rel_ptrs"], s["row_indices"], s["col_indices"],
                        N, slope, p64["loop_weight"], p64["h_bias"])
    ref.backward(go.double())
    want = {"out": ref.detach(), "grad_x": x64.grad, "grad_W": p64["conv_weights"].grad, "grad_attn_l": p64["attn_l"].grad,
            "grad_attn_r": p64["attn_r"].grad, "grad_loop_weight": p64["loop_weight"].grad, "grad_h_bias": p64["h_bias"].grad}
    for name in got:
        a, d = want[name], got[name].double() - want[name]
        rel_l2 = float(d.norm() / a.norm().clamp(min=1e-30))
        worst = float(d.abs().max() / a.abs().max().clamp(min=1e-30))
        # typical: 1.5e-7 (out, grad_x), 1e-6 .. 8e-6 (parameter gradients).  Without the nudge (kink-tolerance test) a parameter
        # gradient may move by ~1e-3 of its norm per edge whose fp32 pre-activation falls on the other side of the kink
        tol_l2, tol_max = (2e-5, 1e-3) if (nudge or slope == 1.0 or name in ("out", "grad_x")) else (3e-3, 6e-3)
        print(f"[full-size vs fp64 oracle] feat={feat} heads={heads} slope={slope} nudge={nudge} flags={sorted(flags)} {name}: rel L2 {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}")
        assert rel_l2 < tol_l2 and worst < tol_max, f"{name}: relative L2 error {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}"


def test_rgat_layer_kink_tolerance_at_full_size():
    """KINK TOLERANCE: the same comparison on the raw random input.  Of its 84 M (edge, head) pre-activations a few hundred
    lie within 2e-6 of the leaky-ReLU kink; fp32 (these kernels, and the reference's) and fp64 may take different branches
    there -- a finite gradient difference that is an error of neither side.  Output and input gradient keep the tight
    bound; the parameter gradients are only held to relative L2 3e-3 here (the nudged runs above hold them to 2e-5)."""
    test_rgat_layer_matches_the_fp64_oracle_at_full_size(4, {}, 0.2, nudge=False)


def test_rgat_feat128_layer_matches_the_fp64_oracle_at_full_size():
    """BASELINE.json configs[4]'s single-GPU shape: RGAT on the full ogbn-mag-shaped graph at feat 128, 4 heads (the fp64
    oracle peaks at 173 GB of the 288 GB here)."""
    torch.cuda.empty_cache()
    test_rgat_layer_matches_the_fp64_oracle_at_full_size(4, {}, 0.2, feat=128)
    torch.cuda.empty_cache()


def _errors(got, want):
    d = got.double() - want
    return float(d.norm() / want.norm().clamp(min=1e-30)), float(d.abs().max() / want.abs().max().clamp(min=1e-30))


def _full_graph():
    from het_amd.graph import HetGraph
    from het_amd.synth import make_mag_like
    coo = make_mag_like(scale=1.0)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(DEV))
    return HetGraph.from_integrated_coo(coo, full=True)


@pytest.mark.parametrize("compact", [False, True])
def test_rgcn_layer_matches_the_fp64_oracle_at_full_size(compact):
    """BASELINE.json configs[1] (RGCN on the ogbn-mag shape, feat 64) against oracle/layers.py in fp64 on the GPU."""
    from oracle import layers as OL
    from het_amd.layers import HET_EglRelGraphConv_EdgeParallel
    g = _full_graph()
    s = g.get_separate_coo_original()
    N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
    gen = torch.Generator(device=DEV).manual_seed(13)
    x0 = torch.randn(N, 64, device=DEV, generator=gen) * 0.3
    norm = torch.rand(E, 1, device=DEV, generator=gen)
    go = torch.randn(N, 64, device=DEV, generator=gen)
    torch.manual_seed(0)
    layer = HET_EglRelGraphConv_EdgeParallel(64, 64, R, compact_as_of_node_flag=compact, compact_direct_indexing_flag=compact).to(DEV)
    x = x0.clone().requires_grad_(True)
    out = layer(g, x, norm)
    out.backward(go)
    w64 = layer.weight.detach().double().requires_grad_(True)
    b64 = layer.h_bias.detach().double().requires_grad_(True)
    x64 = x0.double().requires_grad_(True)
    ref = OL.rgcn_layer(x64, w64, norm.double(), s["rel_ptrs"], s["row_indices"], s["col_indices"], N, b64)
    ref.backward(go.double())
    for name, a, b in (("out", out.detach(), ref.detach()), ("grad_x", x.grad, x64.grad), ("grad_weight", layer.weight.grad, w64.grad),
                       ("grad_h_bias", layer.h_bias.grad, b64.grad)):
        rel_l2, worst = _errors(a, b)
        print(f"[full-size vs fp64 oracle] rgcn compact={compact} {name}: rel L2 {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}")
        assert rel_l2 < 2e-5 and worst < 1e-3, f"{name}: relative L2 error {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}"


@pytest.mark.parametrize("heads", [8, 1])
def test_hgt_layer_matches_the_fp64_oracle_at_full_size(heads):
    """BASELINE.json configs[3] (HGT on the ogbn-mag shape, feat 64, 8 heads; and the reference sweep's 1 head) against
    oracle/layers.py in fp64 on the GPU: output, input gradient, every parameter gradient."""
    from oracle import layers as OL
    from het_amd.layers import HET_HGTLayerHetero
    g = _full_graph()
    s = g.get_separate_coo_original()
    N, R, T = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes()
    gen = torch.Generator(device=DEV).manual_seed(14)
    h0 = torch.randn(N, 64, device=DEV, generator=gen) * 0.5
    go = torch.randn(N, 64, device=DEV, generator=gen)
    torch.manual_seed(0)
    layer = HET_HGTLayerHetero(T, R, 64, 64, num_heads=heads, dropout=0.0).to(DEV)
    with torch.no_grad():
        layer.relation_pri.uniform_(0.5, 1.5)
        layer.skip.uniform_(-1, 1)
    h = h0.clone().requires_grad_(True)
    out = layer(g, h)
    out.backward(go)
    names = ["k_linears", "q_linears", "v_linears", "a_linears", "relation_att", "relation_msg", "relation_pri", "skip"]
    p = {n: getattr(layer, n).detach().double().requires_grad_(True) for n in names}
    h64 = h0.double().requires_grad_(True)
    ref = OL.hgt_layer(h64, g.get_original_node_type_offsets(), s["rel_ptrs"], s["row_indices"], s["col_indices"], N,
                       p["k_linears"], p["q_linears"], p["v_linears"], p["a_linears"], p["relation_att"], p["relation_msg"],
                       p["relation_pri"], p["skip"], heads)
    ref.backward(go.double())
    checks = [("out", out.detach(), ref.detach()), ("grad_h", h.grad, h64.grad)] + [("grad_" + n, getattr(layer, n).grad, p[n].grad) for n in names]
    for name, a, b in checks:
        rel_l2, worst = _errors(a, b)
        print(f"[full-size vs fp64 oracle] hgt heads={heads} {name}: rel L2 {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}")
        tol_l2, tol_max = (2e-5, 1e-3) if name in ("out", "grad_h") else (1e-3, 3e-3)
        assert rel_l2 < tol_l2 and worst < tol_max, f"{name}: relative L2 error {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}"


def test_reference_named_ops_match_the_fp64_oracle_at_full_size():
    """a1 / a2 (segment GEMM, per-edge projection by source) and a4 / a5 (fused GAT, kind 0) called through the
    torch_hrt op names on the full ogbn-mag-shaped graph, against oracle/ops.py evaluated in fp64 on the GPU."""
    from oracle import ops as O
    import het_amd.kernels as k
    g = _full_graph()
    s = g.get_separate_coo_original()
    N, E, R, H, Kd, D = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels(), 4, 64, 16
    gen = torch.Generator(device=DEV).manual_seed(15)
    x = torch.randn(N, Kd, device=DEV, generator=gen) * 0.3
    W = torch.randn(R, H, Kd, D, device=DEV, generator=gen) * 0.2
    by_src = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"], "separate_coo_eids": s["eids"]}
    # a1
    feat = torch.empty(E, H, D, device=DEV)
    k.K.rgnn_relational_matmul(by_src, 0, W, x, feat, True)
    feat64 = torch.zeros(E, H, D, dtype=torch.float64, device=DEV)
    O.rgnn_relational_matmul(by_src, 0, W.double(), x.double(), feat64, True)
    rel_l2, worst = _errors(feat, feat64)
    assert rel_l2 < 1e-6 and worst < 1e-5, f"a1: {rel_l2:.2e} {worst:.2e}"
    # a2
    gfeat = torch.randn(E, H, D, device=DEV, generator=gen)
    gx, gW = torch.zeros(N, Kd, device=DEV), torch.zeros(R, H, Kd, D, device=DEV)
    Wt = W.transpose(2, 3).contiguous()
    k.K.backward_rgnn_relational_matmul(by_src, 0, Wt, x, gfeat, gx, gW, True)
    gx64, gW64 = torch.zeros(N, Kd, dtype=torch.float64, device=DEV), torch.zeros(R, H, Kd, D, dtype=torch.float64, device=DEV)
    O.backward_rgnn_relational_matmul(by_src, 0, Wt.double(), x.double(), gfeat.double(), gx64, gW64, True)
    for name, a, b in (("a2 grad_x", gx, gx64), ("a2 grad_w", gW, gW64)):
        rel_l2, worst = _errors(a, b)
        assert rel_l2 < 2e-6 and worst < 1e-4, f"{name}: {rel_l2:.2e} {worst:.2e}"
    # a4 / a5, edge-order el / er / exp as the reference op defines them
    el, er = torch.randn(E, H, device=DEV, generator=gen) * 0.5, torch.randn(E, H, device=DEV, generator=gen) * 0.5
    sm, ex, ret = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(N, H, D, device=DEV)
    idx = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
    k.K.relational_fused_gat_separate_coo(*idx, 0, {}, feat, el, er, sm, ex, ret, 0.2)
    sm64, ex64, ret64 = (torch.zeros(N, H, dtype=torch.float64, device=DEV), torch.zeros(E, H, dtype=torch.float64, device=DEV),
                         torch.zeros(N, H, D, dtype=torch.float64, device=DEV))
    O.relational_fused_gat_separate_coo(*idx, 0, {}, feat.double(), el.double(), er.double(), sm64, ex64, ret64, 0.2)
    for name, a, b in (("a4 sum", sm, sm64), ("a4 exp", ex, ex64), ("a4 ret", ret, ret64)):
        rel_l2, worst = _errors(a, b)
        assert rel_l2 < 2e-6 and worst < 1e-4, f"{name}: {rel_l2:.2e} {worst:.2e}"
    go = torch.randn(N, H, D, device=DEV, generator=gen)
    gf, gl, gr = torch.empty(E, H, D, device=DEV), torch.empty(E, H, device=DEV), torch.empty(E, H, device=DEV)
    k.K.backward_relational_fused_gat_separate_coo(*idx, 0, {}, feat, el, er, sm, ex, ret, go, gf, gl, gr, 0.2)
    gf64, gl64, gr64 = torch.zeros_like(feat64), torch.zeros_like(ex64), torch.zeros_like(ex64)
    O.backward_relational_fused_gat_separate_coo(*idx, 0, {}, feat.double(), el.double(), er.double(), sm64, ex64, ret64, go.double(),
                                                 gf64, gl64, gr64, 0.2)
    for name, a, b in (("a5 grad_feat", gf, gf64), ("a5 grad_el", gl, gl64), ("a5 grad_er", gr, gr64)):
        rel_l2, worst = _errors(a, b)
        assert rel_l2 < 5e-6 and worst < 1e-3, f"{name}: {rel_l2:.2e} {worst:.2e}"


def test_rgcn_and_hgt_ops_match_the_fp64_oracle_at_full_size():
    """a7 / a8 (RGCN layer op), a10 (HGT edge softmax and its backward) and a11 (HGT fused message + aggregation and its
    backward, 8 heads) through the torch_hrt op names on the full graph, against oracle/ops.py in fp64 on the GPU."""
    from oracle import ops as O
    import het_amd.kernels as k
    K = k.K
    g = _full_graph()
    s = g.get_separate_coo_original()
    N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
    gen = torch.Generator(device=DEV).manual_seed(16)
    rnd = lambda *shape: torch.randn(*shape, device=DEV, generator=gen)
    z64 = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=DEV)

    def check(name, got, want, tol_l2=2e-6, tol_max=1e-4):
        rel_l2, worst = _errors(got, want)
        assert rel_l2 < tol_l2 and worst < tol_max, f"{name}: relative L2 error {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}"

    # a7 / a8
    a = (s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"])
    x, W, norm, go = rnd(N, 64) * 0.3, rnd(R, 64, 64) * 0.2, torch.rand(E, 1, device=DEV, generator=gen), rnd(N, 64)
    ret, ref = torch.zeros(N, 64, device=DEV), z64(N, 64)
    K.rgcn_layer1_separate_coo(*a, x, W, norm, ret)
    O.rgcn_layer1_separate_coo(*a, x.double(), W.double(), norm.double(), ref)
    check("a7 ret", ret, ref)
    Wt = W.transpose(1, 2).contiguous()
    gx, gW, gn = torch.zeros(N, 64, device=DEV), torch.zeros(R, 64, 64, device=DEV), torch.zeros(E, 1, device=DEV)
    K.backward_rgcn_layer1_separate_coo(*a, x, Wt, norm, gn, gx, go, gW)
    gx64, gW64, gn64 = z64(N, 64), z64(R, 64, 64), z64(E, 1)
    O.backward_rgcn_layer1_separate_coo(*a, x.double(), Wt.double(), norm.double(), gn64, gx64, go.double(), gW64)
    check("a8 grad_x", gx, gx64); check("a8 grad_W", gW, gW64)
    del ret, ref, gx, gW, gx64, gW64
    # a10
    H, dk = 8, 8
    idx = (s["row_indices"], s["col_indices"], s["eids"], s["rel_ptrs"])
    score, mu, ga = rnd(E, H), torch.rand(R, H, device=DEV, generator=gen) + 0.5, rnd(E, H)
    sm, m, at = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(E, H, device=DEV)
    K.hgt_full_graph_edge_softmax_ops_separate_coo(*idx, score, mu, sm, m, at)
    sm64, m64, a64 = z64(N, H), z64(E, H), z64(E, H)
    O.hgt_full_graph_edge_softmax_ops_separate_coo(*idx, score.double(), mu.double(), sm64, m64, a64)
    check("a10 sum", sm, sm64); check("a10 m", m, m64); check("a10 a", at, a64)
    gs, gmu, tmp = torch.empty(E, H, device=DEV), torch.zeros(R, H, device=DEV), torch.empty(N, H, device=DEV)
    K.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(*idx, score, at, ga, mu, gs, gmu, tmp)
    gs64, gmu64, tmp64 = z64(E, H), z64(R, H), z64(N, H)
    O.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(*idx, score.double(), a64, ga.double(), mu.double(), gs64, gmu64, tmp64)
    check("a10 grad_score", gs, gs64, 5e-6, 1e-3); check("a10 grad_mu", gmu, gmu64, 1e-4, 1e-3)
    # a11
    v, Wm, gon = rnd(N, H, dk) * 0.5, rnd(R, H, dk, dk) * 0.4, rnd(N, H, dk)
    nh, nh64 = torch.zeros(N, H, dk, device=DEV), z64(N, H, dk)
    K.hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*a, v, Wm, at, nh)
    O.hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*a, v.double(), Wm.double(), a64, nh64)
    check("a11 new_h", nh, nh64)
    Wmt = Wm.transpose(2, 3).contiguous()
    gv, gWm, gat = torch.zeros(N, H, dk, device=DEV), torch.zeros(R, H, dk, dk, device=DEV), torch.empty(E, H, device=DEV)
    K.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*a, v, Wmt, at, nh, gv, gWm, gat, gon)
    gv64, gWm64, gat64 = z64(N, H, dk), z64(R, H, dk, dk), z64(E, H)
    O.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*a, v.double(), Wmt.double(), a64, nh64, gv64, gWm64, gat64, gon.double())
    check("a11 grad_v", gv, gv64); check("a11 grad_W", gWm, gWm64, 1e-5, 1e-4); check("a11 grad_a", gat, gat64)
