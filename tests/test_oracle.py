"""The CPU oracle checked against (i) the golden exp/sum vectors produced by the
reference's ref_rgat.py and (ii) torch autograd of an independent plain
formulation in fp64 (backward ops)."""
import pytest
import torch

from het_amd.graph import HetGraph
from het_amd.synth import IntegratedCOO, make_random
from oracle import ops as O

torch.manual_seed(0)
F64 = torch.float64


def _graph(seed=0, n=23, r=3, e=97, empty_rel=False):
    coo = make_random(n, r + (1 if empty_rel else 0), e, seed=seed)
    if empty_rel:  # make the middle relation empty
        coo.rel[coo.rel == 1] = 0
        coo.rel = torch.sort(coo.rel).values
    return HetGraph.from_integrated_coo(coo)


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_gat_exp_sum_golden(which, golden_toy, golden_mag):
    gold = golden_toy if which == "toy" else golden_mag
    n = int(gold["num_nodes"])
    el, er = gold["gat_el"], gold["gat_er"]
    E, H = el.shape
    feat = torch.randn(E, H, 3)
    s, exp, ret = torch.empty(n, H), torch.empty(E, H), torch.empty(n, H, 3)
    O.relational_fused_gat_separate_coo(torch.arange(E), gold["sep_rel_ptrs"], gold["sep_row"], gold["sep_col"],
                                        0, {}, feat, el, er, s, exp, ret, float(gold["gat_slope"]))
    torch.testing.assert_close(exp, gold["gat_exp"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(s, gold["gat_sum"], rtol=1e-5, atol=1e-6)
    # rows of the attention matrix sum to one for every destination with in-edges
    a = exp / s[gold["sep_col"]]
    tot = torch.zeros(n, H).index_add_(0, gold["sep_col"], a)
    has = torch.zeros(n, dtype=torch.bool); has[gold["sep_col"]] = True
    torch.testing.assert_close(tot[has], torch.ones_like(tot[has]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_gat_compact_exp_sum_golden(which, golden_toy, golden_mag):
    """CompactAsOfNodeKind 4 of the fused GAT forward (el per (relation, source) row, er per (relation, destination) row,
    inverse indices by edge) against the exp / sum the reference's dual-unique-list wrapper produced (ref_rgat.py:77-115)."""
    gold = golden_toy if which == "toy" else golden_mag
    n = int(gold["num_nodes"])
    el, er = gold["gatc_el"], gold["gatc_er"]
    E, H = gold["gatc_exp"].shape
    d = {"edata_idx_to_inverse_idx_row": gold["ss_inverse_indices_row"], "edata_idx_to_inverse_idx_col": gold["ss_inverse_indices_col"]}
    feat = torch.randn(el.shape[0], H, 3)
    s, exp, ret = torch.empty(n, H), torch.empty(E, H), torch.empty(n, H, 3)
    O.relational_fused_gat_separate_coo(torch.arange(E), gold["sep_rel_ptrs"], gold["sep_row"], gold["sep_col"],
                                        4, d, feat, el, er, s, exp, ret, float(gold["gat_slope"]))
    torch.testing.assert_close(exp, gold["gatc_exp"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(s, gold["gatc_sum"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_gat_backward_grad_feat_src_golden(which, golden_toy, golden_mag):
    """grad_feat_src of the fused GAT backward against the reference's own backward (ref_rgat.py:66-75).  The reference
    file indexes feat_src by SOURCE NODE, the exported kind-0 op by edge ([E,H,D] rows): the per-edge rows of the op are
    summed per source node before comparing.  (Its grad_el / grad_er, :64-65, are not comparable: DESIGN.md section 3.)"""
    gold = golden_toy if which == "toy" else golden_mag
    n = int(gold["num_nodes"])
    go = gold["gatb_gradout"]
    E, H = gold["gat_exp"].shape
    D = go.shape[2]
    feat = torch.randn(E, H, D)
    gf, gl, gr = torch.zeros(E, H, D), torch.zeros(E, H), torch.zeros(E, H)
    O.backward_relational_fused_gat_separate_coo(torch.arange(E), gold["sep_rel_ptrs"], gold["sep_row"], gold["sep_col"], 0, {},
                                                 feat, gold["gat_el"], gold["gat_er"], gold["gat_sum"], gold["gat_exp"],
                                                 torch.randn(n, H, D), go, gf, gl, gr, float(gold["gat_slope"]))
    per_node = torch.zeros(n, H, D).index_add_(0, gold["sep_row"], gf)
    # the reference sums a hub source's edges in fp32 in its own order: a few ulp of the largest partial sum
    torch.testing.assert_close(per_node, gold["gatb_grad_feat_src"], rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_gat_round5_pins_golden(which, golden_toy, golden_mag):
    """Kinds 1 / 2 of the fused GAT forward and grad_feat of kinds 4 / 1 of its backward against what the rest of the reference's
    ref_rgat.py produces (tests/golden/make_golden.py, round 5; tests/util.py::check_round5_gat_pins)."""
    from tests.util import check_round5_gat_pins
    gold = golden_toy if which == "toy" else golden_mag
    check_round5_gat_pins(O, "cpu", gold, gold, gold, float(gold["gat_slope"]))


@pytest.mark.parametrize("which", ["toy", "mag"])
def test_gat_kind3_rows_equal_kind4(which, golden_toy, golden_mag):
    """CompactAsOfNodeKind 3 (dual lists, rows found by binary search) addresses the same rows as kind 4 (rows read from the inverse
    indices) on the reference builders' own lists -- so the kind-4 pins cover kind 3's arithmetic."""
    gold = golden_toy if which == "toy" else golden_mag
    E = gold["sep_row"].numel()
    d3 = {"unique_srcs_and_dests_rel_ptrs": gold["ss_rel_ptrs_row"], "unique_srcs_and_dests_rel_ptrs_col": gold["ss_rel_ptrs_col"],
          "unique_srcs_and_dests_node_indices_row": gold["ss_node_indices_row"],
          "unique_srcs_and_dests_node_indices_col": gold["ss_node_indices_col"]}
    d4 = {"edata_idx_to_inverse_idx_row": gold["ss_inverse_indices_row"], "edata_idx_to_inverse_idx_col": gold["ss_inverse_indices_col"]}
    ar = torch.arange(E)
    r3 = O._gat_rows(3, d3, gold["sep_rel_ptrs"], gold["sep_row"], gold["sep_col"], ar)
    r4 = O._gat_rows(4, d4, gold["sep_rel_ptrs"], gold["sep_row"], gold["sep_col"], ar)
    assert torch.equal(r3[0], r4[0]) and torch.equal(r3[1], r4[1])


def _plain_matmul(rp, gather, scatter, W, x, in1head, nrows_out):
    R, H, K, D = W.shape
    out = torch.zeros(nrows_out, H, D, dtype=W.dtype)
    rel = O.rel_of_position(rp)
    xin = x[gather]
    Wp = W[rel]  # [n,H,K,D]
    if in1head:
        y = torch.einsum("nk,nhkd->nhd", xin, Wp)
    else:
        y = torch.einsum("nhk,nhkd->nhd", xin, Wp)
    return out.index_put((scatter,), y)


@pytest.mark.parametrize("kind,in1head", [(0, True), (0, False), (1, True)])
@pytest.mark.parametrize("empty_rel", [False, True])
def test_matmul_fwd_bwd_vs_autograd(kind, in1head, empty_rel):
    g = _graph(seed=3, empty_rel=empty_rel)
    s = g.get_separate_coo_original()
    R, H, K, D = g.get_num_rels(), 2, 5, 3
    N, E = g.get_num_nodes(), g.get_num_edges()
    W = torch.randn(R, H, K, D, dtype=F64, requires_grad=True)
    if kind == 0:
        perm = torch.randperm(E)
        d = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"],
             "separate_coo_eids": perm}
        rp, gather, scatter, nout = s["rel_ptrs"], s["row_indices"], perm, E
    else:
        u = g.get_separate_unique_node_indices_single_sided()
        d = {"unique_srcs_and_dests_rel_ptrs": u["rel_ptrs_row"], "unique_srcs_and_dests_node_indices": u["node_indices_row"]}
        rp, gather, nout = u["rel_ptrs_row"], u["node_indices_row"], int(u["rel_ptrs_row"][-1])
        scatter = torch.arange(nout)
    x = torch.randn(N, K, dtype=F64, requires_grad=True) if in1head else torch.randn(N, H, K, dtype=F64, requires_grad=True)
    ref = _plain_matmul(rp, gather, scatter, W, x, in1head, nout)
    ret = torch.zeros(nout, H, D, dtype=F64)
    O.rgnn_relational_matmul(d, kind, W.detach(), x.detach(), ret, in1head)
    torch.testing.assert_close(ret, ref.detach())
    go = torch.randn_like(ref)
    gW_ref, gx_ref = torch.autograd.grad(ref, (W, x), go)
    gx, gW = torch.zeros_like(x), torch.zeros_like(W)
    O.backward_rgnn_relational_matmul(d, kind, W.detach().transpose(2, 3).contiguous(), x.detach(), go, gx, gW, in1head)
    torch.testing.assert_close(gx, gx_ref)
    torch.testing.assert_close(gW, gW_ref)


@pytest.mark.parametrize("H,per_head", [(1, False), (3, False), (3, True)])
def test_matmul_no_scatter_gather(H, per_head):
    offsets = torch.tensor([0, 4, 4, 11, 17])
    T, K, D, n = 4, 6, 5, 17
    W = torch.randn(T, H, K, D, dtype=F64, requires_grad=True)
    x = torch.randn(n, H, K, dtype=F64, requires_grad=True) if per_head else torch.randn(n, K, dtype=F64, requires_grad=True)
    seg = O.rel_of_position(offsets)
    ref = torch.einsum("nhk,nhkd->nhd", x, W[seg]) if per_head else torch.einsum("nk,nhkd->nhd", x, W[seg])
    ret = torch.zeros(n, H, D, dtype=F64)
    O.rgnn_relational_matmul_no_scatter_gather_list(offsets, W.detach(), x.detach(), ret)
    torch.testing.assert_close(ret, ref.detach())
    go = torch.randn_like(ref)
    gW_ref, gx_ref = torch.autograd.grad(ref, (W, x), go)
    gx, gW = torch.zeros_like(x), torch.zeros_like(W)
    O.backward_rgnn_relational_matmul_no_scatter_gather_list(offsets, W.detach().transpose(2, 3).contiguous(), x.detach(), go, gx, gW)
    torch.testing.assert_close(gx, gx_ref)
    torch.testing.assert_close(gW, gW_ref)


def _gat_dict(g, kind):
    if kind == 0:
        return {}, {}
    if kind == 1:
        u = g.get_separate_unique_node_indices()
        d = {"unique_srcs_and_dests_rel_ptrs": u["rel_ptrs"], "unique_srcs_and_dests_node_indices": u["node_indices"]}
        return d, d
    if kind == 3:
        u = g.get_separate_unique_node_indices_single_sided()
        d = {"unique_srcs_and_dests_rel_ptrs": u["rel_ptrs_row"], "unique_srcs_and_dests_rel_ptrs_col": u["rel_ptrs_col"],
             "unique_srcs_and_dests_node_indices_row": u["node_indices_row"],
             "unique_srcs_and_dests_node_indices_col": u["node_indices_col"]}
        db = dict(d); db["unique_srcs_and_dests_rel_col"] = db.pop("unique_srcs_and_dests_rel_ptrs_col")
        return d, db
    if kind == 2:  # one inverse index for both edge ends (as the reference reads it): the source-side one here
        u = g.get_separate_unique_node_indices_single_sided_inverse_idx()
        d = {"edata_idx_to_inverse_idx": u["inverse_indices_row"]}
        return d, d
    u = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    d = {"edata_idx_to_inverse_idx_row": u["inverse_indices_row"], "edata_idx_to_inverse_idx_col": u["inverse_indices_col"]}
    return d, d


def _gat_sizes(g, kind):
    E = g.get_num_edges()
    if kind == 0:
        return E, E
    if kind == 1:
        u = int(g.get_separate_unique_node_indices()["rel_ptrs"][-1])
        return u, u
    ss = g.get_separate_unique_node_indices_single_sided()
    if kind == 2:
        return int(ss["rel_ptrs_row"][-1]), int(ss["rel_ptrs_row"][-1])
    return int(ss["rel_ptrs_row"][-1]), int(ss["rel_ptrs_col"][-1])


def _plain_gat(g, kind, feat, el, er, slope):
    s = g.get_separate_coo_original()
    d, _ = _gat_dict(g, kind)
    srow, drow = O._gat_rows(kind, d, s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"])
    z = torch.nn.functional.leaky_relu(el[srow] + er[drow], slope)
    N = g.get_num_nodes()
    col = s["col_indices"]
    # softmax over all in-edges of a destination, via a dense max-shifted formulation
    zmax = torch.full((N, z.shape[1]), -1e30, dtype=z.dtype).scatter_reduce(0, col.unsqueeze(-1).expand_as(z), z, "amax")
    ez = torch.exp(z - zmax[col])
    den = torch.zeros(N, z.shape[1], dtype=z.dtype).index_add(0, col, ez)
    a = ez / den[col]
    return torch.zeros(N, *feat.shape[1:], dtype=feat.dtype).index_add(0, col, a.unsqueeze(-1) * feat[srow])


@pytest.mark.parametrize("kind", [0, 1, 3, 4])
def test_fused_gat_fwd_bwd_vs_autograd(kind):
    g = _graph(seed=5, n=19, r=3, e=83)
    s = g.get_separate_coo_original()
    H, D, slope = 2, 4, 0.2
    ns, nd = _gat_sizes(g, kind)
    feat = torch.randn(ns, H, D, dtype=F64, requires_grad=True)
    el = torch.randn(ns, H, dtype=F64, requires_grad=True)
    er = torch.randn(nd, H, dtype=F64, requires_grad=True)
    ref = _plain_gat(g, kind, feat, el, er, slope)
    N, E = g.get_num_nodes(), g.get_num_edges()
    sm, ex, ret = torch.empty(N, H, dtype=F64), torch.empty(E, H, dtype=F64), torch.empty(N, H, D, dtype=F64)
    df, db = _gat_dict(g, kind)
    args = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], kind)
    O.relational_fused_gat_separate_coo(*args, df, feat.detach(), el.detach(), er.detach(), sm, ex, ret, slope)
    torch.testing.assert_close(ret, ref.detach())
    go = torch.randn_like(ref)
    gf_ref, gl_ref, gr_ref = torch.autograd.grad(ref, (feat, el, er), go)
    gf, gl, gr = torch.zeros_like(feat), torch.zeros_like(el), torch.zeros_like(er)
    O.backward_relational_fused_gat_separate_coo(*args, db, feat.detach(), el.detach(), er.detach(), sm, ex, ret, go, gf, gl, gr, slope)
    torch.testing.assert_close(gf, gf_ref)
    torch.testing.assert_close(gl, gl_ref)
    torch.testing.assert_close(gr, gr_ref)


def test_fused_gat_csr_matches_coo():
    g = _graph(seed=6, n=17, r=2, e=61)
    s = g.get_separate_coo_original()
    H, D, slope = 3, 2, 0.2
    N, E = g.get_num_nodes(), g.get_num_edges()
    feat, el, er = torch.randn(E, H, D, dtype=F64), torch.randn(E, H, dtype=F64), torch.randn(E, H, dtype=F64)
    out = []
    for csr in (False, True):
        sm, ex, ret = torch.empty(N, H, dtype=F64), torch.empty(E, H, dtype=F64), torch.empty(N, H, D, dtype=F64)
        gf, gl, gr = torch.zeros_like(feat), torch.zeros_like(el), torch.zeros_like(er)
        go = torch.randn(N, H, D, dtype=F64, generator=torch.Generator().manual_seed(1))
        if csr:
            i, o = g.get_in_csr(), g.get_out_csr()
            dummy = torch.zeros(0, dtype=torch.int64)
            O.relational_fused_gat_csr(i["row_ptrs"], i["col_indices"], i["eids"], i["rel_types"], dummy, dummy, feat, el, er, sm, ex, ret, slope, False)
            O.backward_relational_fused_gat_csr(o["row_ptrs"], o["col_indices"], o["eids"], o["rel_types"], dummy, dummy, feat, el, er, sm, ex, ret, go, gf, gl, gr, slope, False)
        else:
            a = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], 0, {})
            O.relational_fused_gat_separate_coo(*a, feat, el, er, sm, ex, ret, slope)
            O.backward_relational_fused_gat_separate_coo(*a, feat, el, er, sm, ex, ret, go, gf, gl, gr, slope)
        out.append((sm, ex, ret, gf, gl, gr))
    for a, b in zip(*out):
        torch.testing.assert_close(a, b)


def test_rgcn_layer_fwd_bwd_vs_autograd():
    g = _graph(seed=7, n=21, r=4, e=90, empty_rel=True)
    s = g.get_separate_coo_original()
    R, K, D = g.get_num_rels(), 6, 5
    N, E = g.get_num_nodes(), g.get_num_edges()
    x = torch.randn(N, K, dtype=F64, requires_grad=True)
    W = torch.randn(R, K, D, dtype=F64, requires_grad=True)
    norm = torch.rand(E, 1, dtype=F64)
    rel = O.rel_of_position(s["rel_ptrs"])
    msg = torch.einsum("nk,nkd->nd", x[s["row_indices"]] * norm[s["eids"]], W[rel])
    ref = torch.zeros(N, D, dtype=F64).index_add(0, s["col_indices"], msg)
    ret = torch.zeros(N, D, dtype=F64)
    a = (s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"])
    O.rgcn_layer1_separate_coo(*a, x.detach(), W.detach(), norm, ret)
    torch.testing.assert_close(ret, ref.detach())
    go = torch.randn_like(ref)
    gx_ref, gW_ref = torch.autograd.grad(ref, (x, W), go)
    gx, gW, gn = torch.zeros_like(x), torch.zeros_like(W), torch.zeros_like(norm)
    O.backward_rgcn_layer1_separate_coo(*a, x.detach(), W.detach().transpose(1, 2).contiguous(), norm, gn, gx, go, gW)
    torch.testing.assert_close(gx, gx_ref)
    torch.testing.assert_close(gW, gW_ref)
    assert float(gn.abs().sum()) == 0.0


@pytest.mark.parametrize("direct", [False, True])
def test_rgcn_compact_aggregation(direct):
    g = _graph(seed=8, n=15, r=3, e=70)
    s = g.get_separate_coo_original()
    ss = g.get_separate_unique_node_indices_single_sided()
    ssi = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    U, D = int(ss["rel_ptrs_row"][-1]), 4
    N, E = g.get_num_nodes(), g.get_num_edges()
    feat = torch.randn(U, D, dtype=F64, requires_grad=True)
    enorm = torch.rand(E, 1, dtype=F64)
    d = {"inverse_indices_row": ssi["inverse_indices_row"]} if direct else \
        {"rel_ptrs_row": ss["rel_ptrs_row"], "node_indices_row": ss["node_indices_row"]}
    ref = torch.zeros(N, D, dtype=F64).index_add(0, s["col_indices"], enorm[s["eids"]] * feat[ssi["inverse_indices_row"][s["eids"]]])
    ret = torch.empty(N, D, dtype=F64)
    a = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], d)
    O.rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(*a, feat.detach(), enorm, ret, direct)
    torch.testing.assert_close(ret, ref.detach())
    go = torch.randn_like(ref)
    (gf_ref,) = torch.autograd.grad(ref, (feat,), go)
    gf = torch.zeros_like(feat)
    O.backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(*a, feat.detach(), enorm, ret, go, gf, direct)
    torch.testing.assert_close(gf, gf_ref)


# ---------------------------------------------------------------- HGT ops vs autograd
def test_hgt_ops_compose_to_layer_and_match_autograd():
    """The HGT op oracles, chained as the reference layer chains them, reproduce the plain autograd layer:
    outputs and every gradient (edge softmax backward, fused message backward, inner product backward)."""
    from oracle import layers as OL
    coo = make_random(31, 3, 140, seed=9, num_ntypes=3)
    g = HetGraph.from_integrated_coo(coo)
    s = g.get_separate_coo_original()
    N, E, R, T, H, Din, Dout = g.get_num_nodes(), g.get_num_edges(), 3, 3, 2, 6, 8
    dk = Dout // H
    gen = torch.Generator().manual_seed(3)
    mk = lambda *sh: (torch.randn(*sh, generator=gen, dtype=F64) * 0.4).requires_grad_(True)
    h, k_lin, q_lin, v_lin, a_lin = mk(N, Din), mk(T, 1, Din, Dout), mk(T, 1, Din, Dout), mk(T, 1, Din, Dout), mk(T, 1, Dout, Dout)
    rel_att, rel_msg, rel_pri, skip = mk(R, H, dk, dk), mk(R, H, dk, dk), mk(R, H), mk(T, 1, 1, 1)
    offs = g.get_original_node_type_offsets()
    for fused in (False, True):
        ref = OL.hgt_layer(h, offs, s["rel_ptrs"], s["row_indices"], s["col_indices"], N, k_lin, q_lin, v_lin, a_lin,
                           rel_att, rel_msg, rel_pri, skip, H, fused_attn=fused)
        go = torch.randn(N, Dout, generator=gen, dtype=F64)
        leaves = [h, k_lin, q_lin, v_lin, a_lin, rel_att, rel_msg, rel_pri]
        grads_ref = torch.autograd.grad(ref, leaves, go)
        # --- the same layer through the op oracles, manual backward
        idx = (s["row_indices"], s["col_indices"], s["eids"], s["rel_ptrs"])
        d = lambda t: t.detach()
        lin = lambda W, x: (lambda r: (O.rgnn_relational_matmul_no_scatter_gather_list(offs, d(W), d(x), r), r)[1])(torch.zeros(x.shape[0], 1, W.shape[3], dtype=F64))
        k, q, v = (lin(W, h).view(N, H, dk) for W in (k_lin, q_lin, v_lin))
        score = torch.zeros(E, H, dtype=F64)
        inner = torch.zeros(E, H, dk, dtype=F64)
        dd = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"], "separate_coo_eids": s["eids"]}
        if fused:
            O.hgt_full_graph_hetero_attention_ops_coo(*idx, k, q, d(rel_att), inner, score)
        else:
            O.rgnn_relational_matmul(dd, 0, d(rel_att), q, inner, False)
            O.rgnn_inner_product_right_node_separatecoo({}, 0, s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], inner, k, score)
        mu = d(rel_pri) / dk ** 0.5
        sm, m, a = torch.zeros(N, H, dtype=F64), torch.zeros(E, H, dtype=F64), torch.zeros(E, H, dtype=F64)
        O.hgt_full_graph_edge_softmax_ops_separate_coo(*idx, score, mu, sm, m, a)
        new_h = torch.zeros(N, H, dk, dtype=F64)
        O.hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], v, d(rel_msg), a, new_h)
        Wa = d(torch.sigmoid(skip) * a_lin)
        out = lin(Wa, new_h.view(N, Dout)).view(N, Dout)
        torch.testing.assert_close(out, ref.detach())
        # backward
        g_newh, g_Wa = torch.zeros(N, Dout, dtype=F64), torch.zeros_like(Wa)
        O.backward_rgnn_relational_matmul_no_scatter_gather_list(offs, Wa.transpose(2, 3).contiguous(), new_h.view(N, Dout), go.view(N, 1, Dout), g_newh, g_Wa)
        g_v, g_msgW, g_a = torch.zeros_like(v), torch.zeros_like(d(rel_msg)), torch.zeros(E, H, dtype=F64)
        O.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(
            s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], v, d(rel_msg).transpose(2, 3).contiguous(), a, new_h, g_v, g_msgW, g_a, g_newh.view(N, H, dk))
        g_score, g_mu, tmp = torch.zeros(E, H, dtype=F64), torch.zeros(R, H, dtype=F64), torch.zeros(N, H, dtype=F64)
        O.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(*idx, score, a, g_a, mu, g_score, g_mu, tmp)
        g_k, g_q, g_att = torch.zeros_like(k), torch.zeros_like(q), torch.zeros_like(d(rel_att))
        if fused:
            dummy = torch.zeros(0, dtype=torch.int64)
            O.backward_hgt_full_graph_hetero_attention_ops_coo(dummy, dummy, dummy, dummy, *idx, g_att, d(rel_att).transpose(2, 3).contiguous(), k, q, inner, g_score, g_k, g_q)
        else:
            g_inner = torch.zeros_like(inner)
            O.backward_inner_product_right_node_separatecoo({}, 0, s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], inner, k, g_score, g_inner, g_k)
            O.backward_rgnn_relational_matmul(dd, 0, d(rel_att).transpose(2, 3).contiguous(), q, g_inner, g_q, g_att, False)
        g_h = torch.zeros(N, Din, dtype=F64)
        gWs = []
        for W, gx in ((k_lin, g_k), (q_lin, g_q), (v_lin, g_v)):
            gW = torch.zeros_like(d(W))
            O.backward_rgnn_relational_matmul_no_scatter_gather_list(offs, d(W).transpose(2, 3).contiguous(), d(h), gx.reshape(N, 1, Dout), g_h, gW)
            gWs.append(gW)
        torch.testing.assert_close(g_h, grads_ref[0])
        for got, want in zip(gWs, grads_ref[1:4]):
            torch.testing.assert_close(got, want)
        torch.testing.assert_close(g_Wa * torch.sigmoid(d(skip)), grads_ref[4])
        torch.testing.assert_close(g_att, grads_ref[5])
        torch.testing.assert_close(g_msgW, grads_ref[6])
        torch.testing.assert_close(g_mu / dk ** 0.5, grads_ref[7])


@pytest.mark.parametrize("R", [4, 104])
def test_rgcn_oracle_on_the_aifb_shaped_graph(R):
    """BASELINE.json configs[0] (C1) is the CPU path: one RGCN layer, feat 16, on the AIFB-shaped graph (4 relations as
    BASELINE.json states, 104 as the reference observes for DGL's AIFB).  The layer oracle (the reference's op
    composition, RGCN/RGCN.py:264-350) runs on it and agrees with an independent dense formulation -- one [N,N]
    normalised adjacency per relation -- in value and in the input / weight gradients."""
    from het_amd.graph import HetGraph
    from het_amd.synth import make_aifb_like
    from oracle import layers as OL
    g = HetGraph.from_integrated_coo(make_aifb_like(R), full=False)
    s = g.get_separate_coo_original()
    N, E, K, D = g.get_num_nodes(), g.get_num_edges(), 16, 16
    assert (N, E) == (8285, 58086)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(N, K, generator=gen, dtype=torch.float64).requires_grad_(True)
    W = (torch.randn(R, K, D, generator=gen, dtype=torch.float64) * 0.3).requires_grad_(True)
    norm = torch.rand(E, 1, generator=gen, dtype=torch.float64)
    go = torch.randn(N, D, generator=gen, dtype=torch.float64)
    out = OL.rgcn_layer(x, W, norm, s["rel_ptrs"], s["row_indices"], s["col_indices"], N)
    gx, gW = torch.autograd.grad(out, [x, W], go)
    # independent formulation: sparse [N,N] adjacency per relation with the norms as values, ret = SUM_r A_r x W_r
    ref = torch.zeros(N, D, dtype=torch.float64)
    for r in range(R):
        a, b = int(s["rel_ptrs"][r]), int(s["rel_ptrs"][r + 1])
        A = torch.sparse_coo_tensor(torch.stack([s["col_indices"][a:b], s["row_indices"][a:b]]), norm[a:b, 0], (N, N)).coalesce()
        ref = ref + torch.sparse.mm(A, x @ W[r])
    gx2, gW2 = torch.autograd.grad(ref, [x, W], go)
    torch.testing.assert_close(out, ref)
    torch.testing.assert_close(gx, gx2)
    torch.testing.assert_close(gW, gW2)


# ---------------------------------------------------------------- reference_literal (Q3, Q4, Q6, Q7)
def reference_literal_distances(gold):
    """Relative L2 distance || literal - intended || / || intended || per output of the ops whose CUDA code deviates
    deterministically from the reference's own stated intent, on the shipped ogbn_mag_0.1 topology (typed view for the HGT ops,
    one-id-space view for RGCN), fp64, seeded inputs.  Returns [(quirk, op, output, distance, note)]."""
    from tests.golden import recipe
    from het_amd.synth import IntegratedCOO
    g = torch.Generator().manual_seed(123)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=F64)  # noqa: E731
    out = []

    def dist(lit, ref):
        return float((lit - ref).norm() / ref.norm())

    # -- RGCN (a8, a9) on the one-id-space graph
    row, col, rel, eids, n, R = recipe.integrated_coo(gold["coo"])
    G_ = HetGraph.from_integrated_coo(IntegratedCOO(n, R, torch.tensor([0, n]), row, col, rel, eids))
    s = G_.get_separate_coo_original()
    rp, r_, c_, e_ = s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"]
    E, K, D = e_.numel(), 16, 16
    x, Wt, norm, go = rnd(n, K), rnd(R, D, K), torch.rand(E, 1, generator=g, dtype=F64), rnd(n, D)
    res = {}
    for lit in (False, True):
        gx, gW = torch.zeros(n, K, dtype=F64), torch.zeros(R, K, D, dtype=F64)
        O.backward_rgcn_layer1_separate_coo(rp, e_, r_, c_, x, Wt, norm, None, gx, go, gW, reference_literal=lit)
        res[lit] = (gx, gW)
    out.append(("Q3", "backward_rgcn_layer1_separate_coo", "grad_x", dist(res[True][0], res[False][0]), "gradout gathered by src, scattered to dst"))
    out.append(("Q3", "backward_rgcn_layer1_separate_coo", "grad_W", dist(res[True][1], res[False][1]) if float(res[False][1].norm()) else 0.0, "correct as coded"))
    ss = G_.get_separate_unique_node_indices_single_sided()
    d = {"rel_ptrs_row": ss["rel_ptrs_row"], "node_indices_row": ss["node_indices_row"]}
    feat, enorm = rnd(ss["node_indices_row"].numel(), D), torch.rand(E, generator=g, dtype=F64)
    res = {}
    for lit in (False, True):
        ret = torch.empty(n, D, dtype=F64)
        O.rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(e_, rp, r_, c_, d, feat, enorm, ret, False, reference_literal=lit)
        res[lit] = ret
    out.append(("Q4", "rgcn_node_mean_aggregation_compact_as_of_node_separate_coo", "ret", dist(res[True], res[False]), "compact tensor indexed by raw src id"))

    # -- HGT (a10, a11) on the typed view
    trow, tcol, trel, off = recipe.typed_coo(gold["coo"])
    N = int(off[-1])
    Gt = HetGraph.from_integrated_coo(IntegratedCOO(N, 6, off, trow, tcol, trel, torch.arange(trow.numel())))
    s = Gt.get_separate_coo_original()
    rp, r_, c_, e_ = s["rel_ptrs"], s["row_indices"], s["col_indices"], s["eids"]
    H, dk = 2, 4
    score, mu = rnd(E, H) * 0.5, torch.rand(6, H, generator=g, dtype=F64) + 0.5
    res = {}
    for lit in (False, True):
        sm, m, a = torch.zeros(N, H, dtype=F64), torch.zeros(E, H, dtype=F64), torch.zeros(E, H, dtype=F64)
        O.hgt_full_graph_edge_softmax_ops_separate_coo(r_, c_, e_, rp, score, mu, sm, m, a, reference_literal=lit)
        res[lit] = (sm, m, a)
    out.append(("Q6", "hgt_full_graph_edge_softmax_ops_separate_coo", "sum", dist(res[True][0], res[False][0]), "denominator keyed by src; last edge skipped"))
    out.append(("Q6", "hgt_full_graph_edge_softmax_ops_separate_coo", "m", dist(res[True][1], res[False][1]), "last edge's row left as given (zeros here)"))
    out.append(("Q6", "hgt_full_graph_edge_softmax_ops_separate_coo", "a", dist(res[True][2], res[False][2]), "softmax over the OUT-edges of the source"))
    a, grad_a = res[False][2], rnd(E, H)
    res = {}
    for lit in (False, True):
        gs, gmu, tmp = torch.zeros(E, H, dtype=F64), torch.zeros(6, H, dtype=F64), torch.zeros(N, H, dtype=F64)
        O.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(r_, c_, e_, rp, score, a, grad_a, mu, gs, gmu, tmp, reference_literal=lit)
        res[lit] = (gs, gmu)
    out.append(("Q7", "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo", "grad_score", dist(res[True][0], res[False][0]),
                "32 positions x head 0 visited (by every block); the rest left as given (zeros here)"))
    out.append(("Q7", "backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo", "grad_mu", dist(res[True][1], res[False][1]),
                "same positions, added H*G times, without * score"))
    v, Wt, go = rnd(N, H, dk), rnd(6, H, dk, dk), rnd(N, H, dk)
    res = {}
    for lit in (False, True):
        gv, gW, ga = torch.zeros(N, H, dk, dtype=F64), torch.zeros(6, H, dk, dk, dtype=F64), torch.zeros(E, H, dtype=F64)
        O.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(rp, e_, r_, c_, v, Wt, a, None, gv, gW, ga, go, reference_literal=lit)
        res[lit] = (gv, gW, ga)
    out.append(("Q3", "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo", "grad_v", dist(res[True][0], res[False][0]), "transposed direction"))
    out.append(("Q3", "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo", "grad_W", dist(res[True][1], res[False][1]), "correct as coded"))
    out.append(("Q3", "backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo", "grad_a", dist(res[True][2], res[False][2]), "transposed direction, extra factor a"))
    return out


def test_reference_literal_distance_on_the_shipped_topology(golden_mag_full, capsys):
    """What "we differ from the CUDA code on purpose" amounts to, as numbers (DESIGN.md section 3 quotes this table): the
    literal readings are far from the intended semantics wherever SURVEY.md section 9 says they deviate (relative L2 distance
    of order 1) and identical where it says the code is correct as written."""
    rows = reference_literal_distances(golden_mag_full)
    with capsys.disabled():
        print()
        for q, op, name, dv, note in rows:
            print(f"  {q}  {op[:58]:58s} {name:10s} {dv:9.3e}  {note}")
    by = {(q, op, name): dv for q, op, name, dv, _ in rows}
    for (q, op, name), dv in by.items():
        if name == "grad_W":
            assert dv == 0.0, (op, name, dv)
        elif name == "m":
            assert 0.0 < dv < 1e-1, (op, name, dv)  # one row of E
        elif name == "sum":
            assert dv > 0.05, (op, name, dv)  # (the shipped graph holds every relation beside its reverse: in- and out-sums are alike)
        else:
            assert dv > 0.3, (op, name, dv)


def test_reference_literal_small_cases_by_hand():
    """The literal readings on a graph small enough to check by hand: 3 nodes, edges 0->1, 0->2, 1->2 in one relation."""
    rp, row, col, eids = torch.tensor([0, 3]), torch.tensor([0, 0, 1]), torch.tensor([1, 2, 2]), torch.arange(3)
    # Q6: denominators keyed by the source, last position skipped
    score, mu = torch.zeros(3, 1, dtype=F64), torch.ones(1, 1, dtype=F64)
    sm, m, a = torch.zeros(3, 1, dtype=F64), torch.full((3, 1), -7.0, dtype=F64), torch.full((3, 1), -7.0, dtype=F64)
    O.hgt_full_graph_edge_softmax_ops_separate_coo(row, col, eids, rp, score, mu, sm, m, a, reference_literal=True)
    assert sm.flatten().tolist() == [2.0, 0.0, 0.0] and m.flatten().tolist() == [1.0, 1.0, -7.0] and a.flatten().tolist() == [0.5, 0.5, -7.0]
    O.hgt_full_graph_edge_softmax_ops_separate_coo(row, col, eids, rp, score, mu, sm, m, a)
    assert sm.flatten().tolist() == [0.0, 1.0, 2.0] and a.flatten().tolist() == [1.0, 0.5, 0.5]
    # Q3: x -> W -> scattered to dst in the forward; the literal backward sends gradout[src] to grad_x[dst]
    x, Wt, norm, go = torch.eye(3, dtype=F64), torch.eye(3, dtype=F64).unsqueeze(0), torch.ones(3, 1, dtype=F64), torch.tensor([[1., 0, 0], [0, 2., 0], [0, 0, 4.]], dtype=F64)
    gx, gW = torch.zeros(3, 3, dtype=F64), torch.zeros(1, 3, 3, dtype=F64)
    O.backward_rgcn_layer1_separate_coo(rp, eids, row, col, x, Wt, norm, None, gx, go, gW, reference_literal=True)
    assert torch.equal(gx, torch.stack([torch.zeros(3, dtype=F64), go[0], go[0] + go[1]]))
    gx.zero_(); gW.zero_()
    O.backward_rgcn_layer1_separate_coo(rp, eids, row, col, x, Wt, norm, None, gx, go, gW)
    assert torch.equal(gx, torch.stack([go[1] + go[2], go[2], torch.zeros(3, dtype=F64)]))
    # Q7: head 0 of the first 32 positions only, each H * G = 2 * 1 times
    score, mu = torch.ones(3, 2, dtype=F64), torch.ones(1, 2, dtype=F64)
    a, ga = torch.full((3, 2), 0.5, dtype=F64), torch.ones(3, 2, dtype=F64)
    gs, gmu, tmp = torch.full((3, 2), -7.0, dtype=F64), torch.zeros(1, 2, dtype=F64), torch.zeros(3, 2, dtype=F64)
    O.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(row, col, eids, rp, score, a, ga, mu, gs, gmu, tmp, reference_literal=True)
    assert tmp.tolist() == [[2.0, 0.0], [1.0, 0.0], [0.0, 0.0]]          # keyed by src, H*G = 2 times 0.5 per position
    assert gs.tolist() == [[-0.5, -7.0], [-0.5, -7.0], [0.0, -7.0]]      # (1 - tmp[src]) * 0.5 * mu; head 1 untouched
    assert gmu.tolist() == [[2 * (-0.5 - 0.5 + 0.0), 0.0]]


if __name__ == "__main__":  # python -m tests.test_oracle: the table DESIGN.md section 3 quotes
    from tests.conftest import load_golden
    for q, op, name, dv, note in reference_literal_distances(load_golden("mag01_full.npz")):
        print(f"{q}  {op:84s} {name:10s} {dv:9.3e}  {note}")
