"""GPU parity: every torch_hrt op (served by libhet_amd.so through the C ABI)
against the fp64 CPU oracle on the same seeded inputs."""
import pytest
import torch

from oracle import ops as O
from tests.util import assert_close, cpu, mag_graph, random_graph, to64

pytestmark = pytest.mark.gpu
DEV = "cuda"


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


def _dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


# ---------------------------------------------------------------- segment GEMM
SHAPES = [  # (H, K, D, in1head)
    (1, 16, 16, True),    # AIFB-sized RGCN-style projection (generic kernel)
    (4, 64, 16, True),    # RGAT C3 projection (MFMA path, X = 64)
    (2, 64, 64, True),    # X = 128 MFMA
    (1, 32, 32, True),    # smallest MFMA shape
    (4, 16, 1, False),    # attention vector: el = <feat, attn> per head
    (3, 7, 5, False),     # odd everything, per-head input
    (3, 7, 5, True),      # odd, shared input
    (8, 8, 8, False),     # HGT relation_att shape
    (4, 64, 1, True),     # mulfirst: x . (W . attn) with one shared input head
    (2, 16, 1, True),
    (8, 32, 1, True),
    (4, 128, 32, True),   # feat = 128 (BASELINE.json configs[4]): K = X = 128, dW in 64-wide blocks
    (1, 128, 64, True),
    (2, 32, 64, True),    # X = 128 from K = 32
    (4, 256, 64, True),   # K = X = 256: 128-wide slabs of the weight
    (1, 256, 32, True),   # K = 256 only
    (2, 64, 128, True),   # X = 256 only
    (1, 64, 64, False),   # one head, per-head input form (HGT relation_att at --num_heads 1): same rows as the shared form
    (1, 8, 8, False),
]


@pytest.mark.parametrize("H,Kd,D,in1head", SHAPES)
@pytest.mark.parametrize("kind", [0, 1])
def test_rgnn_relational_matmul_fwd_bwd(K, H, Kd, D, in1head, kind):
    g = random_graph(seed=11)
    s = g.get_separate_coo_original()
    R, N, E = g.get_num_rels(), g.get_num_nodes(), g.get_num_edges()
    gen = torch.Generator().manual_seed(5)
    W = torch.randn(R, H, Kd, D, generator=gen)
    x = torch.randn(N, Kd, generator=gen) if in1head else torch.randn(N, H, Kd, generator=gen)
    if kind == 0:
        d = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"],
             "separate_coo_eids": torch.randperm(E, generator=gen)}
        nout = E
    else:
        u = g.get_separate_unique_node_indices_single_sided()
        d = {"unique_srcs_and_dests_rel_ptrs": u["rel_ptrs_row"], "unique_srcs_and_dests_node_indices": u["node_indices_row"]}
        nout = int(u["rel_ptrs_row"][-1])
    ref = torch.zeros(nout, H, D, dtype=torch.float64)
    O.rgnn_relational_matmul(d, kind, to64(W), to64(x), ref, in1head)
    ret = torch.full((nout, H, D), float("nan"), device=DEV)
    K.rgnn_relational_matmul(_dev(d), kind, W.to(DEV), x.to(DEV), ret, in1head)
    assert_close(ret, ref, what="ret")

    go = torch.randn(nout, H, D, generator=gen)
    gx_ref, gW_ref = torch.zeros_like(to64(x)), torch.zeros_like(to64(W))
    O.backward_rgnn_relational_matmul(d, kind, to64(W).transpose(2, 3).contiguous(), to64(x), to64(go), gx_ref, gW_ref, in1head)
    gx, gW = torch.zeros_like(x, device=DEV), torch.zeros_like(W, device=DEV)
    K.backward_rgnn_relational_matmul(_dev(d), kind, W.transpose(2, 3).contiguous().to(DEV), x.to(DEV), go.to(DEV), gx, gW, in1head)
    assert_close(gx, gx_ref, what="grad_x")
    assert_close(gW, gW_ref, what="grad_W")


@pytest.mark.parametrize("H,Kd,D", [(4, 64, 16), (4, 64, 1), (3, 7, 5)])
def test_matmul_backward_kind1_duplicate_rows(K, H, Kd, D):
    """The reference-named a2 with a kind-1 list that holds a node MORE than once inside a relation: every occurrence
    contributes (float atomics, as the reference's compact backward, RGNN/my_shmem_sgemm_func.cu.h:711-776).  The plain
    read-modify-write schedule is only taken when the caller states the list is unique (HET_ACC_DISTINCT_ROWS)."""
    gen = torch.Generator().manual_seed(8)
    N, R = 50, 3
    rp = torch.tensor([0, 300, 300, 700])
    nodes = torch.randint(0, N, (700,), generator=gen)  # ~6 occurrences of every node per relation
    d = {"unique_srcs_and_dests_rel_ptrs": rp, "unique_srcs_and_dests_node_indices": nodes}
    W = torch.randn(R, H, Kd, D, generator=gen)
    x = torch.randn(N, Kd, generator=gen)
    go = torch.randn(700, H, D, generator=gen)
    gx_ref, gW_ref = torch.zeros_like(to64(x)), torch.zeros_like(to64(W))
    O.backward_rgnn_relational_matmul(d, 1, to64(W).transpose(2, 3).contiguous(), to64(x), to64(go), gx_ref, gW_ref, True)
    gx, gW = torch.zeros_like(x, device=DEV), torch.zeros_like(W, device=DEV)
    K.backward_rgnn_relational_matmul(_dev(d), 1, W.transpose(2, 3).contiguous().to(DEV), x.to(DEV), go.to(DEV), gx, gW, True)
    assert_close(gx, gx_ref, what="grad_x")
    assert_close(gW, gW_ref, what="grad_W")


def test_matmul_gather_equals_scatter_list(K):
    """el = feat_edge . attn with node_indices and eids the SAME tensor
    (the reference dispatches on data_ptr equality, RGNNOps.inc.h:253)."""
    g = random_graph(seed=12)
    s = g.get_separate_coo_original()
    R, E, H, D = g.get_num_rels(), g.get_num_edges(), 4, 16
    feat = torch.randn(E, H, D)
    attn = torch.randn(R, H, D, 1)
    eids = s["eids"]
    d = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": eids, "separate_coo_eids": eids}
    ref = torch.zeros(E, H, 1, dtype=torch.float64)
    O.rgnn_relational_matmul(d, 0, to64(attn), to64(feat), ref, False)
    ret = torch.empty(E, H, 1, device=DEV)
    e_dev = eids.to(DEV)
    K.rgnn_relational_matmul({"separate_coo_rel_ptrs": s["rel_ptrs"].to(DEV), "separate_coo_node_indices": e_dev,
                              "separate_coo_eids": e_dev}, 0, attn.to(DEV), feat.to(DEV), ret, False)
    assert_close(ret, ref)


@pytest.mark.parametrize("H,Kd,D,per_head", [(1, 64, 64, False), (1, 16, 16, False), (4, 16, 1, True), (3, 6, 5, False), (8, 8, 8, True),
                                             (1, 256, 32, False), (1, 256, 64, False), (2, 256, 32, False), (4, 256, 16, False), (1, 128, 256, False)])
def test_matmul_no_scatter_gather(K, H, Kd, D, per_head):
    offsets = torch.tensor([0, 130, 130, 131, 700, 1023])
    T, n = 5, 1023
    gen = torch.Generator().manual_seed(3)
    W = torch.randn(T, H, Kd, D, generator=gen)
    x = torch.randn(n, H, Kd, generator=gen) if per_head else torch.randn(n, Kd, generator=gen)
    ref = torch.zeros(n, H, D, dtype=torch.float64)
    O.rgnn_relational_matmul_no_scatter_gather_list(offsets, to64(W), to64(x), ref)
    ret = torch.full((n, H, D), float("nan"), device=DEV)
    K.rgnn_relational_matmul_no_scatter_gather_list(offsets.to(DEV), W.to(DEV), x.to(DEV), ret)
    assert_close(ret, ref)
    go = torch.randn(n, H, D, generator=gen)
    gx_ref, gW_ref = torch.zeros_like(to64(x)), torch.zeros_like(to64(W))
    O.backward_rgnn_relational_matmul_no_scatter_gather_list(offsets, to64(W).transpose(2, 3).contiguous(), to64(x), to64(go), gx_ref, gW_ref)
    gx, gW = torch.zeros_like(x, device=DEV), torch.zeros_like(W, device=DEV)
    K.backward_rgnn_relational_matmul_no_scatter_gather_list(offsets.to(DEV), W.transpose(2, 3).contiguous().to(DEV), x.to(DEV), go.to(DEV), gx, gW)
    assert_close(gx, gx_ref, what="grad_x")
    assert_close(gW, gW_ref, what="grad_W")
    # the layers' "=" form of the same op (plain stores into uninitialised buffers; a 256-wide input gradient from a
    # narrow output runs as column slabs of the weight, which have to compose: ADVICE r2, seg_gemm_mfma.hip)
    import het_amd.kernels as k
    gx2, gW2 = torch.full_like(gx, float("nan")), torch.full_like(gW, float("nan"))
    k.matmul_no_scatter_gather_backward(offsets.to(DEV), W.transpose(2, 3).contiguous().to(DEV), x.to(DEV), go.to(DEV), gx2, gW2,
                                        accumulate=False)
    assert_close(gx2, gx_ref, what="grad_x (=)")
    assert_close(gW2, gW_ref, what="grad_W (=)")


def test_written_tensors_get_a_new_version(K):
    """An op bumps the version counter of every tensor its schema marks as written (het_amd/kernels.py::_op, csrc/torch_export.cpp::
    will_write): the dispatcher does not do that for custom ops and the library writes through raw pointers, so without it a cache
    keyed by (data_ptr, numel, _version) could serve a copy of a buffer that another op has refilled in place."""
    offsets = torch.tensor([0, 40, 100]).to(DEV)
    gen = torch.Generator().manual_seed(4)
    W, x = torch.randn(2, 1, 32, 32, generator=gen).to(DEV), torch.randn(100, 32, generator=gen).to(DEV)
    ret = torch.zeros(100, 1, 32, device=DEV)
    v_ret, v_x, v_W = ret._version, x._version, W._version
    K.rgnn_relational_matmul_no_scatter_gather_list(offsets, W, x, ret)
    assert ret._version > v_ret and x._version == v_x and W._version == v_W
    gx, gW = torch.zeros_like(x), torch.zeros_like(W)
    v = (gx._version, gW._version, ret._version)
    K.backward_rgnn_relational_matmul_no_scatter_gather_list(offsets, W.transpose(2, 3).contiguous(), x, ret, gx, gW)
    assert gx._version > v[0] and gW._version > v[1] and ret._version == v[2]


def test_matmul_empty(K):
    """No rows at all, and a relation list made only of empty relations."""
    rp = torch.zeros(4, dtype=torch.int64, device=DEV)
    e = torch.zeros(0, dtype=torch.int64, device=DEV)
    W = torch.randn(3, 2, 8, 4, device=DEV)
    x = torch.randn(5, 8, device=DEV)
    ret = torch.zeros(0, 2, 4, device=DEV)
    K.rgnn_relational_matmul({"separate_coo_rel_ptrs": rp, "separate_coo_node_indices": e, "separate_coo_eids": e}, 0, W, x, ret, True)
    torch.cuda.synchronize()


# ---------------------------------------------------------------- fused GAT
def _gat_case(g, kind, H, D, seed):
    from tests.test_oracle import _gat_dict, _gat_sizes
    s = g.get_separate_coo_original()
    ns, nd = _gat_sizes(g, kind)
    gen = torch.Generator().manual_seed(seed)
    feat = torch.randn(ns, H, D, generator=gen)
    el = torch.randn(ns, H, generator=gen)
    er = torch.randn(nd, H, generator=gen)
    go = torch.randn(g.get_num_nodes(), H, D, generator=gen)
    df, db = _gat_dict(g, kind)
    return s, feat, el, er, go, df, db


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("H,D,n", [(4, 16, 300), (1, 64, 300), (3, 5, 300), (8, 8, 300), (2, 2, 300), (4, 16, 24)])
def test_fused_gat_separate_coo(K, plan_mode, kind, H, D, n):
    # n = 24: ~50 edges per (relation, node) row, rows with several hundred -- the wave-per-item compact backward and
    # its split (atomic) rows; n = 300: ~4 per row -- the lane-group-per-item one
    g = random_graph(seed=21, n=n, r=4, e=5000)
    s, feat, el, er, go, df, db = _gat_case(g, kind, H, D, seed=9)
    N, E, slope = g.get_num_nodes(), g.get_num_edges(), 0.2
    idx = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
    sm_r, ex_r, ret_r = (torch.empty(N, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64),
                         torch.empty(N, H, D, dtype=torch.float64))
    O.relational_fused_gat_separate_coo(*idx, kind, df, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r, slope)
    gf_r, gl_r, gr_r = torch.zeros_like(to64(feat)), torch.zeros_like(to64(el)), torch.zeros_like(to64(er))
    O.backward_relational_fused_gat_separate_coo(*idx, kind, db, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r,
                                                 to64(go), gf_r, gl_r, gr_r, slope)
    didx = tuple(t.to(DEV) for t in idx)
    # outputs start as garbage: the op must overwrite sum / exp / ret
    sm, ex, ret = (torch.full((N, H), 7.0, device=DEV), torch.full((E, H), 7.0, device=DEV), torch.full((N, H, D), 7.0, device=DEV))
    f, l, r_ = feat.to(DEV), el.to(DEV), er.to(DEV)
    K.relational_fused_gat_separate_coo(*didx, kind, _dev(df), f, l, r_, sm, ex, ret, slope)
    assert_close(ex, ex_r, what="exp")
    assert_close(sm, sm_r, what="sum")
    assert_close(ret, ret_r, what="ret")
    fill = float("nan") if kind == 0 else 0.0  # kind 0 gradients are overwritten, the others accumulated
    gf, gl, gr = (torch.full_like(f, fill), torch.full_like(l, fill), torch.full_like(r_, fill))
    K.backward_relational_fused_gat_separate_coo(*didx, kind, _dev(db), f, l, r_, sm, ex, ret, go.to(DEV), gf, gl, gr, slope)
    assert_close(gf, gf_r, what="grad_feat")
    assert_close(gl, gl_r, what="grad_el")
    assert_close(gr, gr_r, what="grad_er")


@pytest.mark.parametrize("H,D", [(4, 16), (2, 4)])
def test_gat_backward_streams_the_sorted_exp_its_forward_left(K, H, D):
    """The reference-named pair called on its own (kind 0, grouped kernels): the forward leaves a destination-sorted copy of exp,
    the backward streams it when (exp, el, er) come back untouched -- and falls back to gathering by edge id after an in-place
    edit of exp, with other el / er tensors, or with another exp: every case against the oracle (what the backward computes
    from the tensors it is GIVEN), hit / miss asserted through the Python registration's counter."""
    import het_amd.kernels as k
    import het_amd.plan as plan
    old = plan.enabled
    plan.enabled = True
    plan.clear()
    try:
        g = random_graph(seed=23, n=300, r=4, e=5000)
        s, feat, el, er, go, df, db = _gat_case(g, 0, H, D, seed=13)
        N, E, slope = g.get_num_nodes(), g.get_num_edges(), 0.2
        idx = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
        didx = tuple(t.to(DEV) for t in idx)
        f, l, r_, god = feat.to(DEV), el.to(DEV), er.to(DEV), go.to(DEV)
        sm, ex, ret = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(N, H, D, device=DEV)

        def oracle_backward(ex_t, l_t, r_t):
            gf_r, gl_r, gr_r = torch.zeros_like(to64(feat)), torch.zeros_like(to64(el)), torch.zeros_like(to64(er))
            O.backward_relational_fused_gat_separate_coo(*idx, 0, db, to64(feat), to64(l_t), to64(r_t), to64(sm), to64(ex_t), to64(ret),
                                                         to64(go), gf_r, gl_r, gr_r, slope)
            return gf_r, gl_r, gr_r

        def run_backward(ex_t, l_t, r_t, expect_hit):
            before = k.sorted_stream_hits
            gf, gl, gr = torch.full_like(f, float("nan")), torch.full_like(l, float("nan")), torch.full_like(r_, float("nan"))
            K.backward_relational_fused_gat_separate_coo(*didx, 0, {}, f, l_t, r_t, sm, ex_t, ret, god, gf, gl, gr, slope)
            if not k.COMPILED_LIB:  # (the compiled registration keeps its own cache: values only)
                assert (k.sorted_stream_hits - before == 1) == expect_hit, (k.sorted_stream_hits - before, expect_hit)
            for got, want, what in zip((gf, gl, gr), oracle_backward(ex_t, l_t, r_t), ("grad_feat", "grad_el", "grad_er")):
                assert_close(got, want, what=what)

        K.relational_fused_gat_separate_coo(*didx, 0, {}, f, l, r_, sm, ex, ret, slope)
        run_backward(ex, l, r_, True)            # the same tensors come back: streamed
        run_backward(ex, l, r_, True)            # (a second backward of the same forward: still there)
        run_backward(ex.clone(), l, r_, False)   # another exp tensor (same values): gathers
        run_backward(ex, l.clone(), r_, False)   # other attention terms: gathers
        ex.mul_(0.5)                             # in-place edit (the version counter moves): gathers, and the new values count
        sm.mul_(0.5)
        run_backward(ex, l, r_, False)
        K.relational_fused_gat_separate_coo(*didx, 0, {}, f, l, r_, sm, ex, ret, slope)  # a new forward into the same buffers
        run_backward(ex, l, r_, True)
    finally:
        plan.enabled = old
        plan.clear()


@pytest.mark.parametrize("fold,bias", [(False, False), (True, True)])
@pytest.mark.parametrize("H,D,n,e", [(4, 16, 300, 5000), (1, 64, 300, 5000), (8, 8, 40, 9000), (2, 4, 300, 5000), (4, 16, 12, 9000), (4, 32, 300, 700)])
def test_rgat_compact_passes(K, H, D, n, e, fold, bias):
    """het_rgat_aggregate_compact / het_rgat_backward_compact (no exp tensor between the passes) against the oracle's
    relational_fused_gat_separate_coo pair with CompactAsOfNodeKind 4.  n = 12 / 40: (relation, source) rows with
    hundreds of edges (pieces of long segments add atomically, hub destinations are split over work items);
    n = 300: a few edges per row (several whole segments per pack); e = 700: packs of a single short segment."""
    import het_amd.kernels as k
    g = random_graph(seed=27, n=n, r=4, e=e)
    s, feat, el, er, go, df, db = _gat_case(g, 4, H, D, seed=11)
    N, E, slope = g.get_num_nodes(), g.get_num_edges(), 0.2
    idx = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
    sm_r, ex_r, ret_r = (torch.empty(N, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64),
                         torch.empty(N, H, D, dtype=torch.float64))
    O.relational_fused_gat_separate_coo(*idx, 4, df, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r, slope)
    gf_r, gl_r, gr_r = torch.zeros_like(to64(feat)), torch.zeros_like(to64(el)), torch.zeros_like(to64(er))
    O.backward_relational_fused_gat_separate_coo(*idx, 4, db, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r,
                                                 to64(go), gf_r, gl_r, gr_r, slope)
    ss = g.get_separate_unique_node_indices_single_sided()
    R = g.get_num_rels()
    gen = torch.Generator().manual_seed(5)
    attn = torch.randn(R, H, D, generator=gen)
    if fold:  # grad_feat += grad_el (x) attn[r(row)]
        rel_of_row = torch.repeat_interleave(torch.arange(R), ss["rel_ptrs_row"][1:] - ss["rel_ptrs_row"][:-1])
        gf_r = gf_r + gl_r.unsqueeze(-1) * attn.double()[rel_of_row]
    # rows of every edge POSITION (eids of the test graphs are a permutation)
    srow = df["edata_idx_to_inverse_idx_row"][s["eids"]].contiguous().to(DEV)
    drow = df["edata_idx_to_inverse_idx_col"][s["eids"]].contiguous().to(DEV)
    grp = k.rgat_compact_groupings(s["col_indices"].to(DEV), srow, drow, N, feat.shape[0], er.shape[0])
    f, l, r_ = feat.to(DEV), el.to(DEV), er.to(DEV)
    sm, ret = torch.full((N, H), 7.0, device=DEV), torch.full((N, H, D), 7.0, device=DEV)
    k.rgat_aggregate_compact(grp, f, l, r_, sm, ret, slope)
    # `sum` of this entry point is the log-sum-exp of the destination (running-maximum softmax: include/het_amd.h); destinations
    # without in-edges keep 0
    has_in = torch.zeros(N, dtype=torch.bool)
    has_in[s["col_indices"]] = True
    assert_close(sm[has_in.to(DEV)], torch.log(sm_r[has_in]), what="log-sum-exp")
    assert float(sm[(~has_in).to(DEV)].abs().max() if (~has_in).any() else 0.0) == 0.0
    assert_close(ret, ret_r, what="ret")
    # h_inout: the rows are also added into a caller tensor (the first nh destinations), ret untouched elsewhere
    nh = N - 2
    h0 = torch.randn(nh, H * D, generator=gen)
    hio, ret2 = h0.to(DEV), torch.full((N, H, D), 7.0, device=DEV)
    k.rgat_aggregate_compact(grp, f, l, r_, sm, ret2, slope, h_inout=hio)
    assert_close(hio, h0.double() + ret_r.view(N, -1)[:nh], what="h_inout")
    assert_close(ret2[has_in.to(DEV)], ret_r[has_in], what="ret (destinations with in-edges)")
    gf, gl, gr = torch.full_like(f, float("nan")), torch.full_like(l, float("nan")), torch.full_like(r_, float("nan"))
    gb = torch.full((H * D,), float("nan"), device=DEV) if bias else None
    nb = N - 3
    k.rgat_backward_compact(grp, f, l, r_, sm, ret, go.to(DEV), gf, gl, gr, slope, fold_attn_l=attn.to(DEV) if fold else None,
                            row_rel_ptrs=ss["rel_ptrs_row"].to(DEV) if fold else None, grad_bias=gb, bias_rows=nb)
    assert_close(gf, gf_r, what="grad_feat")
    assert_close(gl, gl_r, what="grad_el")
    assert_close(gr, gr_r, what="grad_er")
    if bias:
        assert_close(gb, to64(go).view(N, -1)[:nb].sum(0), what="grad_bias")


@pytest.mark.parametrize("fold,bias", [(False, False), (True, True)])
@pytest.mark.parametrize("H,D,n,e", [(4, 16, 300, 5000), (1, 64, 300, 5000), (4, 16, 12, 9000), (4, 32, 300, 700), (4, 16, 40, 9000),
                                     (2, 32, 12, 9000), (1, 32, 40, 9000), (2, 16, 20, 9000), (2, 16, 300, 3000)])
def test_rgat_compact_run_sums(K, H, D, n, e, fold, bias):
    """het_rgat_aggregate_compact_runs / het_rgat_backward_compact_runs (grad_er from the sums the forward leaves per er row, no
    per-edge term) against the oracle's CompactAsOfNodeKind-4 pair.  The er rows are the distinct (relation, destination) pairs
    here -- the precondition of this form (include/het_amd.h).  n = 12 / 20: hubs (more than 256 in-edges: parked work items,
    several runs per hub); n = 40: destinations of 33 .. 256 in-edges (a pack of their own); n = 300: several destinations per pack."""
    import het_amd.kernels as k
    g = random_graph(seed=31, n=n, r=4, e=e)
    s = g.get_separate_coo_original()
    ss = g.get_separate_unique_node_indices_single_sided()
    N, E, R, slope = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels(), 0.2
    rel = torch.repeat_interleave(torch.arange(R), s["rel_ptrs"][1:] - s["rel_ptrs"][:-1])

    def rows_of(ptrs, nodes, ids):  # row of (relation, node) of every position in a unique (relation, node) list
        key = torch.repeat_interleave(torch.arange(R), ptrs[1:] - ptrs[:-1]) * N + nodes
        return torch.searchsorted(key, rel * N + ids).contiguous()
    srow_p = rows_of(ss["rel_ptrs_row"], ss["node_indices_row"], s["row_indices"])
    drow_p = rows_of(ss["rel_ptrs_col"], ss["node_indices_col"], s["col_indices"])
    assert bool((ss["node_indices_col"][drow_p] == s["col_indices"]).all())
    S_row, S_col = ss["node_indices_row"].numel(), ss["node_indices_col"].numel()
    gen = torch.Generator().manual_seed(13)
    feat, el, er = torch.randn(S_row, H, D, generator=gen), torch.randn(S_row, H, generator=gen), torch.randn(S_col, H, generator=gen)
    go, attn = torch.randn(N, H, D, generator=gen), torch.randn(R, H, D, generator=gen)
    # the oracle's maps are indexed by edge id
    m_row, m_col = torch.empty(E, dtype=torch.int64), torch.empty(E, dtype=torch.int64)
    m_row[s["eids"]], m_col[s["eids"]] = srow_p, drow_p
    df = {"edata_idx_to_inverse_idx_row": m_row, "edata_idx_to_inverse_idx_col": m_col}
    idx = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
    sm_r, ex_r, ret_r = (torch.empty(N, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64),
                         torch.empty(N, H, D, dtype=torch.float64))
    O.relational_fused_gat_separate_coo(*idx, 4, df, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r, slope)
    gf_r, gl_r, gr_r = torch.zeros_like(to64(feat)), torch.zeros_like(to64(el)), torch.zeros_like(to64(er))
    O.backward_relational_fused_gat_separate_coo(*idx, 4, df, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r,
                                                 to64(go), gf_r, gl_r, gr_r, slope)
    if fold:
        rel_of_row = torch.repeat_interleave(torch.arange(R), ss["rel_ptrs_row"][1:] - ss["rel_ptrs_row"][:-1])
        gf_r = gf_r + gl_r.unsqueeze(-1) * attn.double()[rel_of_row]
    grp = k.rgat_compact_groupings(s["col_indices"].to(DEV), srow_p.to(DEV), drow_p.to(DEV), N, S_row, S_col, rel_ptrs=s["rel_ptrs"].to(DEV),
                                   drow_nodes=ss["node_indices_col"].to(DEV), drow_rel_ptrs=ss["rel_ptrs_col"].to(DEV))
    assert grp[2] is None and grp[3] is not None
    f, l, r_ = feat.to(DEV), el.to(DEV), er.to(DEV)
    has_in = torch.zeros(N, dtype=torch.bool)
    has_in[s["col_indices"]] = True
    sm, ret = torch.full((N, H), 7.0, device=DEV), torch.full((N, H, D), 7.0, device=DEV)
    runs = k.rgat_aggregate_compact(grp, f, l, r_, sm, ret, slope, num_rels=R)
    assert_close(sm[has_in.to(DEV)], torch.log(sm_r[has_in]), what="log-sum-exp")
    assert float(sm[(~has_in).to(DEV)].abs().max() if (~has_in).any() else 0.0) == 0.0
    assert_close(ret, ret_r, what="ret")
    # the run sums themselves: q_rows * exp(q_ref) = SUM_e exp(s_e) dl_e feat[srow_e] over the run
    z = to64(el)[srow_p] + to64(er)[drow_p]
    wd = torch.exp(torch.nn.functional.leaky_relu(z, slope)) * torch.where(z > 0, 1.0, slope)
    q_ref = torch.zeros(S_col, H, dtype=torch.float64).index_add_(0, drow_p, wd)
    Q_ref = torch.zeros(S_col, H, D, dtype=torch.float64).index_add_(0, drow_p, wd.unsqueeze(-1) * to64(feat)[srow_p])
    sc = torch.exp(runs[2].double().cpu())
    assert_close(runs[1].double().cpu() * sc, q_ref, what="q_sum")
    assert_close(runs[0].double().cpu() * sc.unsqueeze(-1), Q_ref, what="q_rows")
    if D == 16:
        # el_c that IS <feat_c, attn_l[relation of the row]> (the layer's case): with attn_l and the relation pointers of the rows the
        # pass forms it from the rows it gathers -- the el_c handed over is not read (NaN here) -- same sums as the gathered form
        rel_of_row_ = torch.repeat_interleave(torch.arange(R), ss["rel_ptrs_row"][1:] - ss["rel_ptrs_row"][:-1])
        el2 = (feat * attn[rel_of_row_]).sum(-1).contiguous()
        smA, retA = torch.full((N, H), 7.0, device=DEV), torch.full((N, H, D), 7.0, device=DEV)
        runsA = k.rgat_aggregate_compact(grp, f, el2.to(DEV), r_, smA, retA, slope, num_rels=R)
        smB, retB = torch.full((N, H), 7.0, device=DEV), torch.full((N, H, D), 7.0, device=DEV)
        runsB = k.rgat_aggregate_compact(grp, f, torch.full_like(l, float("nan")), r_, smB, retB, slope, num_rels=R,
                                         attn_l=attn.to(DEV), feat_rel_ptrs=ss["rel_ptrs_row"].to(DEV))
        assert_close(retB, retA.double().cpu(), what="ret (el from the row)")
        assert_close(smB, smA.double().cpu(), what="lse (el from the row)")
        scA, scB = torch.exp(runsA[2].double().cpu()), torch.exp(runsB[2].double().cpu())
        assert_close(runsB[1].double().cpu() * scB, runsA[1].double().cpu() * scA, what="q_sum (el from the row)")
        assert_close(runsB[0].double().cpu() * scB.unsqueeze(-1), runsA[0].double().cpu() * scA.unsqueeze(-1), what="q_rows (el from the row)")
    nh = N - 2
    h0 = torch.randn(nh, H * D, generator=gen)
    hio, ret2, sm2 = h0.to(DEV), torch.full((N, H, D), 7.0, device=DEV), torch.full((N, H), 7.0, device=DEV)
    k.rgat_aggregate_compact(grp, f, l, r_, sm2, ret2, slope, h_inout=hio, num_rels=R)
    assert_close(hio, h0.double() + ret_r.view(N, -1)[:nh], what="h_inout")
    assert_close(ret2[has_in.to(DEV)], ret_r[has_in], what="ret (destinations with in-edges)")
    gf, gl, gr = torch.full_like(f, float("nan")), torch.full_like(l, float("nan")), torch.full_like(r_, float("nan"))
    gb = torch.full((H * D,), float("nan"), device=DEV) if bias else None
    nb = N - 3
    ga = torch.full((R, H, D), float("nan"), device=DEV) if fold else None  # (the attention-vector gradient from the same pass)
    k.rgat_backward_compact(grp, f, l, r_, sm, ret, go.to(DEV), gf, gl, gr, slope, fold_attn_l=attn.to(DEV) if fold else None,
                            row_rel_ptrs=ss["rel_ptrs_row"].to(DEV) if fold else None, grad_bias=gb, bias_rows=nb, runs=runs,
                            drow_nodes=ss["node_indices_col"].to(DEV), grad_attn_l=ga)
    assert_close(gf, gf_r, what="grad_feat")
    assert_close(gl, gl_r, what="grad_el")
    assert_close(gr, gr_r, what="grad_er")
    if bias:
        assert_close(gb, to64(go).view(N, -1)[:nb].sum(0), what="grad_bias")
    if fold:  # grad_attn_l[r,h,:] = SUM over the rows u of relation r of grad_el[u,h] * feat[u,h,:]
        ga_r = torch.zeros(R, H, D, dtype=torch.float64).index_add_(0, rel_of_row, gl_r.unsqueeze(-1) * to64(feat))
        assert_close(ga, ga_r, what="grad_attn_l")
    # the cache evicts and rebuilds the grouping by destination while the (destination, relation) one stays: the hub lists that
    # were built against the old object are rebuilt, same results
    import het_amd.plan as plan
    with plan._cache_lock:
        for key in [kk for kk, v in plan._cache.items() if v is grp[0]]:
            del plan._cache[key]
    old_by_dst = grp[0]
    grp = k.rgat_compact_groupings(s["col_indices"].to(DEV), srow_p.to(DEV), drow_p.to(DEV), N, S_row, S_col, rel_ptrs=s["rel_ptrs"].to(DEV),
                                   drow_nodes=ss["node_indices_col"].to(DEV), drow_rel_ptrs=ss["rel_ptrs_col"].to(DEV))
    assert grp[0] is not old_by_dst
    sm3, ret5 = torch.full((N, H), 7.0, device=DEV), torch.full((N, H, D), 7.0, device=DEV)
    k.rgat_aggregate_compact(grp, f, l, r_, sm3, ret5, slope, num_rels=R)
    assert torch.equal(ret5, ret) and torch.equal(sm3, sm)
    del old_by_dst
    # er rows that are not the (relation, destination) pairs of their edges are refused (checked once per list)
    bad = drow_p.clone()
    bad[0] = (bad[0] + 1) % S_col
    with pytest.raises(Exception, match="er row"):
        k.rgat_compact_groupings(s["col_indices"].to(DEV), srow_p.to(DEV), bad.to(DEV), N, S_row, S_col, rel_ptrs=s["rel_ptrs"].to(DEV),
                                 drow_nodes=ss["node_indices_col"].to(DEV), drow_rel_ptrs=ss["rel_ptrs_col"].to(DEV))


def test_rgat_backward_packs_do_not_depend_on_the_groupings_first_user():
    """The RGAT backward walks the (relation, source) segments of its grouping in packs of its own threshold (64), kept beside the
    library-wide packs (32) that e.g. a segment sum over the SAME grouping object uses (csrc/grouping.hip: grouping_pack_view).
    Whoever touches a fresh grouping first, the backward's result is the same bit pattern (the first user's threshold used
    to decide for everybody)."""
    import het_amd.kernels as k
    import het_amd.plan as plan
    H, D, slope = 4, 16, 0.2
    # (relation, source) segments of ~30 .. 75 edges: on both sides of both thresholds -- a segment of 33 .. 64 edges is summed by one
    # lane group under the backward's packs and by a wave under the library-wide ones (another fp32 order) -- and below the 256 of a
    # split segment (no float atomics: the bit patterns are reproducible)
    g = random_graph(seed=33, n=40, r=4, e=6000)
    s = g.get_separate_coo_original()
    ss = g.get_separate_unique_node_indices_single_sided()
    N, R = g.get_num_nodes(), g.get_num_rels()
    rel = torch.repeat_interleave(torch.arange(R), s["rel_ptrs"][1:] - s["rel_ptrs"][:-1])

    def rows_of(ptrs, nodes, ids):
        key = torch.repeat_interleave(torch.arange(R), ptrs[1:] - ptrs[:-1]) * N + nodes
        return torch.searchsorted(key, rel * N + ids).contiguous()
    srow, drow = rows_of(ss["rel_ptrs_row"], ss["node_indices_row"], s["row_indices"]), rows_of(ss["rel_ptrs_col"], ss["node_indices_col"], s["col_indices"])
    S_row, S_col = ss["node_indices_row"].numel(), ss["node_indices_col"].numel()
    gen = torch.Generator().manual_seed(5)
    f, l, r_ = (torch.randn(S_row, H, D, generator=gen).to(DEV), torch.randn(S_row, H, generator=gen).to(DEV), torch.randn(S_col, H, generator=gen).to(DEV))
    go = torch.randn(N, H, D, generator=gen).to(DEV)
    col_d, srow_d, drow_d = s["col_indices"].to(DEV), srow.to(DEV), drow.to(DEV)
    extra = dict(rel_ptrs=s["rel_ptrs"].to(DEV), drow_nodes=ss["node_indices_col"].to(DEV), drow_rel_ptrs=ss["rel_ptrs_col"].to(DEV))

    def backward(segment_sum_first):
        plan.clear()  # fresh grouping objects
        grp = k.rgat_compact_groupings(col_d, srow_d, drow_d, N, S_row, S_col, **extra)
        if segment_sum_first:  # the library-wide packs of the by-feat-row grouping are built before the backward sees it
            out = torch.zeros(S_row, H * D, device=DEV)
            k._call(out, "het_rows_scatter_add_grouped", grp[1].handle, k._p(go), H * D, k._p(out), out.shape[0], k._stream(out))
        sm, ret = torch.empty(N, H, device=DEV), torch.empty(N, H, D, device=DEV)
        runs = k.rgat_aggregate_compact(grp, f, l, r_, sm, ret, slope, num_rels=R)
        gf, gl, gr = torch.full_like(f, float("nan")), torch.full_like(l, float("nan")), torch.full_like(r_, float("nan"))
        k.rgat_backward_compact(grp, f, l, r_, sm, ret, go, gf, gl, gr, slope, runs=runs, drow_nodes=extra["drow_nodes"])
        return gf.cpu(), gl.cpu(), gr.cpu()

    a, b = backward(False), backward(True)
    for name, u, v in zip(("grad_feat", "grad_el", "grad_er"), a, b):
        assert not torch.isnan(u).any(), name
        assert torch.equal(u, v), f"{name} depends on which op built the grouping's packs first"
    plan.clear()


@pytest.mark.parametrize("H,D,n,e", [(8, 8, 300, 5000), (1, 64, 300, 5000), (4, 16, 40, 9000), (2, 8, 12, 9000), (4, 32, 300, 700),
                                     (1, 32, 30, 4000), (2, 32, 300, 3000), (1, 8, 300, 5000), (1, 8, 12, 9000)])
def test_hgt_compact_passes(K, H, D, n, e):
    """het_hgt_aggregate_compact / het_hgt_backward_compact (include/het_amd.h) against oracle/ops.py::hgt_attention_rows in
    fp64 (backward: its autograd): a = softmax over the in-edges of <k'[srow], q[dst]> per head, out = SUM a * m[srow].  The layer-level parity with the oracle is tests/test_gpu_layers.py::test_hgt_layer_fused.
    n = 12 / 30 / 40: hub destinations split over work items and long (relation, source) segments whose pieces add atomically."""
    import het_amd.kernels as k
    g = random_graph(seed=29, n=n, r=4, e=e)
    s = g.get_separate_coo_original()
    inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    ss = g.get_separate_unique_node_indices_single_sided()
    N, S_row, X = g.get_num_nodes(), ss["node_indices_row"].numel(), H * D
    srow = inv["inverse_indices_row"][s["eids"]].contiguous()  # row of every edge POSITION
    col = s["col_indices"]
    gen = torch.Generator().manual_seed(6)
    kv, q, go = torch.randn(S_row, 2, H, D, generator=gen) * 0.6, torch.randn(N, H, D, generator=gen) * 0.6, torch.randn(N, H, D, generator=gen)
    kv64, q64 = to64(kv).requires_grad_(True), to64(q).requires_grad_(True)
    den, out_r = O.hgt_attention_rows(kv64, q64, srow, col, N)
    gkv_r, gq_r = torch.autograd.grad(out_r, [kv64, q64], to64(go))
    grp = k.hgt_compact_groupings(col.to(DEV), srow.to(DEV), N, S_row)
    kvd, qd = kv.to(DEV), q.to(DEV)
    lsum, out = torch.full((N, H), 7.0, device=DEV), torch.full((N, X), 7.0, device=DEV)
    k.hgt_aggregate_compact(grp, kvd, qd, lsum, out)
    # lsum of this entry point is the log-sum-exp of the destination (running-maximum softmax); 0 without in-edges
    has_in = (den.detach() > 0).any(1)
    assert_close(lsum[has_in.to(DEV)], torch.log(den.detach()[has_in]), what="log-sum-exp")
    assert_close(out.view(N, H, D), out_r.detach(), what="out")
    gkv, gq = torch.full_like(kvd, float("nan")), torch.full_like(qd, float("nan"))
    k.hgt_backward_compact(grp, kvd, qd, lsum, out, go.to(DEV), gkv, gq)
    assert_close(gq, gq_r, what="grad_q")
    assert_close(gkv, gkv_r, what="grad_kv")


def test_fused_gat_hub_destination(K, plan_mode):
    """One destination with thousands of in-edges (its segment is split over several work
    items), many destinations with none, and eids a non-trivial permutation."""
    from het_amd.graph import HetGraph
    from het_amd.synth import IntegratedCOO
    gen = torch.Generator().manual_seed(4)
    N, E, R, H, D = 500, 6000, 3, 4, 16
    col = torch.randint(0, 40, (E,), generator=gen)
    col[: E // 2] = 7  # hub
    row = torch.randint(0, N, (E,), generator=gen)
    rel = torch.sort(torch.randint(0, R, (E,), generator=gen)).values
    g = HetGraph.from_integrated_coo(IntegratedCOO(N, R, torch.tensor([0, N]), row, col, rel, torch.randperm(E, generator=gen)))
    s, feat, el, er, go, df, db = _gat_case(g, 0, H, D, seed=2)
    idx = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
    sm_r, ex_r, ret_r = (torch.empty(N, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64),
                         torch.empty(N, H, D, dtype=torch.float64))
    O.relational_fused_gat_separate_coo(*idx, 0, {}, to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r, 0.2)
    didx = tuple(t.to(DEV) for t in idx)
    sm, ex, ret = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(N, H, D, device=DEV)
    K.relational_fused_gat_separate_coo(*didx, 0, {}, feat.to(DEV), el.to(DEV), er.to(DEV), sm, ex, ret, 0.2)
    assert_close(sm, sm_r, what="sum")
    assert_close(ret, ret_r, what="ret")
    assert float(ret[41:].abs().max()) == 0.0  # destinations without in-edges


@pytest.mark.parametrize("compact", [False, True])
def test_fused_gat_csr(K, plan_mode, compact):
    g = random_graph(seed=23, n=200, r=3, e=2500, empty_rel=False)
    i, o, u = g.get_in_csr(), g.get_out_csr(), g.get_separate_unique_node_indices()
    N, E, H, D, slope = g.get_num_nodes(), g.get_num_edges(), 2, 8, 0.2
    n_rows = int(u["rel_ptrs"][-1]) if compact else E
    gen = torch.Generator().manual_seed(1)
    feat, el, er = torch.randn(n_rows, H, D, generator=gen), torch.randn(n_rows, H, generator=gen), torch.randn(n_rows, H, generator=gen)
    go = torch.randn(N, H, D, generator=gen)
    sm_r, ex_r, ret_r = (torch.empty(N, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64), torch.empty(N, H, D, dtype=torch.float64))
    O.relational_fused_gat_csr(i["row_ptrs"], i["col_indices"], i["eids"], i["rel_types"], u["rel_ptrs"], u["node_indices"],
                               to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r, slope, compact)
    gf_r, gl_r, gr_r = torch.zeros_like(to64(feat)), torch.zeros_like(to64(el)), torch.zeros_like(to64(er))
    O.backward_relational_fused_gat_csr(o["row_ptrs"], o["col_indices"], o["eids"], o["rel_types"], u["rel_ptrs"], u["node_indices"],
                                        to64(feat), to64(el), to64(er), sm_r, ex_r, ret_r, to64(go), gf_r, gl_r, gr_r, slope, compact)
    D_ = lambda d, ks: [d[k].to(DEV) for k in ks]
    sm, ex, ret = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(N, H, D, device=DEV)
    f, l, r_ = feat.to(DEV), el.to(DEV), er.to(DEV)
    from het_amd import _lib as HL
    HL.kernel_timing(True)
    K.relational_fused_gat_csr(*D_(i, ["row_ptrs", "col_indices", "eids", "rel_types"]), *D_(u, ["rel_ptrs", "node_indices"]),
                               f, l, r_, sm, ex, ret, slope, compact)
    assert_close(sm, sm_r, what="sum")
    assert_close(ex, ex_r, what="exp")
    assert_close(ret, ret_r, what="ret")
    gf, gl, gr = torch.zeros_like(f), torch.zeros_like(l), torch.zeros_like(r_)
    K.backward_relational_fused_gat_csr(*D_(o, ["row_ptrs", "col_indices", "eids", "rel_types"]), *D_(u, ["rel_ptrs", "node_indices"]),
                                        f, l, r_, sm, ex, ret, go.to(DEV), gf, gl, gr, slope, compact)
    assert_close(gf, gf_r, what="grad_feat")
    assert_close(gl, gl_r, what="grad_el")
    assert_close(gr, gr_r, what="grad_er")
    # with groupings the CSR pair -- compact rows included -- is served by the destination- / row-grouped kernels, not by the
    # float-atomics edge kernels (the library's kernel timers name the grouped launches)
    fwd_grouped = HL.kernel_timing_read("HET_gat_aggregate_grouped")[1]
    bwd_grouped = HL.kernel_timing_read("HET_gat_backward_grouped")[1] + HL.kernel_timing_read("HET_gat_backward_src")[1]
    HL.kernel_timing(False)
    import het_amd.kernels as k
    if k.COMPILED_LIB:
        # the compiled registration object has its own switch (HET_SHIM_GROUPINGS in its environment, not het_amd.plan): both CSR
        # pairs on the grouped kernels whatever plan_mode says
        assert fwd_grouped >= 1 and bwd_grouped >= 1, (fwd_grouped, bwd_grouped)
    elif plan_mode:
        assert fwd_grouped >= 1 and bwd_grouped >= 1, (fwd_grouped, bwd_grouped)
    else:
        assert fwd_grouped == 0 and bwd_grouped == 0


def test_gat_golden_exp_sum(K, golden_mag):
    """exp / sum against the vectors produced by the reference's ref_rgat.py."""
    gold = golden_mag
    n = int(gold["num_nodes"])
    el, er = gold["gat_el"], gold["gat_er"]
    E, H = el.shape
    feat = torch.randn(E, H, 4)
    sm, ex, ret = torch.empty(n, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(n, H, 4, device=DEV)
    K.relational_fused_gat_separate_coo(torch.arange(E, device=DEV), gold["sep_rel_ptrs"].to(DEV), gold["sep_row"].to(DEV),
                                        gold["sep_col"].to(DEV), 0, {}, feat.to(DEV), el.to(DEV), er.to(DEV), sm, ex, ret,
                                        float(gold["gat_slope"]))
    torch.testing.assert_close(cpu(ex), gold["gat_exp"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cpu(sm), gold["gat_sum"], rtol=1e-5, atol=1e-5)


def test_gat_golden_compact_exp_sum(K, plan_mode, golden_mag):
    """CompactAsOfNodeKind 4 forward: exp / sum against the reference's dual-unique-list wrapper (ref_rgat.py:77-115)."""
    gold = golden_mag
    n = int(gold["num_nodes"])
    el, er = gold["gatc_el"], gold["gatc_er"]
    E, H = gold["gatc_exp"].shape
    d = {"edata_idx_to_inverse_idx_row": gold["ss_inverse_indices_row"].to(DEV),
         "edata_idx_to_inverse_idx_col": gold["ss_inverse_indices_col"].to(DEV)}
    feat = torch.randn(el.shape[0], H, 4)
    sm, ex, ret = torch.empty(n, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(n, H, 4, device=DEV)
    K.relational_fused_gat_separate_coo(torch.arange(E, device=DEV), gold["sep_rel_ptrs"].to(DEV), gold["sep_row"].to(DEV),
                                        gold["sep_col"].to(DEV), 4, d, feat.to(DEV), el.to(DEV), er.to(DEV), sm, ex, ret,
                                        float(gold["gat_slope"]))
    torch.testing.assert_close(cpu(ex), gold["gatc_exp"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cpu(sm), gold["gatc_sum"], rtol=1e-5, atol=1e-5)


def test_gat_golden_backward_grad_feat_src(K, plan_mode, golden_mag):
    """a5's grad_feat_src against the reference's own backward (ref_rgat.py:66-75; per-edge rows summed per source node as
    that file indexes them -- tests/test_oracle.py::test_gat_backward_grad_feat_src_golden)."""
    gold = golden_mag
    n = int(gold["num_nodes"])
    go = gold["gatb_gradout"]
    E, H = gold["gat_exp"].shape
    D = go.shape[2]
    gen = torch.Generator().manual_seed(4)
    feat, ret = torch.randn(E, H, D, generator=gen), torch.randn(n, H, D, generator=gen)
    gf, gl, gr = (torch.zeros(E, H, D, device=DEV), torch.zeros(E, H, device=DEV), torch.zeros(E, H, device=DEV))
    K.backward_relational_fused_gat_separate_coo(torch.arange(E, device=DEV), gold["sep_rel_ptrs"].to(DEV), gold["sep_row"].to(DEV),
                                                 gold["sep_col"].to(DEV), 0, {}, feat.to(DEV), gold["gat_el"].to(DEV),
                                                 gold["gat_er"].to(DEV), gold["gat_sum"].to(DEV), gold["gat_exp"].to(DEV), ret.to(DEV),
                                                 go.to(DEV), gf, gl, gr, float(gold["gat_slope"]))
    per_node = torch.zeros(n, H, D, dtype=torch.float64).index_add_(0, gold["sep_row"], cpu(gf).double())
    torch.testing.assert_close(per_node.float(), gold["gatb_grad_feat_src"], rtol=5e-5, atol=5e-6)  # (fp32 sums of hub sources)


def test_gat_golden_round5_pins(K, plan_mode, golden_mag):
    """Kinds 1 / 2 forward and kinds 4 / 1 backward grad_feat against the reference's own outputs (round-5 fixtures;
    tests/util.py::check_round5_gat_pins; the oracle's side is tests/test_oracle.py::test_gat_round5_pins_golden)."""
    from tests.util import check_round5_gat_pins
    check_round5_gat_pins(K, DEV, golden_mag, golden_mag, golden_mag, float(golden_mag["gat_slope"]), rtol_exp=1e-5, atol_exp=1e-6,
                          rtol_sum=1e-5, atol_sum=1e-5)


# ---------------------------------------------------------------- RGCN
@pytest.mark.parametrize("Kd,D", [(16, 16), (64, 64), (7, 3)])
def test_rgcn_layer1(K, plan_mode, Kd, D):
    g = random_graph(seed=31)
    s = g.get_separate_coo_original()
    R, N, E = g.get_num_rels(), g.get_num_nodes(), g.get_num_edges()
    gen = torch.Generator().manual_seed(8)
    x, W, norm = torch.randn(N, Kd, generator=gen), torch.randn(R, Kd, D, generator=gen), torch.rand(E, 1, generator=gen)
    go = torch.randn(N, D, generator=gen)
    a = (s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"])
    ref = torch.zeros(N, D, dtype=torch.float64)
    O.rgcn_layer1_separate_coo(*a, to64(x), to64(W), to64(norm), ref)
    gx_r, gW_r, gn_r = torch.zeros(N, Kd, dtype=torch.float64), torch.zeros(R, Kd, D, dtype=torch.float64), torch.zeros(E, 1, dtype=torch.float64)
    O.backward_rgcn_layer1_separate_coo(*a, to64(x), to64(W).transpose(1, 2).contiguous(), to64(norm), gn_r, gx_r, to64(go), gW_r)
    da = tuple(t.to(DEV) for t in a)
    ret = torch.zeros(N, D, device=DEV)
    K.rgcn_layer1_separate_coo(*da, x.to(DEV), W.to(DEV), norm.to(DEV), ret)
    assert_close(ret, ref, what="ret")
    gx, gW, gn = torch.zeros(N, Kd, device=DEV), torch.zeros(R, Kd, D, device=DEV), torch.zeros(E, 1, device=DEV)
    K.backward_rgcn_layer1_separate_coo(*da, x.to(DEV), W.transpose(1, 2).contiguous().to(DEV), norm.to(DEV), gn, gx, go.to(DEV), gW)
    assert_close(gx, gx_r, what="grad_x")
    assert_close(gW, gW_r, what="grad_W")
    assert float(gn.abs().sum()) == 0.0


@pytest.mark.parametrize("direct", [False, True])
def test_rgcn_compact_aggregation(K, plan_mode, direct):
    g = random_graph(seed=32)
    s = g.get_separate_coo_original()
    ss, ssi = g.get_separate_unique_node_indices_single_sided(), g.get_separate_unique_node_indices_single_sided_inverse_idx()
    U, X, N, E = int(ss["rel_ptrs_row"][-1]), 16, g.get_num_nodes(), g.get_num_edges()
    gen = torch.Generator().manual_seed(6)
    feat, enorm, go = torch.randn(U, X, generator=gen), torch.rand(E, 1, generator=gen), torch.randn(N, X, generator=gen)
    d = {"inverse_indices_row": ssi["inverse_indices_row"]} if direct else {"rel_ptrs_row": ss["rel_ptrs_row"], "node_indices_row": ss["node_indices_row"]}
    a = (s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"])
    ref, gf_r = torch.empty(N, X, dtype=torch.float64), torch.zeros(U, X, dtype=torch.float64)
    O.rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(*a, d, to64(feat), to64(enorm), ref, direct)
    O.backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(*a, d, to64(feat), to64(enorm), ref, to64(go), gf_r, direct)
    da = tuple(t.to(DEV) for t in a)
    ret, gf = torch.full((N, X), 3.0, device=DEV), torch.zeros(U, X, device=DEV)
    K.rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(*da, _dev(d), feat.to(DEV), enorm.to(DEV), ret, direct)
    K.backward_rgcn_node_mean_aggregation_compact_as_of_node_separate_coo(*da, _dev(d), feat.to(DEV), enorm.to(DEV), ret, go.to(DEV), gf, direct)
    assert_close(ret, ref, what="ret")
    assert_close(gf, gf_r, what="grad_feat")


# ---------------------------------------------------------------- error behaviour
def test_cpu_tensors_are_rejected(K):
    from het_amd._lib import HetError
    e = torch.zeros(1, dtype=torch.int64)
    with pytest.raises((HetError, RuntimeError)):
        K.rgcn_layer1_separate_coo(torch.tensor([0, 1]), e, e, e, torch.randn(2, 4), torch.randn(1, 4, 4), torch.rand(1), torch.zeros(2, 4))


# ---------------------------------------------------------------- HGT ops
def _hub_graph(seed, N=500, E=6000, R=3):
    """One destination with half of all edges (its segment is split over several work items), many without any."""
    from het_amd.graph import HetGraph
    from het_amd.synth import IntegratedCOO
    gen = torch.Generator().manual_seed(seed)
    col = torch.randint(0, 40, (E,), generator=gen)
    col[: E // 2] = 7
    row = torch.randint(0, N, (E,), generator=gen)
    rel = torch.sort(torch.randint(0, R, (E,), generator=gen)).values
    return HetGraph.from_integrated_coo(IntegratedCOO(N, R, torch.tensor([0, N]), row, col, rel, torch.randperm(E, generator=gen)))


@pytest.mark.parametrize("H,dk,hub", [(8, 8, False), (8, 8, True), (4, 8, True), (16, 4, False), (2, 16, False), (3, 5, False),
                                      (1, 4, False)])
def test_hgt_edge_softmax_fwd_bwd(K, plan_mode, H, dk, hub):
    g = _hub_graph(9) if hub else random_graph(seed=61, n=260, r=4, e=4000)
    s = g.get_separate_coo_original()
    N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
    gen = torch.Generator().manual_seed(12)
    score, mu = torch.randn(E, H, generator=gen), torch.rand(R, H, generator=gen) + 0.5
    ga = torch.randn(E, H, generator=gen)
    idx = (s["row_indices"], s["col_indices"], s["eids"], s["rel_ptrs"])
    sm_r, m_r, a_r = torch.empty(N, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64), torch.empty(E, H, dtype=torch.float64)
    O.hgt_full_graph_edge_softmax_ops_separate_coo(*idx, to64(score), to64(mu), sm_r, m_r, a_r)
    gs_r, gmu_r, tmp_r = torch.zeros(E, H, dtype=torch.float64), torch.zeros(R, H, dtype=torch.float64), torch.zeros(N, H, dtype=torch.float64)
    O.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(*idx, to64(score), a_r, to64(ga), to64(mu), gs_r, gmu_r, tmp_r)
    didx = tuple(t.to(DEV) for t in idx)
    sm, m, a = torch.full((N, H), 5.0, device=DEV), torch.empty(E, H, device=DEV), torch.empty(E, H, device=DEV)
    K.hgt_full_graph_edge_softmax_ops_separate_coo(*didx, score.to(DEV), mu.to(DEV), sm, m, a)
    assert_close(sm, sm_r, what="sum"); assert_close(m, m_r, what="m"); assert_close(a, a_r, what="a")
    gs, gmu, tmp = torch.empty(E, H, device=DEV), torch.zeros(R, H, device=DEV), torch.empty(N, H, device=DEV)
    K.backward_hgt_full_graph_enorm_to_unnormalized_attn_score_separate_coo(*didx, score.to(DEV), a, ga.to(DEV), mu.to(DEV), gs, gmu, tmp)
    assert_close(gs, gs_r, what="grad_score"); assert_close(gmu, gmu_r, what="grad_mu")


@pytest.mark.parametrize("H,dk", [(8, 8), (4, 16), (2, 6), (1, 64), (2, 32), (1, 32), (1, 128)])
def test_hgt_fused_message_fwd_bwd(K, plan_mode, H, dk):
    g = random_graph(seed=62, n=280, r=4, e=4500)
    s = g.get_separate_coo_original()
    N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
    gen = torch.Generator().manual_seed(13)
    v, W, a = torch.randn(N, H, dk, generator=gen), torch.randn(R, H, dk, dk, generator=gen) * 0.4, torch.rand(E, H, generator=gen)
    go = torch.randn(N, H, dk, generator=gen)
    idx = (s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"])
    nh_r = torch.zeros(N, H, dk, dtype=torch.float64)
    O.hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*idx, to64(v), to64(W), to64(a), nh_r)
    gv_r, gW_r, ga_r = torch.zeros(N, H, dk, dtype=torch.float64), torch.zeros(R, H, dk, dk, dtype=torch.float64), torch.zeros(E, H, dtype=torch.float64)
    O.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*idx, to64(v), to64(W).transpose(2, 3).contiguous(), to64(a), nh_r, gv_r, gW_r, ga_r, to64(go))
    didx = tuple(t.to(DEV) for t in idx)
    nh = torch.zeros(N, H, dk, device=DEV)
    K.hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*didx, v.to(DEV), W.to(DEV), a.to(DEV), nh)
    assert_close(nh, nh_r, what="new_h")
    gv, gW, ga = torch.zeros(N, H, dk, device=DEV), torch.zeros(R, H, dk, dk, device=DEV), torch.full((E, H), float("nan"), device=DEV)
    K.backward_hgt_full_graph_fused_message_calc_and_mean_aggregation_separate_coo(*didx, v.to(DEV), W.transpose(2, 3).contiguous().to(DEV), a.to(DEV), nh, gv, gW, ga, go.to(DEV))
    assert_close(gv, gv_r, what="grad_v"); assert_close(gW, gW_r, what="grad_W"); assert_close(ga, ga_r, what="grad_a")


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("H,dk", [(8, 8), (3, 5)])
def test_inner_product_right_node(K, plan_mode, kind, H, dk):
    g = random_graph(seed=63, n=240, r=3, e=3500, empty_rel=False)
    s = g.get_separate_coo_original()
    ss, ssi = g.get_separate_unique_node_indices_single_sided(), g.get_separate_unique_node_indices_single_sided_inverse_idx()
    N, E = g.get_num_nodes(), g.get_num_edges()
    nl = E if kind == 0 else int(ss["rel_ptrs_col"][-1])
    d = {} if kind == 0 else ({"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"], "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}
                              if kind == 1 else {"edata_idx_to_inverse_idx": ssi["inverse_indices_col"]})
    gen = torch.Generator().manual_seed(14)
    left, right, go = torch.randn(nl, H, dk, generator=gen), torch.randn(N, H, dk, generator=gen), torch.randn(E, H, generator=gen)
    idx = (s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"])
    out_r = torch.zeros(E, H, dtype=torch.float64)
    O.rgnn_inner_product_right_node_separatecoo(d, kind, *idx, to64(left), to64(right), out_r)
    gl_r, gr_r = torch.zeros(nl, H, dk, dtype=torch.float64), torch.zeros(N, H, dk, dtype=torch.float64)
    O.backward_inner_product_right_node_separatecoo(d, kind, *idx, to64(left), to64(right), to64(go), gl_r, gr_r)
    didx = tuple(t.to(DEV) for t in idx)
    out = torch.full((E, H), float("nan"), device=DEV)
    K.rgnn_inner_product_right_node_separatecoo(_dev(d), kind, *didx, left.to(DEV), right.to(DEV), out)
    assert_close(out, out_r, what="out")
    gl, gr = torch.zeros(nl, H, dk, device=DEV), torch.zeros(N, H, dk, device=DEV)
    K.backward_inner_product_right_node_separatecoo(_dev(d), kind, *didx, left.to(DEV), right.to(DEV), go.to(DEV), gl, gr)
    assert_close(gl, gl_r, what="grad_left"); assert_close(gr, gr_r, what="grad_right")


@pytest.mark.parametrize("H,dk", [(8, 8), (2, 6)])
def test_hgt_fused_attention(K, plan_mode, H, dk):
    g = random_graph(seed=64, n=230, r=3, e=3200)
    s = g.get_separate_coo_original()
    N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
    gen = torch.Generator().manual_seed(15)
    k, q, W = torch.randn(N, H, dk, generator=gen), torch.randn(N, H, dk, generator=gen), torch.randn(R, H, dk, dk, generator=gen) * 0.4
    gs = torch.randn(E, H, generator=gen)
    idx = (s["row_indices"], s["col_indices"], s["eids"], s["rel_ptrs"])
    inner_r, score_r = torch.zeros(E, H, dk, dtype=torch.float64), torch.zeros(E, H, dtype=torch.float64)
    O.hgt_full_graph_hetero_attention_ops_coo(*idx, to64(k), to64(q), to64(W), inner_r, score_r)
    gW_r, gk_r, gq_r = torch.zeros(R, H, dk, dk, dtype=torch.float64), torch.zeros(N, H, dk, dtype=torch.float64), torch.zeros(N, H, dk, dtype=torch.float64)
    dummy = torch.zeros(0, dtype=torch.int64)
    O.backward_hgt_full_graph_hetero_attention_ops_coo(dummy, dummy, dummy, dummy, *idx, gW_r, to64(W).transpose(2, 3).contiguous(), to64(k), to64(q), inner_r, to64(gs), gk_r, gq_r)
    didx = tuple(t.to(DEV) for t in idx)
    inner, score = torch.empty(E, H, dk, device=DEV), torch.empty(E, H, device=DEV)
    K.hgt_full_graph_hetero_attention_ops_coo(*didx, k.to(DEV), q.to(DEV), W.to(DEV), inner, score)
    assert_close(inner, inner_r, what="inner"); assert_close(score, score_r, what="score")
    gW, gk, gq = torch.zeros(R, H, dk, dk, device=DEV), torch.zeros(N, H, dk, device=DEV), torch.zeros(N, H, dk, device=DEV)
    dd = dummy.to(DEV)
    K.backward_hgt_full_graph_hetero_attention_ops_coo(dd, dd, dd, dd, *didx, gW, W.transpose(2, 3).contiguous().to(DEV), k.to(DEV), q.to(DEV), inner, gs.to(DEV), gk, gq)
    assert_close(gW, gW_r, what="grad_W"); assert_close(gk, gk_r, what="grad_k"); assert_close(gq, gq_r, what="grad_q")


def test_gat_and_hgt_on_a_graph_without_edges(K):
    """E = 0: outputs are zero-filled, nothing faults."""
    z = torch.zeros(0, dtype=torch.int64, device=DEV)
    rp = torch.zeros(3, dtype=torch.int64, device=DEV)
    N, H, D = 7, 4, 16
    sm, ex, ret = torch.full((N, H), 3.0, device=DEV), torch.zeros(0, H, device=DEV), torch.full((N, H, D), 3.0, device=DEV)
    K.relational_fused_gat_separate_coo(z, rp, z, z, 0, {}, torch.zeros(0, H, D, device=DEV), torch.zeros(0, H, device=DEV),
                                        torch.zeros(0, H, device=DEV), sm, ex, ret, 0.2)
    assert float(sm.abs().sum()) == 0.0 and float(ret.abs().sum()) == 0.0
    s2 = torch.full((N, H), 3.0, device=DEV)
    K.hgt_full_graph_edge_softmax_ops_separate_coo(z, z, z, rp, torch.zeros(0, H, device=DEV), torch.ones(2, H, device=DEV), s2,
                                                   torch.zeros(0, H, device=DEV), torch.zeros(0, H, device=DEV))
    assert float(s2.abs().sum()) == 0.0
    torch.cuda.synchronize()


@pytest.mark.parametrize("H,D,nrel", [(4, 16, 4), (1, 64, 4), (2, 8, 4), (4, 16, 11)])
def test_fused_gat_with_folded_attn_l(H, D, nrel):
    """el = <feat, attn_l[r]> + GAT under one autograd node (fold_attn_l of include/het_amd.h) against the unfused
    composition of the two reference-named functions, on a graph whose relations are interleaved (eids != arange)."""
    import het_amd.backend as B
    g = random_graph(seed=77, n=300, r=nrel, e=5000, empty_rel=False)  # > 8 relations: separate weight-gradient pass
    s = g.get_separate_coo_original()
    assert not torch.equal(s["eids"], torch.arange(s["eids"].numel()))
    E, R = g.get_num_edges(), g.get_num_rels()
    gen = torch.Generator().manual_seed(5)
    feat = (0.5 * torch.randn(E, H, D, generator=gen)).to(DEV)
    attn = (0.5 * torch.randn(R, H, D, generator=gen)).to(DEV)
    er = (0.5 * torch.randn(E, H, generator=gen)).to(DEV)
    go = torch.randn(g.get_num_nodes(), H, D, generator=gen).to(DEV)
    g.to_(DEV)
    sd = g.get_separate_coo_original()
    by_eid = {"separate_coo_rel_ptrs": sd["rel_ptrs"], "separate_coo_node_indices": sd["eids"],
              "separate_coo_eids": sd["eids"]}
    outs = []
    for fused in (False, True):
        f, a, r = (t.clone().requires_grad_(True) for t in (feat, attn, er))
        if fused:
            assert B.relational_fused_gat_separate_coo_with_attn_l_ok(g, f, a, 0.2)
            out = B.relational_fused_gat_separate_coo_with_attn_l(g, f, a, r, 0.2)
        else:
            el = B.rgnn_relational_matmul(by_eid, a.unsqueeze(-1), f, False, 0).view(E, H)
            out = B.relational_fused_gat_separate_coo(g, f, el, r, 0.2)
        out.backward(go)
        outs.append((out.detach(), f.grad, a.grad, r.grad))
    g.cpu_()
    for name, u, v in zip(("out", "grad_feat", "grad_attn_l", "grad_er"), outs[0], outs[1]):
        assert_close(v, u.cpu(), what=name)


@pytest.mark.parametrize("H,Kd,D", [(4, 64, 16), (1, 32, 32), (2, 64, 64), (8, 128, 4)])
def test_matmul_with_attn_dot_epilogue(H, Kd, D):
    """het_rgnn_relational_matmul_attn_dot (projection + attention term from the GEMM epilogue) against the oracle's
    two ops, and the autograd node against the unfused composition (feat used, feat unused)."""
    import het_amd.backend as B
    g = random_graph(seed=91, n=300, r=4, e=5000)
    s = g.get_separate_coo_original()
    E, R, N = g.get_num_edges(), g.get_num_rels(), g.get_num_nodes()
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(N, Kd, generator=gen)
    W = 0.2 * torch.randn(R, H, Kd, D, generator=gen)
    attn = torch.randn(R, H, D, generator=gen)
    by_src = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"],
              "separate_coo_eids": s["eids"]}
    by_eid = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["eids"], "separate_coo_eids": s["eids"]}
    feat_ref = torch.zeros(E, H, D, dtype=torch.float64)
    O.rgnn_relational_matmul(by_src, 0, to64(W), to64(x), feat_ref, True)
    dot_ref = torch.zeros(E, H, 1, dtype=torch.float64)
    O.rgnn_relational_matmul(by_eid, 0, to64(attn).unsqueeze(-1), feat_ref, dot_ref, False)
    assert B.rgnn_relational_matmul_with_attn_dot_ok(W, x.to(DEV))
    go_f = torch.randn(E, H, D, generator=gen).to(DEV)
    go_d = torch.randn(E, H, generator=gen).to(DEV)
    res = {}
    for mode in ("fused", "plain"):
        xd, Wd, ad = (t.to(DEV).requires_grad_(True) for t in (x, W, attn))
        if mode == "fused":
            feat, dot = B.rgnn_relational_matmul_with_attn_dot(_dev(by_src), Wd, xd, ad)
        else:
            feat = B.rgnn_relational_matmul(_dev(by_src), Wd, xd, True, 0)
            dot = B.rgnn_relational_matmul(_dev(by_eid), ad.unsqueeze(-1), feat, False, 0).view(E, H)
        ((feat * go_f).sum() + (dot * go_d).sum()).backward()
        res[mode] = (feat.detach(), dot.detach(), xd.grad, Wd.grad, ad.grad)
        xd2, Wd2, ad2 = (t.to(DEV).requires_grad_(True) for t in (x, W, attn))
        if mode == "fused":  # projection output unused: only the attention term carries gradient
            _, dot2 = B.rgnn_relational_matmul_with_attn_dot(_dev(by_src), Wd2, xd2, ad2)
        else:
            f2 = B.rgnn_relational_matmul(_dev(by_src), Wd2, xd2, True, 0)
            dot2 = B.rgnn_relational_matmul(_dev(by_eid), ad2.unsqueeze(-1), f2, False, 0).view(E, H)
        (dot2 * go_d).sum().backward()
        res[mode + "_dot_only"] = (xd2.grad, Wd2.grad, ad2.grad)
    assert_close(res["fused"][0], feat_ref, what="feat")
    assert_close(res["fused"][1], dot_ref.view(E, H), what="dot")
    for name, u, v in zip(("grad_x", "grad_W", "grad_attn"), res["plain"][2:], res["fused"][2:]):
        assert_close(v, u.cpu(), what=name)
    for name, u, v in zip(("grad_x", "grad_W", "grad_attn"), res["plain_dot_only"], res["fused_dot_only"]):
        assert_close(v, u.cpu(), what=name + " (dot only)")


@pytest.mark.parametrize("H,Kd,D", [(4, 64, 16), (8, 128, 4), (4, 32, 8), (1, 64, 64), (2, 64, 32)])
def test_attn_dot_only_without_the_per_edge_projection(H, Kd, D):
    """er[e,h] = <x[dst_e] . W[r,h], attn[r,h,:]> formed on the distinct (relation, node) rows and duplicated as [E,H]
    only, against the two reference-named ops; values and all gradients."""
    import het_amd.backend as B
    g = random_graph(seed=93, n=260, r=4, e=4500)
    s = g.get_separate_coo_original()
    E, R, N = g.get_num_edges(), g.get_num_rels(), g.get_num_nodes()
    gen = torch.Generator().manual_seed(8)
    x, W, attn = torch.randn(N, Kd, generator=gen), 0.2 * torch.randn(R, H, Kd, D, generator=gen), torch.randn(R, H, D, generator=gen)
    go = torch.randn(E, H, generator=gen).to(DEV)
    by_dst = _dev({"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"],
                   "separate_coo_eids": s["eids"]})
    by_eid = _dev({"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["eids"], "separate_coo_eids": s["eids"]})
    res = []
    for fused in (False, True):
        xd, Wd, ad = (t.to(DEV).requires_grad_(True) for t in (x, W, attn))
        if fused:
            assert B.rgnn_relational_matmul_attn_dot_only_ok(by_dst, Wd, xd)
            er = B.rgnn_relational_matmul_attn_dot_only(by_dst, Wd, xd, ad)
        else:
            f = B.rgnn_relational_matmul(by_dst, Wd, xd, True, 0)
            er = B.rgnn_relational_matmul(by_eid, ad.unsqueeze(-1), f, False, 0).view(E, H)
        (er * go).sum().backward()
        res.append((er.detach(), xd.grad, Wd.grad, ad.grad))
    for name, u, v in zip(("er", "grad_x", "grad_W", "grad_attn"), res[0], res[1]):
        assert_close(v, u.cpu(), what=name)


def test_rows_add_bias():
    import het_amd.kernels as k
    gen = torch.Generator().manual_seed(3)
    a, b, bias = torch.randn(333, 64, generator=gen), torch.randn(333, 64, generator=gen), torch.randn(64, generator=gen)
    for bb, cc in ((b, bias), (None, bias), (b, None)):
        out = k.rows_add_bias(a.to(DEV), None if bb is None else bb.to(DEV), None if cc is None else cc.to(DEV))
        ref = a + (0 if bb is None else bb) + (0 if cc is None else cc)
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("H,Kd,D,n,nrel,gather", [(4, 64, 16, 5003, 1, False), (4, 64, 16, 2777, 3, True), (1, 32, 32, 901, 2, False),
                                                  (2, 64, 64, 1500, 1, False), (4, 128, 32, 1203, 2, True), (1, 32, 64, 7, 1, False),
                                                  (4, 64, 16, 0, 1, False)])
def test_rows_matmul_backward_dw_with_column_sums(H, Kd, D, n, nrel, gather):
    """het_rows_matmul_backward_dw_colsum: the weight gradient as without it, and colsum = the sum of the launch's gradout rows
    (the bias gradient the RGAT layer takes from its self-loop's launch) -- against fp64 sums of the same rows."""
    import het_amd.kernels as k
    gen = torch.Generator().manual_seed(11 + n)
    X, Nx = H * D, max(n, 1) + 17
    cuts = torch.sort(torch.randint(0, n + 1, (nrel - 1,), generator=gen)).values
    rp = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([n])]).to(torch.int64)
    x, go = torch.randn(Nx, Kd, generator=gen), torch.randn(n, X, generator=gen)
    idx = torch.randint(0, Nx, (n,), generator=gen) if gather else None
    gw = torch.full((nrel, H, Kd, D), 7.0)
    gw2 = gw.clone()
    cs = torch.full((X,), 7.0)
    xd, god, rpd, idxd = x.to(DEV), go.to(DEV), rp.to(DEV), None if idx is None else idx.to(DEV)
    gwd, gw2d, csd = gw.to(DEV), gw2.to(DEV), cs.to(DEV)
    k.rows_matmul_backward_dw(rpd, idxd, xd if gather else xd[:n], god, gwd, accumulate=False, colsum=csd)
    k.rows_matmul_backward_dw(rpd, idxd, xd if gather else xd[:n], god, gw2d, accumulate=False)
    assert_close(csd, go.double().sum(0), what="colsum")
    xs = (x[idx] if gather else x[:n]).double()
    for r in range(nrel):
        a, b = int(rp[r]), int(rp[r + 1])
        ref = (xs[a:b].t() @ go[a:b].double()).view(Kd, H, D).permute(1, 0, 2)
        assert_close(gwd[r], ref, what=f"grad_w[{r}]")
        assert_close(gw2d[r], ref, what=f"grad_w[{r}] without the sums")


@pytest.mark.parametrize("H,D,nrel", [(4, 16, 4), (8, 8, 3), (4, 16, 11), (1, 64, 4), (2, 32, 4)])
def test_gat_rank_order_extensions(H, D, nrel):
    """The kind-0 GAT ops with their attention terms in the destination-grouped ("rank") order of the kernels
    (include/het_amd.h: el_sorted / er_sorted of a4, grad_el_sorted of a5, het_grouping_rank_of_position) against the
    oracle on edge-order tensors, on a graph with interleaved relations, a hub destination and eids != arange."""
    import het_amd.kernels as k
    g = random_graph(seed=79, n=240, r=nrel, e=5000, empty_rel=False)
    s = g.get_separate_coo_original()
    E, N = g.get_num_edges(), g.get_num_nodes()
    gen = torch.Generator().manual_seed(9)
    feat, el, er = (0.5 * torch.randn(E, H, D, generator=gen), 0.5 * torch.randn(E, H, generator=gen),
                    0.5 * torch.randn(E, H, generator=gen))
    go = torch.randn(N, H, D, generator=gen)
    f64, l64, r64, go64 = to64(feat), to64(el), to64(er), to64(go)
    sm, ex, ret = torch.zeros(N, H, dtype=torch.float64), torch.zeros(E, H, dtype=torch.float64), torch.zeros(N, H, D, dtype=torch.float64)
    O.relational_fused_gat_separate_coo(s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], 0, {}, f64, l64, r64, sm, ex, ret, 0.2)
    gf, gl, gr = torch.zeros_like(f64), torch.zeros_like(l64), torch.zeros_like(r64)
    O.backward_relational_fused_gat_separate_coo(s["eids"], s["rel_ptrs"], s["row_indices"], s["col_indices"], 0, {}, f64, l64, r64,
                                                 sm, ex, ret, go64, gf, gl, gr, 0.2)
    d = _dev(s)
    eids, rp, row, col = d["eids"], d["rel_ptrs"], d["row_indices"], d["col_indices"]
    rank = k.gat_rank_of_position(rp, row, col, eids, N)
    # rank is a permutation of the positions that sorts them by destination, stably
    assert torch.equal(torch.sort(rank).values.cpu(), torch.arange(E))
    inv = torch.empty_like(rank)
    inv[rank] = torch.arange(E, device=DEV)
    assert torch.equal(col[inv].cpu(), torch.sort(s["col_indices"], stable=True).values)
    # tensors in rank order: row rank[i] holds the value of the edge at position i (edge id eids[i])
    el_s, er_s = torch.empty(E, H, device=DEV), torch.empty(E, H, device=DEV)
    el_s[rank], er_s[rank] = el.to(DEV)[eids], er.to(DEV)[eids]
    new = lambda *shape: torch.empty(shape, device=DEV)
    smd, retd, exs = new(N, H), new(N, H, D), new(E, H)
    used = k.fused_gat_forward(eids, rp, row, col, 0, {}, feat.to(DEV), None, None, smd, None, retd, 0.2, exs, el_sorted=el_s, er_sorted=er_s)
    assert used
    assert_close(smd, sm, what="sum")
    assert_close(retd, ret, what="ret")
    exs_ref = torch.empty(E, H, dtype=torch.float64)
    exs_ref[rank.cpu()] = ex[s["eids"]]
    assert_close(exs, exs_ref, what="exp_sorted")
    g_feat, g_el_s = new(E, H, D), new(E, H)
    k.fused_gat_backward(eids, rp, row, col, 0, {}, feat.to(DEV), None, None, smd, None, retd, go.to(DEV), g_feat, None, None, 0.2, exs,
                         grad_el_sorted=g_el_s)
    assert_close(g_feat, gf, what="grad_feat")
    gl_s_ref = torch.empty(E, H, dtype=torch.float64)
    gl_s_ref[rank.cpu()] = gl[s["eids"]]
    assert_close(g_el_s, gl_s_ref, what="grad_el_sorted")
    assert_close(gl, gr, what="oracle grad_el == grad_er (kind 0)")


@pytest.mark.parametrize("H,Kd,D", [(4, 64, 16), (8, 64, 8)])
def test_attn_dot_with_dot_rows(H, Kd, D):
    """het_rgnn_relational_matmul_attn_dot with dot_grouping: the projected rows go to their edge-id rows, the attention
    terms to caller-chosen rows (a permutation here); and the one-head row-dot forward on distinct rows (a1 with grouping)."""
    import het_amd.kernels as k
    g = random_graph(seed=95, n=260, r=4, e=4500)
    s = g.get_separate_coo_original()
    E, R, N = g.get_num_edges(), g.get_num_rels(), g.get_num_nodes()
    gen = torch.Generator().manual_seed(10)
    x, W, attn = torch.randn(N, Kd, generator=gen), 0.2 * torch.randn(R, H, Kd, D, generator=gen), torch.randn(R, H, D, generator=gen)
    by_src = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"], "separate_coo_eids": s["eids"]}
    by_eid = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["eids"], "separate_coo_eids": s["eids"]}
    feat_ref = torch.zeros(E, H, D, dtype=torch.float64)
    O.rgnn_relational_matmul(by_src, 0, to64(W), to64(x), feat_ref, True)
    dot_ref = torch.zeros(E, H, 1, dtype=torch.float64)
    O.rgnn_relational_matmul(by_eid, 0, to64(attn).unsqueeze(-1), feat_ref, dot_ref, False)
    perm = torch.randperm(E, generator=gen)
    feat, dots = torch.empty(E, H, D, device=DEV), torch.empty(E, H, device=DEV)
    k.matmul_attn_dot(_dev(by_src), 0, W.to(DEV), x.to(DEV), feat, attn.to(DEV), dots, dot_rows=perm.to(DEV))
    assert_close(feat, feat_ref, what="feat")
    ref = torch.empty(E, H, dtype=torch.float64)
    ref[perm] = dot_ref.view(E, H)[s["eids"]]
    assert_close(dots, ref, what="dots at dot_rows")
    # a1, one shared input head, D_out = 1: same values with and without the (relation, node) grouping
    wa = torch.randn(R, H, Kd, 1, generator=gen)
    out_ref = torch.zeros(E, H, 1, dtype=torch.float64)
    O.rgnn_relational_matmul(by_src, 0, to64(wa), to64(x), out_ref, True)
    out = torch.empty(E, H, 1, device=DEV)
    k.K.rgnn_relational_matmul(_dev(by_src), 0, wa.to(DEV), x.to(DEV), out, True)
    assert_close(out, out_ref, what="row-dot on distinct rows")


def test_halo_pack_unpack_rows():
    """het_rows_gather / het_rows_scatter_add (multi-GPU halo pack / unpack) against index_select / index_add_."""
    import het_amd.kernels as k
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(500, 64, generator=gen)
    idx = torch.randint(0, 500, (3000,), generator=gen)  # repeated rows
    out = k.rows_gather(x.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), x.index_select(0, idx))
    src = torch.randn(3000, 64, generator=gen)
    acc = torch.randn(500, 64, generator=gen)
    ref = acc.double().index_add_(0, idx, src.double())
    got = k.rows_scatter_add_(acc.to(DEV), idx.to(DEV), src.to(DEV))  # (grouped by destination row: no atomics)
    assert_close(got, ref, what="scatter_add")
    import het_amd.plan as plan
    with plan.forced(False):  # the float-atomics kernel
        got = k.rows_scatter_add_(acc.to(DEV), idx.to(DEV), src.to(DEV))
    assert_close(got, ref, what="scatter_add (atomics)")
    # a hub row (its contributions span several work items) and rows of other widths
    idx2 = torch.cat([torch.full((2000,), 7), torch.randint(0, 500, (1000,), generator=gen)])
    for X in (4, 32, 128):
        s2, a2 = torch.randn(3000, X, generator=gen), torch.randn(500, X, generator=gen)
        got = k.rows_scatter_add_(a2.to(DEV), idx2.to(DEV), s2.to(DEV))
        assert_close(got, a2.double().index_add_(0, idx2, s2.double()), what=f"scatter_add X={X}")
    assert k.rows_gather(x.to(DEV), idx[:0].to(DEV)).shape == (0, 64)


# ---------------------------------------------------------------- node-major backward GEMMs (layer-level extension)
@pytest.mark.parametrize("H,Kd,D,R", [(4, 64, 16, 4), (1, 64, 64, 3), (2, 32, 16, 5), (4, 64, 8, 1), (8, 32, 8, 4), (2, 64, 32, 6),
                                      (8, 64, 8, 2), (2, 64, 32, 4), (1, 32, 32, 2)])
@pytest.mark.parametrize("with_loop,with_er,typed", [(True, True, False), (False, True, True), (True, False, True), (True, True, True)])
def test_rgat_node_backward_dx(K, H, Kd, D, R, with_loop, with_er, typed):
    """het_rgat_node_backward_dx (csrc/node_gemm.hip) against the per-term definition in fp64 (the terms of a2 / a3:
    RGNNOps.inc.h:946-1010, 660-753): whole range and split ranges, nodes without rows, a self-loop prefix n_loop < N."""
    import het_amd.kernels as k
    if not k.rgat_node_gemm_ok(R, H, Kd, D):
        pytest.skip("shape outside the node-major kernels")
    gen = torch.Generator().manual_seed(17 + H + R)
    N, X = 333, H * D
    n_loop = 301

    def unique_lists(p):
        ptr, nodes = [0], []
        for r in range(R):
            if typed:  # a relation's nodes live in one third of the id range
                lo = (r % 3) * (N // 3)
                cand = torch.arange(lo, lo + N // 3)
            else:
                cand = torch.arange(N)
            keep = cand[torch.rand(cand.numel(), generator=gen) < p]
            nodes.append(keep)
            ptr.append(ptr[-1] + keep.numel())
        return torch.tensor(ptr), torch.cat(nodes)

    rp_row, n_row = unique_lists(0.6)
    rp_col, n_col = unique_lists(0.4)
    S_row, S_col = n_row.numel(), n_col.numel()
    gh = torch.randn(n_loop, X, generator=gen, dtype=torch.float64)
    g_rows = torch.randn(S_row, X, generator=gen, dtype=torch.float64)
    g_er = torch.randn(S_col, H, generator=gen, dtype=torch.float64)
    loop_w = torch.randn(Kd, X, generator=gen, dtype=torch.float64)
    W = torch.randn(R, H, Kd, D, generator=gen, dtype=torch.float64)
    wa = torch.randn(R, H, Kd, generator=gen, dtype=torch.float64)
    # reference, term by term
    gx = torch.zeros(N, Kd, dtype=torch.float64)
    if with_loop:
        gx[:n_loop] += gh @ loop_w.t()
    for r in range(R):
        rows = slice(int(rp_row[r]), int(rp_row[r + 1]))
        nodes = n_row[rows]
        Wr = W[r].permute(1, 0, 2).reshape(Kd, X)  # [K, (h, d)]
        gx.index_add_(0, nodes, g_rows[rows] @ Wr.t())
        if with_er:
            rows = slice(int(rp_col[r]), int(rp_col[r + 1]))
            nodes = n_col[rows]
            gx.index_add_(0, nodes, g_er[rows] @ wa[r])
    f = lambda t: t.float().to(DEV).contiguous()
    row_map = k.node_row_map(rp_row.to(DEV), n_row.to(DEV), N)
    dst_map = k.node_row_map(rp_col.to(DEV), n_col.to(DEV), N)
    ref_map = torch.full((R, N), -1, dtype=torch.int32)
    for r in range(R):
        ref_map[r, n_row[int(rp_row[r]):int(rp_row[r + 1])]] = torch.arange(int(rp_row[r]), int(rp_row[r + 1]), dtype=torch.int32)
    assert torch.equal(row_map.cpu(), ref_map)
    Wt = f(W.transpose(2, 3))
    args = (f(gh) if with_loop else None, f(loop_w.t()) if with_loop else None, f(g_rows), Wt, row_map,
            f(g_er) if with_er else None, f(wa) if with_er else None, dst_map if with_er else None)
    for ranges in ([(0, N)], [(300, N), (0, 300)], [(0, 31), (31, 64), (64, N)]):
        out = torch.full((N, Kd), float("nan"), device=DEV)
        for b, e in ranges:
            k.rgat_node_backward_dx(b, e, n_loop, *args, out)
        assert_close(out, gx, what=f"grad_x {ranges}")
    # node_order: the nodes of a call are entries [b, e) of a list -- sorted by relation presence (homogeneous tiles), whole
    # range and the two ranges of a partition (split = 300), and an arbitrary permutation
    order = k.node_order_by_presence(row_map, dst_map if with_er else None)
    assert sorted(order.cpu().tolist()) == list(range(N))
    out = torch.full((N, Kd), float("nan"), device=DEV)
    k.rgat_node_backward_dx(0, N, n_loop, *args, out, node_order=order)
    assert_close(out, gx, what="grad_x (nodes by relation presence)")
    order2 = k.node_order_by_presence(row_map, dst_map if with_er else None, split=300)
    assert int(order2[:300].max()) < 300 <= int(order2[300:].min())
    out = torch.full((N, Kd), float("nan"), device=DEV)
    for b, e in ((300, N), (0, 300)):
        k.rgat_node_backward_dx(b, e, n_loop, *args, out, node_order=order2)
    assert_close(out, gx, what="grad_x (nodes by relation presence, two ranges)")
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(torch.int32).to(DEV)
    out = torch.full((N, Kd), float("nan"), device=DEV)
    k.rgat_node_backward_dx(0, N, n_loop, *args, out, node_order=perm)
    assert_close(out, gx, what="grad_x (random node order)")



@pytest.mark.parametrize("R,n,e", [(4, 300, 5000), (1, 50, 400), (6, 2000, 3000)])
def test_grouping_segment_map_and_payload_gather(R, n, e):
    """het_grouping_segment_map (segment of every (relation, key) pair, -1 where there is none) and het_grouping_gather_payload1
    (per-edge-id values in the grouping's order) against their definitions; het_node_rows_matmul_sum_bias with a bias row and
    sources of 32 / 64 floats against the per-source products in fp64 (nodes without any row get the bias)."""
    import het_amd.kernels as k
    g = random_graph(seed=71, n=n, r=R, e=e, shuffle=True, empty_rel=R > 2)
    s = {kk: v.to(DEV) for kk, v in g.get_separate_coo_original().items()}
    N, E = g.get_num_nodes(), g.get_num_edges()
    grp = k._plan.get_grouping(s["rel_ptrs"], s["col_indices"], N, s["row_indices"], s["eids"])
    m = k._grouping_segment_map(grp, s["rel_ptrs"], s["col_indices"], N).cpu()
    rel = torch.repeat_interleave(torch.arange(R), (s["rel_ptrs"][1:] - s["rel_ptrs"][:-1]).cpu())
    pairs = torch.unique(rel * N + s["col_indices"].cpu())  # sorted: segment s is the s-th distinct (relation, key) pair
    ref = torch.full((R, N), -1, dtype=torch.int32)
    ref.view(-1)[pairs] = torch.arange(pairs.numel(), dtype=torch.int32)
    assert grp.num_segments == pairs.numel() and torch.equal(m, ref)
    # values by edge id -> grouping order: rank j holds the value of the edge at sorted rank j
    vals = torch.rand(E, 1, device=DEV)
    out = torch.empty_like(vals)
    k._call(vals, "het_grouping_gather_payload1", grp.handle, k._p(vals), 1, k._p(out), k._stream(vals))
    rank = torch.empty(E, dtype=torch.int64, device=DEV)
    k._call(rank, "het_grouping_rank_of_position", grp.handle, k._p(rank), k._stream(rank))
    want = torch.empty_like(vals)
    want[rank] = vals[s["eids"]]  # position p sits at rank[p]; its edge id is eids[p]
    assert torch.equal(out, want)
    # node-major sum with a bias row
    for KS, XO in ((64, 64), (32, 64), (64, 32), (32, 32)):
        S = min(R, 3)
        gen = torch.Generator().manual_seed(5)
        rows = [torch.randn(pairs.numel() + 1, KS, generator=gen) for _ in range(S)]
        wts = [torch.randn(KS, XO, generator=gen) * 0.2 for _ in range(S)]
        bias = torch.randn(XO, generator=gen)
        maps = [ref[r_].clone() for r_ in range(S)]
        want = bias.double().expand(N, XO).clone()
        for r_ in range(S):
            has = maps[r_] >= 0
            want[has] += rows[r_].double()[maps[r_][has].long()] @ wts[r_].double()
        outn = torch.full((N, XO), float("nan"), device=DEV)
        order = torch.randperm(N, generator=gen).to(torch.int32).to(DEV)
        srcs = [(rows[r_].to(DEV), 0, maps[r_].to(DEV), wts[r_].to(DEV)) for r_ in range(S)]
        ptrs = (k.C.c_void_p * S)(*[t[0].data_ptr() for t in srcs])
        strides = (k.C.c_int64 * S)(*[KS] * S)
        mps = (k.C.c_void_p * S)(*[t[2].data_ptr() for t in srcs])
        ident = (k.C.c_int64 * S)(*[0] * S)
        w_ = (k.C.c_void_p * S)(*[t[3].data_ptr() for t in srcs])
        b_ = bias.to(DEV)
        k._call(outn, "het_node_rows_matmul_sum_bias", 0, N, N, S, ptrs, strides, mps, ident, w_, k._p(b_), k._p(outn), KS, XO, k._p(order),
                k._stream(outn))
        assert_close(outn, want, what=f"node sum + bias {KS}->{XO}")


def test_rgat_row_kernels_with_64_bit_offsets():
    """The RGAT row kernels take their byte offsets as a template argument: 32-bit off scalar bases when every indexed table is below
    4 GiB (what every other test runs), 64-bit otherwise.  HET_RGAT_WIDE_OFFSETS=1 forces the 64-bit instantiation; it is read once per
    process, so the run-sum cases run again in a child interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HET_RGAT_WIDE_OFFSETS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-x", "-m", "gpu", "-k",
                        "test_rgat_compact_run_sums and True-True", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900,
                       env=env, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
