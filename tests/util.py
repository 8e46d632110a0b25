"""Shared helpers of the GPU parity tests."""
import torch

from het_amd.graph import HetGraph
from het_amd.synth import make_mag_like, make_random


def cpu(t):
    return t.detach().cpu()


def to64(t):
    return t.detach().cpu().double()


def random_graph(seed=0, n=257, r=5, e=3001, empty_rel=True, shuffle=True):
    coo = make_random(n, r, e, seed=seed)
    if empty_rel and r > 2:  # leave one relation empty, one tiny
        coo.rel[coo.rel == 1] = 0
        coo.rel = torch.sort(coo.rel).values
    g = HetGraph.from_integrated_coo(coo)
    if shuffle:
        permute_eids(g, seed + 1000)
    return g


def permute_eids(g, seed):
    """HetGraph canonicalises eids to arange(E) in separate-COO order, as the reference does; the ops accept any
    edge numbering, so renumber every layout's eids with one random permutation (edge data row != position)."""
    E = g.get_num_edges()
    pi = torch.randperm(E, generator=torch.Generator().manual_seed(seed))

    def walk(d):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v)
            elif k == "eids":
                d[k] = pi[v]
            elif k in ("inverse_indices_row", "inverse_indices_col") and v.numel() == E:
                # the single-sided inverse indices are read by EDGE ID (inverse_indices_row[eid] = compact row of that edge's
                # source: SURVEY.md section 9, layouts table): the entry of position p moves to the position's new id
                moved = torch.empty_like(v)
                moved[pi] = v
                d[k] = moved

    walk(g.graph_data)
    g._plans.clear()


def mag_graph(scale=2e-3):
    return HetGraph.from_integrated_coo(make_mag_like(scale=scale))


def assert_close(actual, expected, rtol=2e-4, atol=2e-5, what=""):
    """fp32 HIP result vs fp64 oracle.  Tolerance: the reference states rtol 1e-3 between
    implementations (hrt/python/utils_lite/graphiler_bench.py:22-26); we hold 2e-4, with atol
    scaled to the magnitude of the expected tensor (sums over up to thousands of edges)."""
    expected = expected.detach().to(torch.float64)
    scale = float(expected.abs().max()) if expected.numel() else 1.0
    torch.testing.assert_close(actual.detach().cpu().double(), expected, rtol=rtol, atol=atol * max(1.0, scale),
                               msg=lambda m: f"{what}: {m}")


def rgat_min_abs_preactivation(x, W, attn_l, attn_r, sep):
    """min |el + er| over the (edge, head) pairs of an RGAT layer, in fp64.  A pre-activation within fp32 rounding of 0 sits
    on the leaky-ReLU kink: the fp32 kernels (the reference's included) and the fp64 oracle can then take different
    branches, which changes a gradient by a finite amount -- inherent to fp32, not an error of either side.  The layer
    tests nudge their random input until no edge sits there."""
    R = W.shape[0]
    rel = torch.repeat_interleave(torch.arange(R), sep["rel_ptrs"][1:] - sep["rel_ptrs"][:-1])
    x, W = x.double(), W.detach().double()
    wl = torch.einsum("rhkd,rhd->rhk", W, attn_l.detach().double())  # el = x[src] . (W . attn_l)
    wr = torch.einsum("rhkd,rhd->rhk", W, attn_r.detach().double())
    z = torch.einsum("ek,ehk->eh", x[sep["row_indices"]], wl[rel]) + torch.einsum("ek,ehk->eh", x[sep["col_indices"]], wr[rel])
    return float(z.abs().min()) if z.numel() else 1.0


def rgat_nudge_off_kink(x, W, attn_l, attn_r, sep, margin=2e-6, max_rounds=40, step=1e-3, seed=0):
    """x with the rows of a few nodes moved by ~`step` so that no (edge, head) pre-activation el + er (fp64) lies within
    `margin` of the leaky-ReLU kink; returns (x, min |el + er|).  Works on any device and at full size: with 84 M
    pre-activations a few hundred land within 2e-6 of zero whatever the seed, so re-drawing the whole input cannot clear
    them -- instead only ONE endpoint of every offending edge is moved (the one that touches fewer edges), which re-draws
    the pre-activations of that node's edges only; a couple of rounds leave none."""
    R = W.shape[0]
    dev = x.device
    rp, row, col = sep["rel_ptrs"].to(dev), sep["row_indices"].to(dev), sep["col_indices"].to(dev)
    rel = torch.repeat_interleave(torch.arange(R, device=dev), rp[1:] - rp[:-1])
    W64 = W.detach().double()
    wl = torch.einsum("rhkd,rhd->rhk", W64, attn_l.detach().double())  # el = x[src] . (W . attn_l)
    wr = torch.einsum("rhkd,rhd->rhk", W64, attn_r.detach().double())
    N = x.shape[0]
    deg = torch.bincount(row, minlength=N) + torch.bincount(col, minlength=N)
    gen = torch.Generator(device=dev).manual_seed(seed)
    x = x.clone()
    zmin = 1.0
    for _ in range(max_rounds):
        x64 = x.double()
        el_n = torch.einsum("nk,rhk->rnh", x64, wl)  # [R, N, H]
        er_n = torch.einsum("nk,rhk->rnh", x64, wr)
        z = el_n[rel, row] + er_n[rel, col]           # [E, H]
        za = z.abs().min(dim=1).values
        zmin = float(za.min()) if za.numel() else 1.0
        bad = torch.nonzero(za < margin).flatten()
        del el_n, er_n, z, za
        if bad.numel() == 0:
            break
        s, d = row[bad], col[bad]
        move = torch.unique(torch.where(deg[s] <= deg[d], s, d))
        x[move] += step * torch.randn(move.numel(), x.shape[1], device=dev, generator=gen, dtype=x.dtype)
    return x, zmin


def check_round5_gat_pins(ops, dev, gold, lists, inputs, slope, rtol_exp=1e-6, atol_exp=1e-7, rtol_sum=1e-5, atol_sum=1e-6):
    """The round-5 golden vectors of tests/golden/make_golden.py (what the rest of the reference's ref_rgat.py can pin), held against
    ``ops`` -- the oracle module or ``torch.ops.torch_hrt`` -- on device ``dev``:
      kind 1 forward  exp / sum   (el / er on the two-sided unique list, rows found by search)         gatk1_exp, gatk1_sum
      kind 2 forward  sum         (both edge ends through ONE inverse index; the reference loses exp)  gatk2_sum
      kind 4 backward grad_feat   (compact rows summed per source node, as the reference indexes it)   gatcb_grad_feat_src
      kind 1 backward grad_feat   (same)                                                               gatk1b_grad_feat_src
    lists: sep_rel_ptrs / sep_row / sep_col, ts_rel_ptrs / ts_node_indices / ts_inverse_indices, ss_inverse_indices_row / _col,
    ss_node_indices_row (the reference builders' outputs, or ours on the same graph -- they are equal: layout tests);
    inputs: gatk1_el / gatk1_er [U,H], gatc_el / gatc_er, gatb_gradout [N,H,D]."""
    from tests.golden import recipe
    to = lambda t: t.to(dev)  # noqa: E731
    rp, row, col = lists["sep_rel_ptrs"], lists["sep_row"], lists["sep_col"]
    E, n = row.numel(), gold["gatk1_sum"].shape[0]
    go = inputs["gatb_gradout"]
    H, D = go.shape[1], go.shape[2]
    ar = to(torch.arange(E))
    idx = (ar, to(rp), to(row), to(col))
    el, er = inputs["gatk1_el"], inputs["gatk1_er"]
    U = el.shape[0]
    gen = torch.Generator().manual_seed(77)
    featu, ret = torch.randn(U, H, D, generator=gen), torch.randn(n, H, D, generator=gen)
    # ---- kind 1 forward
    d1 = {"unique_srcs_and_dests_rel_ptrs": to(lists["ts_rel_ptrs"]), "unique_srcs_and_dests_node_indices": to(lists["ts_node_indices"])}
    sm, ex, rt = torch.full((n, H), 7.0, device=dev), torch.full((E, H), 7.0, device=dev), torch.full((n, H, D), 7.0, device=dev)
    ops.relational_fused_gat_separate_coo(*idx, 1, d1, to(featu), to(el), to(er), sm, ex, rt, slope)
    torch.testing.assert_close(cpu(ex), gold["gatk1_exp"], rtol=rtol_exp, atol=atol_exp)
    torch.testing.assert_close(cpu(sm), gold["gatk1_sum"], rtol=rtol_sum, atol=atol_sum)
    # ---- kind 2 forward: one inverse index (the row side of the two-sided list's) for both ends
    inv_row, _ = recipe.two_sided_inverse(lists["ts_inverse_indices"], rp)
    d2 = {"edata_idx_to_inverse_idx": to(inv_row)}
    sm2, ex2 = torch.full((n, H), 7.0, device=dev), torch.full((E, H), 7.0, device=dev)
    ops.relational_fused_gat_separate_coo(*idx, 2, d2, to(featu), to(el), to(er), sm2, ex2, rt, slope)
    torch.testing.assert_close(cpu(sm2), gold["gatk2_sum"], rtol=rtol_sum, atol=atol_sum)
    # ---- backward grad_feat, kinds 4 and 1, on the reference's own exp / sum
    d4 = {"edata_idx_to_inverse_idx_row": to(lists["ss_inverse_indices_row"]), "edata_idx_to_inverse_idx_col": to(lists["ss_inverse_indices_col"])}
    S_row = inputs["gatc_el"].shape[0]
    featc = torch.randn(S_row, H, D, generator=gen)
    cases = ((4, d4, featc, inputs["gatc_el"], inputs["gatc_er"], gold["gatc_sum"], gold["gatc_exp"], lists["ss_node_indices_row"],
              "gatcb_grad_feat_src"),
             (1, d1, featu, el, er, gold["gatk1_sum"], gold["gatk1_exp"], lists["ts_node_indices"], "gatk1b_grad_feat_src"))
    for kind, d, feat, l, r, s_ref, e_ref, node_of_row, key in cases:
        gf = torch.zeros(feat.shape[0], H, D, device=dev)
        gl, gr = torch.zeros(l.shape[0], H, device=dev), torch.zeros(r.shape[0], H, device=dev)
        ops.backward_relational_fused_gat_separate_coo(*idx, kind, d, to(feat), to(l), to(r), to(s_ref), to(e_ref), to(ret), to(go),
                                                       gf, gl, gr, slope)
        per_node = torch.zeros(n, H, D, dtype=torch.float64).index_add_(0, node_of_row, cpu(gf).double()).float()
        # fp32 sums of mixed-sign terms over a hub source's edges, in the reference's order on one side and in the kernels' (float
        # atomics when the groupings are off) on the other: the repository's fp32 tolerance, not the 5e-5 that
        # the kind-0 pin happens to meet -- 1 element of 589 104 on the full topology differs by 8.4e-5 relative with atomics
        torch.testing.assert_close(per_node, gold[key], rtol=2e-4, atol=2e-5)
