#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own
importable Python (run in the build container, where /root/reference exists;
the GPU box only sees the committed .npz files).

What is pinned:
  * layouts, exactly (int64): integrated COO -> separate COO (bucketed by
    relation, sorted by eid), out-CSR, transposed CSR, single-sided and
    two-sided unique (relation, node) lists with inverse indices -- computed
    by the reference's
      hrt/python/testing/adjacency_manipulation.py
      hrt/python/utils/coo_sorters.py
      hrt/python/utils_lite/mydgl_graph_methods.py
      hrt/python/utils_lite/sparse_matrix_converters.py
    loaded by file path (their packages import DGL, which is absent here);
  * the ``exp`` and ``sum`` outputs of the fused GAT forward, computed by
    hrt/python/testing/ref_kernels_lite/ref_rgat.py (its ``ret`` is never
    written by that file, so ``ret`` is NOT pinned by it);
  * the same two outputs through its dual-unique-list compact wrapper
    (ref_rgat.py:77-115): CompactAsOfNodeKind 4;
  * ``grad_feat_src`` of its backward (ref_rgat.py:66-75).  Its ``grad_el`` /
    ``grad_er`` (:64-65) are NOT pinned: they add ``slope`` to the leaky-ReLU
    derivative and drop the dot product, unlike the CUDA kernel they mirror.
  * round 5 -- what the rest of ref_rgat.py can still pin:
      - CompactAsOfNodeKind 1 forward (``gatk1_exp`` / ``gatk1_sum``): the same
        dual-list wrapper (:77-115) fed el / er on the TWO-SIDED unique list
        and that list's inverse index split into its row and col sides -- the
        rows the CUDA kernel finds by binary search (kernel_enums.h:101-119);
      - CompactAsOfNodeKind 2 forward, ``sum`` only (``gatk2_sum``): the
        single-list wrapper (:182-220) maps BOTH edge ends through one inverse
        index, as RGATKernelsSeparateCOO.cu.h:163-170 does; it writes its
        ``exp`` into a temporary (index_select copy), so ``exp`` is lost;
      - backward ``grad_feat_src`` for kinds 4 and 1 (``gatcb_grad_feat_src``,
        ``gatk1b_grad_feat_src``): ref_rgat.py:66-75 (the unwrapped backward)
        on the ``exp`` / ``sum`` its compact forward wrappers produced.  Node-
        indexed as that file is: the compact rows of our ops are summed per
        node before the comparison.
    What it cannot: the two backward wrappers (:117-180, :222-270) raise for
    every shape (RuntimeError in :64, ``grad_exp [E,H,D] * (el + er) [E,H]``
    assigned into an [E,H] buffer; the single-list one also passes the compact
    el / er where per-edge tensors are indexed) -- run in this container on
    the toy graph, recorded by main() below; their grad_el / grad_er would be
    the ``+ slope`` values of :64-65 anyway.

Three fixtures: the 4-node toy graph of SURVEY.md section 10; a slice of the
only real topology shipped with the reference, hrt/data/ogbn_mag_0.1/*.npy
(first 4096 edges of each of the six relation files), with every input and
output stored; and that topology WHOLE (345 172 edges, the arrays the
reference's own round-trip test loads, hrt/src/test_hyb.cu.cc:26-36) as
mag01_full.npz: the edges as one [3, E] int32 array, the reference builders'
layouts as SHA-256 digests, the reference's float outputs in full; its float
inputs are regenerated from seeds on the test side (tests/golden/recipe.py).

Usage: python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/hrt"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


adj = load("python/testing/adjacency_manipulation.py", "ref_adj")
sorters = load("python/utils/coo_sorters.py", "ref_sorters")
methods = load("python/utils_lite/mydgl_graph_methods.py", "ref_methods")
conv = load("python/utils_lite/sparse_matrix_converters.py", "ref_conv")
ref_rgat = load("python/testing/ref_kernels_lite/ref_rgat.py", "ref_rgat")


def layouts(row, col, rel, eids, num_rels):
    out = {"row": row, "col": col, "rel": rel, "eids": eids}
    rp, r, c, e = adj.convert_integrated_coo_to_separate_coo(row, col, rel, eids)
    rp, r, c, e = sorters.sort_coo_by_etype_eids_torch_tensors(rp, r, c, e)
    out.update(sep_rel_ptrs=rp, sep_row=r, sep_col=c, sep_eids=e)
    nr, pr, nc, pc, ir, ic = methods.generate_separate_unique_node_indices_single_sided_for_each_etype(
        num_rels, rp, r, c, get_inverse_idx=True)
    out.update(ss_node_indices_row=nr, ss_rel_ptrs_row=pr, ss_node_indices_col=nc, ss_rel_ptrs_col=pc,
               ss_inverse_indices_row=ir, ss_inverse_indices_col=ic)
    nu, pu, iu = methods.generate_separate_unique_node_indices_for_each_etype(num_rels, rp, r, c, get_inverse_idx=True)
    out.update(ts_node_indices=nu, ts_rel_ptrs=pu, ts_inverse_indices=iu)
    # out-CSR by the reference converter; its argsort is not stable, so only the
    # order-independent parts (row_ptrs, and the per-row multiset) are compared.
    ptr, ccol, crel, ceid = conv.coo2csr(row, col, rel, eids, torch_flag=True)
    out.update(csr_row_ptrs=ptr, csr_col=ccol, csr_rel=crel, csr_eids=ceid)
    tptr, tcol, teid, trel = adj.transpose_csr(ptr, ccol, ceid, crel)
    out.update(tcsr_row_ptrs=tptr, tcsr_col=tcol, tcsr_eids=teid, tcsr_rel=trel)
    return out


def gat_exp_sum(lay, num_nodes, H, seed):
    g = torch.Generator().manual_seed(seed)
    E = lay["sep_row"].numel()
    el = torch.randn(E, H, generator=g)
    er = torch.randn(E, H, generator=g)
    feat = torch.randn(max(E, num_nodes), H, 2, generator=g)  # only feeds the (discarded) ret
    s = torch.zeros(num_nodes, H)
    exp = torch.zeros(E, H)
    ret = torch.zeros(num_nodes, H, 2)
    # the reference function is written for eids == arange (canonicalised separate COO)
    eids = torch.arange(E)
    ref_rgat.relational_fused_gat_separate_coo(eids, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"],
                                              feat, el, er, s, exp, ret, 0.2)
    return {"gat_el": el, "gat_er": er, "gat_exp": exp, "gat_sum": s, "gat_slope": torch.tensor(0.2)}


def gat_backward_grad_feat_src(lay, num_nodes, H, D, seed):
    """grad_feat_src of the reference's own backward (ref_rgat.py:66-75: += gradout[col] * exp / s[col], node-indexed).
    The function is run UNMODIFIED; its grad_el / grad_er statements (:64-65, wrong by "+ slope" and shaped for D == 1) are
    fed shapes they do not raise on ([E,H,1] pre-activations, [E,H,D] dummies) and their outputs are discarded."""
    g = torch.Generator().manual_seed(seed)
    E = lay["sep_row"].numel()
    el, er = lay["gat_el"], lay["gat_er"]  # the pre-activations whose exp / sum the forward fixture pins
    exp, s = lay["gat_exp"], lay["gat_sum"]
    feat_src = torch.randn(num_nodes, H, D, generator=g)
    ret = torch.randn(num_nodes, H, D, generator=g)
    gradout = torch.zeros(num_nodes, H, D)  # random on the destination nodes only (the others are never read): small file
    dst = torch.unique(lay["sep_col"])
    gradout[dst] = torch.randn(dst.numel(), H, D, generator=g)
    grad_feat_src = torch.zeros(num_nodes, H, D)
    dummy_l, dummy_r = torch.zeros(E, H, D), torch.zeros(E, H, D)
    ref_rgat.backward_relational_fused_gat_separate_coo(torch.arange(E), lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"],
                                                       feat_src, el.unsqueeze(-1), er.unsqueeze(-1), s, exp, ret, gradout,
                                                       grad_feat_src, dummy_l, dummy_r, 0.2)
    return {"gatb_gradout": gradout, "gatb_grad_feat_src": grad_feat_src}


def gat_compact_exp_sum(lay, num_nodes, H, seed):
    """exp / sum through the reference's dual-unique-list compact wrapper (ref_rgat.py:77-115): el per distinct
    (relation, source) row, er per distinct (relation, destination) row, expanded with the inverse indices of the
    single-sided unique lists -- CompactAsOfNodeKind 4 of the fused GAT forward."""
    g = torch.Generator().manual_seed(seed)
    E = lay["sep_row"].numel()
    el_c = torch.randn(lay["ss_node_indices_row"].numel(), H, generator=g)
    er_c = torch.randn(lay["ss_node_indices_col"].numel(), H, generator=g)
    feat = torch.randn(num_nodes, H, 2, generator=g)  # feeds the (discarded) ret only
    s, exp, ret = torch.zeros(num_nodes, H), torch.zeros(E, H), torch.zeros(num_nodes, H, 2)
    ref_rgat.towrap_relational_fused_gat_kernel_compact_as_of_node_separate_coo_dual_unique_node_list(
        lay["ss_inverse_indices_row"], lay["ss_inverse_indices_col"], torch.arange(E), lay["sep_rel_ptrs"], lay["sep_row"],
        lay["sep_col"], lay["ss_rel_ptrs_row"], lay["ss_rel_ptrs_col"], lay["ss_node_indices_row"], lay["ss_node_indices_col"],
        feat, el_c, er_c, s, exp, ret, 0.2)
    return {"gatc_el": el_c, "gatc_er": er_c, "gatc_exp": exp, "gatc_sum": s}


def gat_round5(lay, num_nodes, H, D, gradout, seed=None, inputs=None):
    """The round-5 pins (module docstring).  Needs gatc_* (kind-4 forward) in ``lay``; gradout [num_nodes,H,D]."""
    E = lay["sep_row"].numel()
    ar = torch.arange(E)
    U = lay["ts_node_indices"].numel()
    if inputs is None:
        g = torch.Generator().manual_seed(seed)
        inputs = {"gatk1_el": torch.randn(U, H, generator=g), "gatk1_er": torch.randn(U, H, generator=g)}
    el, er = inputs["gatk1_el"], inputs["gatk1_er"]
    inv_row, inv_col = recipe.two_sided_inverse(lay["ts_inverse_indices"], lay["sep_rel_ptrs"])
    out = dict(inputs)
    # kind 1: dual wrapper, both lists = the two-sided list
    s1, exp1 = torch.zeros(num_nodes, H), torch.zeros(E, H)
    ref_rgat.towrap_relational_fused_gat_kernel_compact_as_of_node_separate_coo_dual_unique_node_list(
        inv_row, inv_col, ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"], lay["ts_rel_ptrs"], lay["ts_rel_ptrs"],
        lay["ts_node_indices"], lay["ts_node_indices"], torch.zeros(num_nodes, H, 2), el, er, s1, exp1, torch.zeros(num_nodes, H, 2), 0.2)
    out["gatk1_exp"], out["gatk1_sum"] = exp1, s1
    # kind 2: single-list wrapper, one inverse index (the row side) for both ends; exp stays in its temporary
    s2, exp2 = torch.zeros(num_nodes, H), torch.zeros(U, H)
    ref_rgat.towrap_relational_fused_gat_kernel_compact_as_of_node_separate_coo(
        inv_row, ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"], lay["ts_rel_ptrs"], lay["ts_node_indices"],
        torch.zeros(num_nodes, H, 2), el, er, s2, exp2, torch.zeros(num_nodes, H, 2), 0.2)
    assert float(exp2.abs().sum()) == 0.0  # (the wrapper's exp never reaches the caller)
    out["gatk2_sum"] = s2
    # backward grad_feat_src (node-indexed) on the compact forwards' exp / sum: kinds 4 and 1
    for key, exp, s in (("gatcb_grad_feat_src", lay["gatc_exp"], lay["gatc_sum"]), ("gatk1b_grad_feat_src", exp1, s1)):
        gfs = torch.zeros(num_nodes, H, D)
        z = torch.zeros(E, H, 1)  # (el / er only feed the discarded grad_el / grad_er lines)
        ref_rgat.backward_relational_fused_gat_separate_coo(ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"],
                                                           torch.zeros(num_nodes, H, D), z, z, s, exp, torch.zeros(num_nodes, H, D),
                                                           gradout, gfs, torch.zeros(E, H, D), torch.zeros(E, H, D), 0.2)
        out[key] = gfs
    return out


def probe_backward_wrappers(lay, num_nodes, H, D):
    """Runs the reference's two backward wrappers as written and returns what they raise (ordinary Python errors)."""
    E = lay["sep_row"].numel()
    ar = torch.arange(E)
    inv_row, _ = recipe.two_sided_inverse(lay["ts_inverse_indices"], lay["sep_rel_ptrs"])
    U = lay["ts_node_indices"].numel()
    notes = []
    try:
        ref_rgat.towrap_backward_relational_fused_gat_compact_as_of_node_separate_coo_dual_unique_node_list(
            lay["ss_inverse_indices_row"], lay["ss_inverse_indices_col"], ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"],
            lay["ss_rel_ptrs_row"], lay["ss_rel_ptrs_col"], lay["ss_node_indices_row"], lay["ss_node_indices_col"],
            torch.zeros(num_nodes, H, D), lay["gatc_el"], lay["gatc_er"], lay["gatc_sum"], lay["gatc_exp"], torch.zeros(num_nodes, H, D),
            torch.zeros(num_nodes, H, D), torch.zeros(num_nodes, H, D), torch.zeros_like(lay["gatc_el"]), torch.zeros_like(lay["gatc_er"]), 0.2)
        notes.append("dual-list backward wrapper (:117-180): ran")
    except Exception as e:  # noqa: BLE001
        notes.append(f"dual-list backward wrapper (:117-180): {type(e).__name__}: {e}")
    try:
        ref_rgat.towrap_backward_relational_fused_gat_compact_as_of_node_separate_coo(
            inv_row, ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"], lay["ts_rel_ptrs"], lay["ts_node_indices"],
            torch.zeros(num_nodes, H, D), torch.zeros(U, H), torch.zeros(U, H), torch.ones(num_nodes, H), torch.zeros(U, H),
            torch.zeros(num_nodes, H, D), torch.zeros(num_nodes, H, D), torch.zeros(num_nodes, H, D), torch.zeros(U, H), torch.zeros(U, H), 0.2)
        notes.append("single-list backward wrapper (:222-270): ran")
    except Exception as e:  # noqa: BLE001
        notes.append(f"single-list backward wrapper (:222-270): {type(e).__name__}: {e}")
    return notes


def main_full():
    """The whole shipped topology (fixture 2)."""
    import json
    parts = []
    for r, n in enumerate(recipe.MAG01_RELATIONS):
        a = np.load(os.path.join(REF, "data/ogbn_mag_0.1", f"{n}_coo_2.npy"))
        assert a.dtype == np.int32 and a.shape[0] == 2
        parts.append(np.concatenate([a, np.full((1, a.shape[1]), r, dtype=np.int32)], axis=0))
    coo3 = np.ascontiguousarray(np.concatenate(parts, axis=1))  # [3, E]: row (source), col (destination), relation; file order
    row, col, rel, eids, n, R = recipe.integrated_coo(coo3)
    E = row.numel()
    lay = layouts(row, col, rel, eids, R)
    H, D = recipe.H, recipe.D
    inp = recipe.gat_inputs(E, n, lay["ss_node_indices_row"].numel(), lay["ss_node_indices_col"].numel())
    out = {"coo": coo3, "num_nodes": np.int64(n), "num_rels": np.int64(R)}
    # -- layouts: digests of everything, the small arrays also in full
    dig = {}
    for k in ("sep_rel_ptrs", "sep_row", "sep_col", "sep_eids", "ss_node_indices_row", "ss_rel_ptrs_row", "ss_node_indices_col",
              "ss_rel_ptrs_col", "ss_inverse_indices_row", "ss_inverse_indices_col", "ts_node_indices", "ts_rel_ptrs",
              "ts_inverse_indices", "csr_row_ptrs", "tcsr_row_ptrs"):
        dig[k] = recipe.digest(lay[k])
    for pre in ("csr", "tcsr"):  # rows as multisets: entries sorted by (col, rel, eid) inside every row (recipe.canonical_csr)
        c, r_, e = recipe.canonical_csr(lay[pre + "_row_ptrs"], lay[pre + "_col"], lay[pre + "_rel"], lay[pre + "_eids"])
        dig[pre + "_col_canonical"], dig[pre + "_rel_canonical"], dig[pre + "_eids_canonical"] = (recipe.digest(c), recipe.digest(r_),
                                                                                                  recipe.digest(e))
    for k in ("sep_rel_ptrs", "ss_rel_ptrs_row", "ss_rel_ptrs_col", "ts_rel_ptrs"):
        out[k] = lay[k]
    out["csr_row_ptrs"] = lay["csr_row_ptrs"].to(torch.int32)
    out["tcsr_row_ptrs"] = lay["tcsr_row_ptrs"].to(torch.int32)
    # -- the reference's float outputs on the regenerated inputs
    for k, v in inp.items():
        dig["input_" + k] = recipe.digest(v)
    s, exp, ret = torch.zeros(n, H), torch.zeros(E, H), torch.zeros(n, H, 2)
    feat = torch.zeros(max(E, n), H, 2)  # feeds the (discarded) ret only
    ar = torch.arange(E)
    ref_rgat.relational_fused_gat_separate_coo(ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"], feat, inp["gat_el"],
                                              inp["gat_er"], s, exp, ret, recipe.SLOPE)
    out["gat_exp"], out["gat_sum"] = exp, s
    gfs = torch.zeros(n, H, D)
    ref_rgat.backward_relational_fused_gat_separate_coo(ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"], torch.zeros(n, H, D),
                                                       inp["gat_el"].unsqueeze(-1), inp["gat_er"].unsqueeze(-1), s, exp,
                                                       torch.zeros(n, H, D), inp["gatb_gradout"], gfs, torch.zeros(E, H, D),
                                                       torch.zeros(E, H, D), recipe.SLOPE)
    out["gatb_grad_feat_src"] = gfs
    s2, exp2 = torch.zeros(n, H), torch.zeros(E, H)
    ref_rgat.towrap_relational_fused_gat_kernel_compact_as_of_node_separate_coo_dual_unique_node_list(
        lay["ss_inverse_indices_row"], lay["ss_inverse_indices_col"], ar, lay["sep_rel_ptrs"], lay["sep_row"], lay["sep_col"],
        lay["ss_rel_ptrs_row"], lay["ss_rel_ptrs_col"], lay["ss_node_indices_row"], lay["ss_node_indices_col"],
        torch.zeros(n, H, 2), inp["gatc_el"], inp["gatc_er"], s2, exp2, torch.zeros(n, H, 2), recipe.SLOPE)
    out["gatc_exp"], out["gatc_sum"] = exp2, s2
    # round 5: kinds 1 / 2 forward, kinds 4 / 1 backward grad_feat_src (inputs regenerated on the test side: recipe.gat_inputs_round5)
    inp5 = recipe.gat_inputs_round5(lay["ts_node_indices"].numel())
    for k, v in inp5.items():
        dig["input_" + k] = recipe.digest(v)
    lay5 = dict(lay); lay5["gatc_exp"], lay5["gatc_sum"] = exp2, s2
    r5 = gat_round5(lay5, n, H, D, inp["gatb_gradout"], inputs=inp5)
    for k in ("gatk1_exp", "gatk1_sum", "gatk2_sum", "gatcb_grad_feat_src", "gatk1b_grad_feat_src"):
        out[k] = r5[k]
    out["digests_json"] = np.array(json.dumps(dig, sort_keys=True))
    save("mag01_full.npz", out)


def save(name, d):
    arrs = {}
    for k, v in d.items():
        a = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        arrs[k] = a
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print(name, {k: tuple(a.shape) for k, a in arrs.items()})


def main():
    # toy graph
    row = torch.tensor([0, 1, 2, 2, 3, 0])
    col = torch.tensor([1, 2, 0, 3, 1, 3])
    rel = torch.tensor([0, 0, 1, 1, 0, 1])
    eids = torch.arange(6)
    lay = layouts(row, col, rel, eids, 2)
    lay.update(gat_exp_sum(lay, 4, 2, seed=1))
    lay.update(gat_backward_grad_feat_src(lay, 4, 2, 3, seed=11))
    lay.update(gat_compact_exp_sum(lay, 4, 2, seed=21))
    lay.update(gat_round5(lay, 4, 2, 3, lay["gatb_gradout"], seed=31))
    for note in probe_backward_wrappers(lay, 4, 2, 3):
        print("reference, as written, on the toy graph --", note)
    lay["num_nodes"], lay["num_rels"] = torch.tensor(4), torch.tensor(2)
    save("toy.npz", lay)

    # slice of the shipped ogbn_mag_0.1 topology; interleave the relations so the
    # integrated COO is NOT relation-sorted and eids are a non-trivial permutation
    names = ["cited", "citing", "has", "is-about", "writing", "written-by"]
    rows, cols, rels = [], [], []
    for r, n in enumerate(names):
        a = np.load(os.path.join(REF, "data/ogbn_mag_0.1", f"{n}_coo_2.npy"))[:, :4096].astype(np.int64)
        rows.append(a[0]); cols.append(a[1]); rels.append(np.full(a.shape[1], r, dtype=np.int64))
    row = torch.from_numpy(np.concatenate(rows)); col = torch.from_numpy(np.concatenate(cols))
    rel = torch.from_numpy(np.concatenate(rels))
    perm = torch.randperm(row.numel(), generator=torch.Generator().manual_seed(7))
    row, col, rel = row[perm], col[perm], rel[perm]
    eids = torch.randperm(row.numel(), generator=torch.Generator().manual_seed(8))
    lay = layouts(row, col, rel, eids, 6)
    n = int(max(row.max(), col.max())) + 1
    lay.update(gat_exp_sum(lay, n, 4, seed=2))
    lay.update(gat_backward_grad_feat_src(lay, n, 4, 4, seed=12))
    lay.update(gat_compact_exp_sum(lay, n, 4, seed=22))
    lay.update(gat_round5(lay, n, 4, 4, lay["gatb_gradout"], seed=32))
    lay["num_nodes"], lay["num_rels"] = torch.tensor(n), torch.tensor(6)
    save("mag01_slice.npz", lay)
    main_full()


if __name__ == "__main__":
    sys.exit(main())
