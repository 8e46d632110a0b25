"""Host logic of the one-node HGT attention (het_amd/backend/hgt_fused_layer.py), no GPU: the folded per-relation weight
w_kv reproduces, row by row, what the reference's composition feeds the edge softmax and the aggregation --
k' = (h . K_st)[h] . att'[r,h] . pri[r,h] / sqrt(dk) and m = (h . V_st)[h] . msg[r,h] (HGT/models.py:159-262) -- and the
destination lists split at the node-type offsets."""
import pytest
import torch

from het_amd.backend.hgt_fused_layer import fold_source_weights


@pytest.mark.parametrize("fused_attn", [False, True])
@pytest.mark.parametrize("H,dk,in_dim", [(1, 8, 8), (2, 4, 6), (8, 8, 64)])
def test_folded_source_weights_match_the_composition(fused_attn, H, dk, in_dim):
    gen = torch.Generator().manual_seed(3)
    T, R, X = 3, 5, H * dk
    k_lin = torch.randn(T, 1, in_dim, X, generator=gen, dtype=torch.float64)
    v_lin = torch.randn(T, 1, in_dim, X, generator=gen, dtype=torch.float64)
    att = torch.randn(R, H, dk, dk, generator=gen, dtype=torch.float64)
    msg = torch.randn(R, H, dk, dk, generator=gen, dtype=torch.float64)
    pri = torch.rand(R, H, generator=gen, dtype=torch.float64) + 0.5
    st = torch.tensor([0, 2, 1, 1, 0])
    w_kv = fold_source_weights(k_lin, v_lin, att, msg, pri, st, H, fused_attn)
    assert w_kv.shape == (R, 1, in_dim, 2 * X)
    h = torch.randn(7, in_dim, generator=gen, dtype=torch.float64)
    for r in range(R):
        k = (h @ k_lin[st[r], 0]).view(-1, H, dk)
        v = (h @ v_lin[st[r], 0]).view(-1, H, dk)
        a = att[r] if fused_attn else att[r].transpose(1, 2)  # s = <k . att, q>  /  s = <q . att, k> = <k . att^T, q>
        k_ref = torch.einsum("nhk,hkd->nhd", k, a) * (pri[r] / dk ** 0.5).view(1, H, 1)
        m_ref = torch.einsum("nhk,hkd->nhd", v, msg[r])
        got = h @ w_kv[r, 0]
        torch.testing.assert_close(got[:, :X].view(-1, H, dk), k_ref, rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(got[:, X:].view(-1, H, dk), m_ref, rtol=1e-12, atol=1e-12)


def test_destination_lists_split_at_the_type_offsets():
    import het_amd.kernels as k
    col = torch.tensor([5, 5, 0, 9, 2, 9, 9, 7])
    offs = torch.tensor([0, 3, 3, 8, 10])  # four node types, the second one empty
    nodes, rank, ptrs = k.destination_lists(col, offs)
    assert nodes.tolist() == [0, 2, 5, 7, 9]
    assert rank.tolist() == [2, 2, 0, 4, 1, 4, 4, 3]
    assert ptrs.tolist() == [0, 2, 2, 4, 5]  # rows of a type are a contiguous piece of the sorted list


@pytest.mark.parametrize("fused_attn", [False, True])
def test_row_formulation_equals_the_oracle_layer(fused_attn):
    """oracle/ops.py::hgt_attention_rows on the folded weights + the typed output projection == oracle/layers.py::hgt_layer
    (the composition the reference's model code spells out) on a small typed graph, in fp64: what the GPU op tests compare
    the row kernels with is the same function as the layer oracle."""
    from oracle import layers as OL
    from oracle import ops as O
    from tests.util import mag_graph
    g = mag_graph(3e-4)
    s = g.get_separate_coo_original()
    ss = g.get_separate_unique_node_indices_single_sided()
    inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
    N, R, T, H, dk, in_dim = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes(), 2, 4, 6
    X = H * dk
    gen = torch.Generator().manual_seed(9)
    rnd = lambda *shape: torch.randn(*shape, generator=gen, dtype=torch.float64) * 0.5
    k_lin, q_lin, v_lin, a_lin = rnd(T, 1, in_dim, X), rnd(T, 1, in_dim, X), rnd(T, 1, in_dim, X), rnd(T, 1, X, X)
    att, msg, pri, skip = rnd(R, H, dk, dk), rnd(R, H, dk, dk), torch.rand(R, H, generator=gen, dtype=torch.float64) + 0.5, rnd(T, 1, 1, 1)
    h = rnd(N, in_dim)
    offs = g.get_original_node_type_offsets()
    ref = OL.hgt_layer(h, offs, s["rel_ptrs"], s["row_indices"], s["col_indices"], N, k_lin, q_lin, v_lin, a_lin, att, msg, pri,
                       skip, H, fused_attn=fused_attn)
    st, _ = g.get_rel_node_types()
    w_kv = fold_source_weights(k_lin, v_lin, att, msg, pri, st, H, fused_attn)
    rp_row, rows_node = ss["rel_ptrs_row"], ss["node_indices_row"]
    rel_of_row = torch.repeat_interleave(torch.arange(R), rp_row[1:] - rp_row[:-1])
    kv_c = torch.einsum("nk,nkx->nx", h[rows_node], w_kv[rel_of_row, 0]).view(-1, 2, H, dk)
    typed = lambda x, W: torch.cat([x[int(offs[t]):int(offs[t + 1])] @ W[t, 0] for t in range(T)])
    q = typed(h, q_lin).view(N, H, dk)
    srow = inv["inverse_indices_row"][s["eids"]]
    _, new_h = O.hgt_attention_rows(kv_c, q, srow, s["col_indices"], N)
    out = typed(new_h.reshape(N, X), torch.sigmoid(skip) * a_lin)
    torch.testing.assert_close(out, ref, rtol=1e-10, atol=1e-12)
