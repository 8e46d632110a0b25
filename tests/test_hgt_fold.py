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
