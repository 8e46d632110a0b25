"""Layer-level CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/ops.py).

Plain-PyTorch message passing (index_select -> per-relation matmul ->
leaky_relu / exp -> index_add), differentiable by torch autograd, restating what
one HET layer computes end to end (reference compositions:
hrt/python/RGAT/models.py:265-385, hrt/python/RGCN/RGCN.py:264-350).  It is the
checker for the layer-level parity tests and the timed ``cpu_baseline`` of
bench.py ("port": the reference's own CPU fallback is its DGL model, which cannot
run without DGL).  HET semantics: the edge softmax normalises over ALL in-edges
of a destination, across relations (RGAT/RGATKernelsSeparateCOO.cu.h:190-196),
unlike DGL's per-relation HeteroGraphConv.
"""
import torch


def rel_of_position(rel_ptrs):
    R = rel_ptrs.numel() - 1
    return torch.repeat_interleave(torch.arange(R, device=rel_ptrs.device), rel_ptrs[1:] - rel_ptrs[:-1])


def rgat_layer(x, conv_weights, attn_l, attn_r, rel_ptrs, row, col, num_nodes, slope=0.2,
               loop_weight=None, h_bias=None):
    """x [N,K]; conv_weights [R,H,K,D]; attn_l/attn_r [R,H,D]; separate COO (eids = arange).  Returns [N, H*D]."""
    R, H, K, D = conv_weights.shape
    feat = x.new_empty((row.numel(), H, D))
    el = x.new_empty((row.numel(), H))
    er = x.new_empty((row.numel(), H))
    parts_f, parts_l, parts_r = [], [], []
    for r in range(R):
        a, b = int(rel_ptrs[r]), int(rel_ptrs[r + 1])
        Wr = conv_weights[r].permute(1, 0, 2).reshape(K, H * D)
        fs = (x[row[a:b]] @ Wr).view(-1, H, D)
        fd = (x[col[a:b]] @ Wr).view(-1, H, D)
        parts_f.append(fs)
        parts_l.append((fs * attn_l[r]).sum(-1))
        parts_r.append((fd * attn_r[r]).sum(-1))
    feat, el, er = torch.cat(parts_f), torch.cat(parts_l), torch.cat(parts_r)
    z = el + er
    e = torch.exp(torch.where(z > 0, z, z * slope))
    den = torch.zeros(num_nodes, H, dtype=x.dtype, device=x.device).index_add(0, col, e)
    a = e / den[col]
    h = torch.zeros(num_nodes, H, D, dtype=x.dtype, device=x.device).index_add(0, col, a.unsqueeze(-1) * feat)
    h = h.view(num_nodes, H * D)
    if loop_weight is not None:
        h = h + x @ loop_weight
    if h_bias is not None:
        h = h + h_bias
    return h


def rgcn_layer(x, weight, norm, rel_ptrs, row, col, num_nodes, h_bias=None):
    """x [N,K]; weight [R,K,D]; norm [E] or [E,1] (eids = arange).  Returns [N, D]."""
    R = weight.shape[0]
    nv = norm.reshape(-1, 1)
    out = torch.zeros(num_nodes, weight.shape[2], dtype=x.dtype, device=x.device)
    for r in range(R):
        a, b = int(rel_ptrs[r]), int(rel_ptrs[r + 1])
        if a == b:
            continue
        out = out.index_add(0, col[a:b], (x[row[a:b]] * nv[a:b]) @ weight[r])
    if h_bias is not None:
        out = out + h_bias
    return out


def hgt_layer(h, node_type_offsets, rel_ptrs, row, col, num_nodes, k_lin, q_lin, v_lin, a_lin, rel_att, rel_msg, rel_pri,
              skip, num_heads, fused_attn=False):
    """One HGT layer as the reference composes it (hrt/python/HGT/models.py:120-286, non-compact path).
    k_lin/q_lin/v_lin/a_lin [T,1,in,out]; rel_att/rel_msg [R,H,dk,dk]; rel_pri [R,H]; skip [T,1,1,1].
    Unfused score:  s = < q[dst] @ rel_att[r,h], k[src] >;  fused op:  s = < k[src] @ rel_att[r,h], q[dst] >."""
    T = node_type_offsets.numel() - 1
    H = num_heads
    out_dim = k_lin.shape[3]
    dk = out_dim // H

    def typed_linear(x, W):
        parts = []
        for t in range(T):
            a, b = int(node_type_offsets[t]), int(node_type_offsets[t + 1])
            parts.append(x[a:b] @ W[t, 0])
        return torch.cat(parts)

    k = typed_linear(h, k_lin).view(-1, H, dk)
    q = typed_linear(h, q_lin).view(-1, H, dk)
    v = typed_linear(h, v_lin).view(-1, H, dk)
    rel = rel_of_position(rel_ptrs)
    R = rel_ptrs.numel() - 1

    def per_relation(x_rows, W):  # [E,H,dk] rows times the [H,dk,dk] matrix of each row's relation (relation-bucketed rows)
        parts = []
        for r in range(R):
            a, b = int(rel_ptrs[r]), int(rel_ptrs[r + 1])
            parts.append(torch.einsum("nhk,hkd->nhd", x_rows[a:b], W[r]))
        return torch.cat(parts)

    if fused_attn:
        s = (per_relation(k[row], rel_att) * q[col]).sum(-1)
    else:
        s = (per_relation(q[col], rel_att) * k[row]).sum(-1)
    mu = (rel_pri / (dk ** 0.5))[rel]
    m = torch.exp(s * mu)
    den = torch.zeros(num_nodes, H, dtype=h.dtype, device=h.device).index_add(0, col, m)
    a = m / den[col]
    msg = per_relation(v[row] * a.unsqueeze(-1), rel_msg)
    new_h = torch.zeros(num_nodes, H, dk, dtype=h.dtype, device=h.device).index_add(0, col, msg).view(num_nodes, out_dim)
    return typed_linear(new_h, torch.sigmoid(skip) * a_lin)
