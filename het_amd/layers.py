"""Layer modules: the op compositions of the reference's model scripts, without
DGL.  Parameter names, shapes, initialisation and the order of ``B.*`` calls
follow the reference so that a state_dict and the kernel sequence line up:

  HET_RGATLayer                       hrt/python/RGAT/models.py:16-385
  HET_EglRelGraphConv_EdgeParallel    hrt/python/RGCN/RGCN.py:194-350
  HET_RelGraphEmbed                   hrt/python/RGNNUtils/RGNNUtils.py:78-119
  HET_HGTLayerHetero                  hrt/python/HGT/models.py:15-286
"""
from __future__ import annotations

import math

import os

import torch as th
import torch.nn as nn

from . import backend as B
from .backend import rgat_fused_layer as FL
from .backend import hgt_fused_layer as _hgt_fused


PAD_HEADS = os.environ.get("HET_RGAT_PAD_HEADS", "1") != "0"  # zero-pad narrow / odd RGAT heads to the row kernels' widths
PAD_WIDTHS = os.environ.get("HET_PAD_WIDTHS", "1") != "0"     # zero-pad RGCN output widths to 32 / 64 / 128


def _padded_in_width(K: int) -> int:
    """Input width the matrix-core projections run with: 32 / 64 / 128.  Other widths up to 128 (16, 100 ...) run zero-padded --
    zero columns appended to the input rows and zero rows to every weight that multiplies them add nothing to any product; the
    padded columns of the input gradient are dropped by autograd.  One padded copy of the input per step (N x K' floats) against
    the any-shape GEMMs: ogbn-mag, 100 -> 64: RGAT 27 -> 7 ms, RGCN 35 -> 5 ms."""
    if not PAD_WIDTHS:
        return K
    return next((w for w in (32, 64, 128) if w >= K), K)


class HET_RelGraphEmbed(nn.Module):
    """Learnable node embeddings used as input features (RGNNUtils.py:78-119)."""

    def __init__(self, num_nodes: int, embed_size: int):
        super().__init__()
        self.embeds = nn.Parameter(th.Tensor(num_nodes, embed_size))
        nn.init.xavier_uniform_(self.embeds)

    def forward(self):
        return self.embeds


class HET_RGATLayer(nn.Module):
    """Relational graph attention layer (RGAT/models.py:16-385)."""

    def __init__(self, in_feat, out_feat, num_rels, num_heads, *, bias=True, activation=None, self_loop=False,
                 compact_as_of_node_flag=False, compact_direct_indexing_flag=False,
                 multiply_among_weights_first_flag=False, gat_edge_parallel_flag=True, dropout=0.5,
                 leaky_relu_slope=0.2, reference_op_sequence=False):
        super().__init__()
        assert out_feat % num_heads == 0, "out_feat must be a multiple of num_heads"
        self.in_feat, self.out_feat, self.num_rels, self.num_heads = in_feat, out_feat, num_rels, num_heads
        self.bias, self.activation, self.self_loop = bias, activation, self_loop
        self.compact_as_of_node_flag = compact_as_of_node_flag
        self.compact_direct_indexing_flag = compact_direct_indexing_flag
        self.multiply_among_weights_first_flag = multiply_among_weights_first_flag
        self.gat_edge_parallel_flag = gat_edge_parallel_flag
        self.leaky_relu_slope = leaky_relu_slope
        # True: run the reference's op sequence literally (only reference-named torch_hrt ops, the reference wrappers'
        # zero-filled "+=" buffers; het_amd/backend/reference_protocol.py) -- the drop-in path bench.py times as
        # variants.reference_op_sequence.  Non-compact flags, full graph.
        self.reference_op_sequence = reference_op_sequence
        # True (or HET_RGAT_FUSED=0): the reference's model code line by line on this package's backend wrappers (het_amd.backend:
        # same function names as hrt/python/backend, buffers allocated without fills, "=" gradients) instead of the one-node layer --
        # what a reference checkout gets when it imports het_amd.backend in place of its own backend package
        self.op_by_op = os.environ.get("HET_RGAT_FUSED", "1") == "0"
        self.conv_weights = nn.Parameter(th.Tensor(num_rels, num_heads, in_feat, out_feat // num_heads))
        self.attn_l = nn.Parameter(th.Tensor(num_rels, num_heads, out_feat // num_heads))
        self.attn_r = nn.Parameter(th.Tensor(num_rels, num_heads, out_feat // num_heads))
        if bias:
            self.h_bias = nn.Parameter(th.Tensor(out_feat))
        if self_loop:
            self.loop_weight = nn.Parameter(th.Tensor(in_feat, out_feat))
        self.dropout = nn.Dropout(dropout)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if self.bias:
            nn.init.zeros_(self.h_bias)
        if self.self_loop:
            nn.init.xavier_uniform_(self.loop_weight, gain=gain)
        nn.init.xavier_uniform_(self.conv_weights, gain=gain)
        nn.init.xavier_uniform_(self.attn_l, gain=gain)
        nn.init.xavier_uniform_(self.attn_r, gain=gain)

    def _w_attn_r(self):
        dk = self.out_feat // self.num_heads
        return th.bmm(self.conv_weights.view(-1, self.in_feat, dk), self.attn_r.view(-1, dk, 1)).view(
            -1, self.num_heads, self.in_feat, 1)

    def forward_with_halo(self, g, x_own: th.Tensor, halo):
        """Multi-GPU form (het_amd/dist.py): ``g`` is this rank's local graph (owned nodes first, then halo nodes), ``x_own``
        the features of the owned nodes; the layer runs the halo exchange itself (``halo``: a dist.HaloContext) and overlaps
        it with the work that needs owned rows only.  Returns the owned rows, or None when the one-node path does not
        cover the configuration (the caller then exchanges first and calls forward)."""
        if not (self.gat_edge_parallel_flag and self.self_loop and not self.reference_op_sequence and
                FL.rgat_layer_halo_ok(g, x_own, self.conv_weights, self.leaky_relu_slope, self.compact_as_of_node_flag,
                                      self.multiply_among_weights_first_flag)):
            return None
        h = FL.rgat_layer_fused(g, x_own, self.conv_weights, self.attn_l, self.attn_r, self.loop_weight,
                                self.h_bias if self.bias else None, self.leaky_relu_slope, self.compact_as_of_node_flag,
                                self.compact_direct_indexing_flag, x_own.shape[0], self.multiply_among_weights_first_flag,
                                halo=halo)
        if self.activation:
            h = self.activation(h)
        return self.dropout(h)

    def _padded_head(self, g, inputs):
        """(input width K', head width D') the one-node layer should run with, or None.  Narrow outputs -- the reference's own RGAT
        experiment is 128 -> 8 classes over 8 heads, one float per head (hrt/experiments/run_het_rgat.sh) -- and heads whose
        width is not a power of two are outside the row kernels (4+ floats per head, 32+ per row for the matrix-core
        projection).  Zero-padding every head to D' changes no value: the padded columns of W, attn_l, attn_r, the self-loop
        weight and the bias are zero, so el / er, the softmax and the first D components of every head are the same sums and
        the padded output components are zero (dropped below); the gradients of the padding are dropped by autograd.
        HET_RGAT_PAD_HEADS=0: off (op-by-op composition / any-shape kernels as before)."""
        if not (PAD_HEADS and self.gat_edge_parallel_flag and not self.op_by_op and inputs.is_cuda):
            return None
        H, D, K = self.num_heads, self.out_feat // self.num_heads, self.in_feat
        if H & (H - 1) or H > 32:
            return None
        Dp = max(4, 32 // H, 1 << max(0, D - 1).bit_length())
        Kp = _padded_in_width(K)  # (input widths outside 32 / 64 / 128: zero columns of x, zero rows of the weights)
        if (Dp == D and Kp == K) or H * Dp > 256:
            return None
        shape = th.empty((self.num_rels, H, Kp, Dp), device="meta")
        ok = FL.rgat_layer_fused_ok(g, inputs, shape, self.leaky_relu_slope, self.compact_as_of_node_flag,
                                    self.multiply_among_weights_first_flag)
        return (Kp, Dp) if ok else None

    def _forward_padded(self, g, inputs, num_dst, KDp, halo=None):
        H, D, K = self.num_heads, self.out_feat // self.num_heads, self.in_feat
        Kp, Dp = KDp
        pad = (0, Dp - D)
        x = nn.functional.pad(inputs, (0, Kp - K)) if Kp != K else inputs
        W = nn.functional.pad(self.conv_weights, pad + (0, Kp - K))
        al, ar = nn.functional.pad(self.attn_l, pad), nn.functional.pad(self.attn_r, pad)
        loop = nn.functional.pad(self.loop_weight.view(K, H, D), pad + (0, 0, 0, Kp - K)).view(Kp, H * Dp) if self.self_loop else None
        bias = nn.functional.pad(self.h_bias.view(H, D), pad).view(H * Dp) if self.bias else None
        h = FL.rgat_layer_fused(g, x, W, al, ar, loop, bias, self.leaky_relu_slope, self.compact_as_of_node_flag,
                                self.compact_direct_indexing_flag, num_dst, self.multiply_among_weights_first_flag, halo=halo)
        h = h.view(h.shape[0], H, Dp)[:, :, :D].reshape(h.shape[0], self.out_feat)
        if self.activation:
            h = self.activation(h)
        return self.dropout(h)

    def forward(self, g, inputs: th.Tensor, num_dst=None):
        """``num_dst``: the destination nodes of ``g`` are its first ``num_dst`` nodes (a sampled block, or the owned
        nodes of a partition followed by halo nodes): only their rows are returned and the self-loop runs on them only."""
        if self.reference_op_sequence:
            assert not self.compact_as_of_node_flag and num_dst is None and self.gat_edge_parallel_flag
            from .backend.reference_protocol import rgat_layer_reference_sequence
            return rgat_layer_reference_sequence(self, g, inputs)
        KDp = self._padded_head(g, inputs)
        if KDp is not None:
            return self._forward_padded(g, inputs, num_dst, KDp)
        if (self.gat_edge_parallel_flag and not self.op_by_op and
                FL.rgat_layer_fused_ok(g, inputs, self.conv_weights, self.leaky_relu_slope, self.compact_as_of_node_flag,
                                       self.multiply_among_weights_first_flag)):
            # the whole layer as one autograd node (het_amd/backend/rgat_fused_layer.py): same ops and values as the
            # composition below, gradients of the shared input accumulated in place instead of summed by autograd
            h = FL.rgat_layer_fused(g, inputs, self.conv_weights, self.attn_l, self.attn_r,
                                    self.loop_weight if self.self_loop else None, self.h_bias if self.bias else None,
                                    self.leaky_relu_slope, self.compact_as_of_node_flag, self.compact_direct_indexing_flag,
                                    num_dst, self.multiply_among_weights_first_flag)
            if self.activation:
                h = self.activation(h)
            return self.dropout(h)
        if self.compact_as_of_node_flag:  # models.py:152-263
            ss = g.get_separate_unique_node_indices_single_sided()
            d_row = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"],
                     "unique_srcs_and_dests_node_indices": ss["node_indices_row"]}
            d_col = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"],
                     "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}
            feat_compact = B.rgnn_relational_matmul(d_row, self.conv_weights, inputs, True, 1)
            fuse_el = self.gat_edge_parallel_flag and B.relational_fused_gat_compact_with_attn_l_ok(
                g, feat_compact, self.attn_l, self.leaky_relu_slope)
            if not fuse_el:
                el_compact = B.rgnn_relational_matmul_no_scatter_gather_list(
                    ss["rel_ptrs_row"], self.attn_l.unsqueeze(-1), feat_compact)
            if self.multiply_among_weights_first_flag:
                er_compact = B.rgnn_relational_matmul(d_col, self._w_attn_r(), inputs, True, 1)
            else:
                feat_compact_dst = B.rgnn_relational_matmul(d_col, self.conv_weights, inputs, True, 1)
                er_compact = B.rgnn_relational_matmul_no_scatter_gather_list(
                    ss["rel_ptrs_col"], self.attn_r.unsqueeze(-1), feat_compact_dst)
            if not self.gat_edge_parallel_flag:
                raise NotImplementedError("single-sided unique node lists need the edge-parallel op (models.py:253-256)")
            if fuse_el:  # el and the GAT op under one autograd node (same values, one gradient store for feat_compact)
                h = B.relational_fused_gat_compact_with_attn_l(
                    g, feat_compact, self.attn_l, er_compact.view(er_compact.shape[0], self.num_heads),
                    self.leaky_relu_slope, self.compact_direct_indexing_flag)
            else:
                h = B.relational_fused_gat_compact_as_of_node_separate_coo_single_sided(
                    g, feat_compact, el_compact.view(el_compact.shape[0], self.num_heads),
                    er_compact.view(er_compact.shape[0], self.num_heads), self.leaky_relu_slope,
                    self.compact_direct_indexing_flag)
        else:  # models.py:265-372
            s = g.get_separate_coo_original()
            by_src = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["row_indices"],
                      "separate_coo_eids": s["eids"]}
            by_dst = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"],
                      "separate_coo_eids": s["eids"]}
            by_eid = {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["eids"],
                      "separate_coo_eids": s["eids"]}
            # Same values as the reference's op sequence, fewer passes over the [E,H,D] tensors where the shapes
            # allow: the attention terms el / er come out of the projection GEMM's epilogue (dot_ok), and el and the
            # GAT op share one autograd node (fuse_el) so the two gradients of feat_src_per_edge are written by one store
            dot_ok = B.rgnn_relational_matmul_with_attn_dot_ok(self.conv_weights, inputs)
            fuse_el = self.gat_edge_parallel_flag and inputs.is_cuda and B.relational_fused_gat_separate_coo_with_attn_l_ok(
                g, inputs, self.attn_l, self.leaky_relu_slope)
            el = None
            if dot_ok:
                feat_src_per_edge, el = B.rgnn_relational_matmul_with_attn_dot(by_src, self.conv_weights, inputs,
                                                                               self.attn_l, folded=fuse_el)
            else:
                feat_src_per_edge = B.rgnn_relational_matmul(by_src, self.conv_weights, inputs, True, 0)
                if not fuse_el:
                    el = B.rgnn_relational_matmul(by_eid, self.attn_l.unsqueeze(-1), feat_src_per_edge, False, 0)
            if self.multiply_among_weights_first_flag:
                # one input head, [R,H,K,1] weight (the reference passes False here, models.py:310-326,
                # which only works for num_heads == 1)
                er = B.rgnn_relational_matmul(by_dst, self._w_attn_r(), inputs, True, 0)
            elif dot_ok and B.rgnn_relational_matmul_attn_dot_only_ok(by_dst, self.conv_weights, inputs):
                # the per-edge projection by destination feeds nothing but er: it is not materialised
                er = B.rgnn_relational_matmul_attn_dot_only(by_dst, self.conv_weights, inputs, self.attn_r)
            elif dot_ok:
                _, er = B.rgnn_relational_matmul_with_attn_dot(by_dst, self.conv_weights, inputs, self.attn_r)
            else:
                feat_dst_per_edge = B.rgnn_relational_matmul(by_dst, self.conv_weights, inputs, True, 0)
                er = B.rgnn_relational_matmul(by_eid, self.attn_r.unsqueeze(-1), feat_dst_per_edge, False, 0)
            er = er.view(-1, self.num_heads)
            if fuse_el:
                h = B.relational_fused_gat_separate_coo_with_attn_l(g, feat_src_per_edge, self.attn_l, er,
                                                                    self.leaky_relu_slope, el=el)
            elif self.gat_edge_parallel_flag:
                el = el.view(-1, self.num_heads)
                h = B.relational_fused_gat_separate_coo(g, feat_src_per_edge, el, er, self.leaky_relu_slope)
            else:
                h = B.relational_fused_gat_csr(g, feat_src_per_edge, el.view(-1, self.num_heads), er, self.leaky_relu_slope)
        h = h.view(-1, self.out_feat)  # models.py:377-385
        loop_in = inputs
        if num_dst is not None and num_dst < inputs.shape[0]:
            h, loop_in = h[:num_dst], inputs[:num_dst]
        if self.self_loop:
            # the reference calls th.matmul here (models.py:378-379); same product through the segment GEMM
            # with a single segment (MFMA kernel instead of a generic BLAS pick for a 64-wide GEMM)
            key = (loop_in.shape[0], inputs.device)
            if getattr(self, "_loop_offs_key", None) != key:  # built once per (N, device): no per-step host sync
                self._loop_offs = th.tensor([0, loop_in.shape[0]], dtype=th.int64, device=inputs.device)
                self._loop_offs_key = key
            h = h + B.rgnn_relational_matmul_no_scatter_gather_list(
                self._loop_offs, self.loop_weight.view(1, 1, self.in_feat, self.out_feat), loop_in)
        if self.bias:
            h = h + self.h_bias
        if self.activation:
            h = self.activation(h)
        return self.dropout(h)


class HET_EglRelGraphConv_EdgeParallel(nn.Module):
    """RGCN layer on the fused separate-COO op (RGCN/RGCN.py:194-350)."""

    def __init__(self, in_feat, out_feat, num_rels, num_bases=-1, *, bias=True, activation=None,
                 compact_as_of_node_flag=False, compact_direct_indexing_flag=False, dropout=0.0):
        super().__init__()
        self.in_feat, self.out_feat, self.num_rels = in_feat, out_feat, num_rels
        self.num_bases = num_rels if num_bases <= 0 or num_bases > num_rels else num_bases
        self.bias, self.activation = bias, activation
        self.compact_as_of_node_flag = compact_as_of_node_flag
        self.compact_direct_indexing_flag = compact_direct_indexing_flag
        gain = nn.init.calculate_gain("relu")
        self.weight = nn.Parameter(th.Tensor(self.num_bases, in_feat, out_feat))
        nn.init.xavier_uniform_(self.weight, gain=gain)
        if self.num_bases < num_rels:
            self.w_comp = nn.Parameter(th.Tensor(num_rels, self.num_bases))
            nn.init.xavier_uniform_(self.w_comp, gain=gain)
        if bias:
            self.h_bias = nn.Parameter(th.zeros(out_feat))
        self.dropout = nn.Dropout(dropout)

    def forward(self, g, x, norm, num_dst=None):
        if self.num_bases < self.num_rels:  # basis decomposition, RGCN.py:286-301
            weight = th.matmul(self.w_comp, self.weight.view(self.num_bases, -1)).view(
                self.num_rels, self.in_feat, self.out_feat)
        else:
            weight = self.weight
        Xp = next((w for w in (32, 64, 128) if w >= self.out_feat), self.out_feat)
        if PAD_WIDTHS and x.is_cuda and Xp != self.out_feat:
            # Output widths below / between the matrix-core kernels' (the reference's RGCN experiments are 128 | 32 -> 16 | 8
            # classes, hrt/experiments/run_het_rgcn.sh): zero columns appended to the weights change no value -- the extra
            # output columns are zero and dropped, their gradients too -- and the whole layer runs on the 32 / 64 / 128-wide
            # kernels (ogbn-mag, 128 -> 8: 5.15 -> 3.9 ms per step).  HET_PAD_WIDTHS=0: the any-shape kernels as before.
            weight = nn.functional.pad(weight, (0, Xp - self.out_feat))
        Kp = _padded_in_width(self.in_feat) if x.is_cuda else self.in_feat
        if Kp != self.in_feat:  # (input widths outside 32 / 64 / 128: _padded_in_width)
            x, weight = nn.functional.pad(x, (0, Kp - self.in_feat)), nn.functional.pad(weight, (0, 0, 0, Kp - self.in_feat))
        if self.compact_as_of_node_flag:  # RGCN.py:310-336
            ss = g.get_separate_unique_node_indices_single_sided()
            d_row = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_row"],
                     "unique_srcs_and_dests_node_indices": ss["node_indices_row"]}
            feat_compact = B.rgnn_relational_matmul(d_row, weight.unsqueeze(1), x, True, 1)
            node_repr = B.rgcn_node_mean_aggregation_compact_as_of_node_separate_coo_single_sided(
                g, feat_compact.view(feat_compact.shape[0], -1), norm, self.compact_direct_indexing_flag)
        else:
            # (the bias joins the op's output buffer when all rows are kept: backend RgcnLayer1SeparateCooBias)
            bias_in_op = self.bias and (num_dst is None or num_dst >= g.get_num_nodes()) and weight.shape[2] == self.out_feat
            node_repr = B.rgcn_layer1_separate_coo(g, x, weight, norm, self.h_bias if bias_in_op else None)
            if bias_in_op:
                if self.activation:
                    node_repr = self.activation(node_repr)
                return self.dropout(node_repr)
        if num_dst is not None and num_dst < node_repr.shape[0]:
            node_repr = node_repr[:num_dst]
        if node_repr.shape[1] != self.out_feat:  # (padded widths: see above)
            node_repr = node_repr[:, :self.out_feat]
        if self.bias:
            node_repr = node_repr + self.h_bias
        if self.activation:
            node_repr = self.activation(node_repr)
        return self.dropout(node_repr)


def _heads_first(w, H):
    """[R, 1, in, H*dk] typed projection weights -> [R, H, in, dk] (head h = columns h*dk .. (h+1)*dk)."""
    R, _, K, X = w.shape
    return w.view(R, K, H, X // H).permute(0, 2, 1, 3)


class HET_HGTLayerHetero(nn.Module):
    """Heterogeneous graph transformer layer (HGT/models.py:15-286): typed K/Q/V projections per node type,
    per-relation attention and message weights, edge softmax with the relation prior as temperature, typed
    output projection gated by sigmoid(skip).

    ``multiply_among_weights_first_flag`` (HGT/models.py:124-151, in the reference's sweep hrt/utils/_do_all_cases.sh:6,34
    with --num_heads 1): the typed K / Q / V projections are folded into the per-relation weights --
    W_att[r,h] = Q_dt(r),h . att[r,h] . K_st(r),h^T ([in,in]; K . att . Q^T for the fused score op) and
    W_msg[r,h] = V_st(r),h . msg[r,h] ([in,dk]) -- and the edge ops take the layer input itself as k = q = v.  The
    reference's code for it reshapes with ``view`` where a transpose is meant, multiplies relation_msg and V in the
    opposite order and indexes V by the destination type (its own "fixme"s): taken literally it is a different model and
    only type-checks for one head with in_dim == d_k.  Built here with the INTENDED meaning -- the same function as the
    layer without the flag (associativity), any number of heads -- which is what oracle/layers.py::hgt_layer checks."""

    def __init__(self, num_ntypes, num_rels, in_dim, out_dim, num_heads=1, dropout=0.2, use_norm=False,
                 hgt_fused_attn_score_flag=False, compact_as_of_node_flag=False, compact_direct_indexing_flag=False,
                 fused_message_mean_aggregation_flag=True, multiply_among_weights_first_flag=False):
        super().__init__()
        if not fused_message_mean_aggregation_flag:  # (the reference's HGT/models.py:251-277 branch)
            B.hgt_full_graph_edge_softmax_and_message_mean_aggregation_csr(None, None, None, None)  # raises HetUnsupported, named
        assert not use_norm, "use_norm is off in the reference scripts"
        self.num_ntypes, self.num_relations, self.in_dim, self.out_dim = num_ntypes, num_rels, in_dim, out_dim
        self.num_heads, self.d_k = num_heads, out_dim // num_heads
        self.sqrt_dk = math.sqrt(self.d_k)
        self.hgt_fused_attn_score_flag = hgt_fused_attn_score_flag
        self.compact_as_of_node_flag = compact_as_of_node_flag
        self.compact_direct_indexing_flag = compact_direct_indexing_flag
        self.multiply_among_weights_first_flag = multiply_among_weights_first_flag
        self.k_linears = nn.Parameter(th.Tensor(num_ntypes, 1, in_dim, out_dim))
        self.q_linears = nn.Parameter(th.Tensor(num_ntypes, 1, in_dim, out_dim))
        self.v_linears = nn.Parameter(th.Tensor(num_ntypes, 1, in_dim, out_dim))
        self.a_linears = nn.Parameter(th.Tensor(num_ntypes, 1, out_dim, out_dim))
        self.relation_pri = nn.Parameter(th.ones(num_rels, num_heads))
        self.relation_att = nn.Parameter(th.Tensor(num_rels, num_heads, self.d_k, self.d_k))
        self.relation_msg = nn.Parameter(th.Tensor(num_rels, num_heads, self.d_k, self.d_k))
        self.skip = nn.Parameter(th.ones(num_ntypes, 1, 1, 1))
        self.drop = nn.Dropout(dropout)
        self.reset_parameters()

    def reset_parameters(self):
        for p in (self.relation_att, self.relation_msg, self.k_linears, self.q_linears, self.v_linears, self.a_linears):
            nn.init.xavier_uniform_(p)

    def forward(self, G, h, num_dst=None):
        """``num_dst``: ``G`` is a sampled block whose first ``num_dst`` nodes are its destinations (only their rows are
        returned); its nodes are runs of equal type (``node_segment_types``, het_amd/sampling.py) instead of one run per type."""
        offs = G.get_original_node_type_offsets()
        seg_types = G.graph_data["original"].get("node_segment_types") if hasattr(G, "graph_data") else None
        per_run = (lambda w: w) if seg_types is None else (lambda w: w.index_select(0, seg_types))
        if _hgt_fused.hgt_fused_ok(G, h, self.num_heads, self.d_k):
            # attention + aggregation as one node on the distinct (relation, source) rows (backend/hgt_fused_layer.py): the
            # same function of the parameters for every flag combination below, without the per-edge tensors
            k_w, q_w, v_w = self.k_linears, self.q_linears, self.v_linears
            Kp = _padded_in_width(self.in_dim)
            if Kp != self.in_dim:  # (the layer input meets the three typed projections only: zero columns / zero weight rows)
                rows = (0, 0, 0, Kp - self.in_dim)
                h = nn.functional.pad(h, (0, Kp - self.in_dim))
                k_w, q_w, v_w = nn.functional.pad(k_w, rows), nn.functional.pad(q_w, rows), nn.functional.pad(v_w, rows)
            out = _hgt_fused.hgt_layer_fused(G, h, offs, per_run(q_w), per_run(th.sigmoid(self.skip) * self.a_linears),
                                             k_w, v_w, self.relation_att, self.relation_msg,
                                             self.relation_pri, self.num_heads, self.hgt_fused_attn_score_flag)
            return out if num_dst is None else out[:num_dst]
        if self.multiply_among_weights_first_flag:
            return self._forward_weights_first(G, h, offs, per_run, num_dst)
        k = B.rgnn_relational_matmul_no_scatter_gather_list(offs, per_run(self.k_linears), h).view(-1, self.num_heads, self.d_k)
        q = B.rgnn_relational_matmul_no_scatter_gather_list(offs, per_run(self.q_linears), h).view(-1, self.num_heads, self.d_k)
        v = B.rgnn_relational_matmul_no_scatter_gather_list(offs, per_run(self.v_linears), h).view(-1, self.num_heads, self.d_k)
        # The per-edge tensor q[dst] . relation_att[r] of the default flags (models.py:215-241) is read by the inner
        # product only and its rows repeat for every edge of a (relation, destination) pair: when the graph carries the
        # unique (relation, node) lists it is formed on those rows and indexed directly -- the reference's compact
        # dataflow, same values -- instead of being written and re-read as an [E,H,dk] tensor.
        has_lists = "unique_node_indices_single_sided" in getattr(G, "graph_data", {}).get("separate", {})
        as_compact = self.compact_as_of_node_flag or (has_lists and h.is_cuda and B.plan_enabled())
        direct = self.compact_direct_indexing_flag or not self.compact_as_of_node_flag
        if self.hgt_fused_attn_score_flag:  # models.py:172-175
            attn_score = B.hgt_full_graph_hetero_attention_ops_coo(G, self.relation_att, k, q)
        elif as_compact:  # models.py:177-214
            ss = G.get_separate_unique_node_indices_single_sided()
            compact = B.rgnn_relational_matmul(
                {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"],
                 "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}, self.relation_att, q, False, 1)
            attn_score = B.rgnn_inner_product_right_node(G, compact, k, 2 if direct else 1, "_col")
        else:  # models.py:215-241
            s = G.get_separate_coo_original()
            per_edge = B.rgnn_relational_matmul(
                {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"],
                 "separate_coo_eids": s["eids"]}, self.relation_att, q, False, 0)
            attn_score = B.rgnn_inner_product_right_node(G, per_edge, k, 0, "_col")
        new_h = B.hgt_full_graph_message_calc_edge_softmax_and_message_mean_aggregation_coo(
            self.relation_msg, v, G, (self.relation_pri / self.sqrt_dk), attn_score)
        out = B.rgnn_relational_matmul_no_scatter_gather_list(
            offs, per_run(th.sigmoid(self.skip) * self.a_linears), new_h.view(-1, self.out_dim))
        return out if num_dst is None else out[:num_dst]

    def _forward_weights_first(self, G, h, offs, per_run, num_dst):
        """--multiply_among_weights_first_flag (see the class docstring): two small batched products per step replace the three
        typed projections of the nodes; the per-relation [in,in] / [in,dk] weights then meet the layer input directly."""
        H, dk = self.num_heads, self.d_k
        st, dt = G.get_rel_node_types()
        Kr = _heads_first(self.k_linears.index_select(0, st), H)  # [R,H,in,dk]
        Qr = _heads_first(self.q_linears.index_select(0, dt), H)
        Vr = _heads_first(self.v_linears.index_select(0, st), H)
        if self.hgt_fused_attn_score_flag:  # s = < k[src] . att, q[dst] > = h[src] . (K att Q^T) . h[dst]^T
            w_att = th.matmul(th.matmul(Kr, self.relation_att), Qr.transpose(2, 3))
        else:                               # s = < q[dst] . att, k[src] > = h[dst] . (Q att K^T) . h[src]^T
            w_att = th.matmul(th.matmul(Qr, self.relation_att), Kr.transpose(2, 3))
        w_msg = th.matmul(Vr, self.relation_msg)  # [R,H,in,dk]
        hh = h.unsqueeze(1) if H == 1 else h.unsqueeze(1).expand(-1, H, -1)
        hh = hh.contiguous()  # k = q = v: the layer input once per head (the reference: h.unsqueeze(1).repeat(1, H, 1))
        if self.hgt_fused_attn_score_flag:
            attn_score = B.hgt_full_graph_hetero_attention_ops_coo(G, w_att.contiguous(), hh, hh)
        else:
            s = G.get_separate_coo_original()
            per_edge = B.rgnn_relational_matmul(
                {"separate_coo_rel_ptrs": s["rel_ptrs"], "separate_coo_node_indices": s["col_indices"],
                 "separate_coo_eids": s["eids"]}, w_att.contiguous(), h, True, 0)  # [E,H,in] = h[dst] . W_att[r,h]
            attn_score = B.rgnn_inner_product_right_node(G, per_edge, hh, 0, "_col")
        new_h = B.hgt_full_graph_message_calc_edge_softmax_and_message_mean_aggregation_coo(
            w_msg.contiguous(), hh, G, (self.relation_pri / self.sqrt_dk), attn_score)
        out = B.rgnn_relational_matmul_no_scatter_gather_list(
            offs, per_run(th.sigmoid(self.skip) * self.a_linears), new_h.view(-1, self.out_dim))
        return out if num_dst is None else out[:num_dst]
