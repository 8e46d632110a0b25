"""How much of the dense half of the RGAT backward (node-major input gradient + weight gradients, matrix-core work) could hide
behind the edge pass (gather-bound) if it did not have to wait for it?  UPPER BOUND probe: the dense kernels are launched on
the side stream at the START of the backward on whatever the buffers hold (results are wrong -- timing only), the edge pass runs
on the main stream beside them.  Compared with the shipped order (edge pass, then input gradient || weight gradients).
A chunked pipeline (edge pass of node range c+1 beside the dense work of range c) can at best reach the probe's number."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from het_amd import kernels as _k
from het_amd.backend import rgat_fused_layer as F
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = th.device("cuda:0")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo)
th.manual_seed(0)
layer = HET_RGATLayer(64, 64, g.get_num_rels(), 4, self_loop=True, dropout=0.0).to(dev)
embed = th.nn.Parameter(th.empty(coo.num_nodes, 64, device=dev))
th.nn.init.xavier_uniform_(embed)
go = th.randn(coo.num_nodes, 64, device=dev)


def step():
    for q in layer.parameters():
        q.grad = None
    embed.grad = None
    layer(g, embed).backward(go)


def timeit(tag, n=10):
    for _ in range(3):
        step()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    th.cuda.synchronize()
    print(f"{tag:60s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms / step", flush=True)


shipped = F.RgatLayerFunction._backward_node_major


def probe(mode):
    def bwd(ctx, grad_h, Wt):
        x, W, attn_l, attn_r, loop_w, offs, sm, ex, ret, featc, elc, erc = ctx.saved_tensors
        g, nd, slope = ctx.g, ctx.nd, ctx.slope
        N, Kd = x.shape
        R, H, _, D = W.shape
        X = H * D
        ss = g.get_separate_unique_node_indices_single_sided()
        rp_row = ss["rel_ptrs_row"]
        row_map = _k.node_row_map(rp_row, ss["node_indices_row"], N)
        dst_map = _k.node_row_map(ss["rel_ptrs_col"], ss["node_indices_col"], N)
        gof = grad_h.view(nd, H, D)
        g_featc, g_elc, g_erc = th.zeros_like(featc), th.zeros_like(elc), th.zeros_like(erc)
        grad_bias = th.empty(X, dtype=x.dtype, device=x.device)
        grad_loop = th.empty_like(loop_w)
        main, side = th.cuda.current_stream(x.device), F._side_stream(x.device)
        grad_attn_l = th.empty_like(attn_l)
        wa_t = th.bmm(W.view(-1, Kd, D), attn_r.view(-1, D, 1)).view(R, H, Kd)
        grad_x = th.empty_like(x)
        grad_W, grad_wa = th.empty_like(W), th.empty((R, H, Kd), dtype=x.dtype, device=x.device)
        loop_wt = loop_w.t().contiguous()
        d_col = {"unique_srcs_and_dests_rel_ptrs": ss["rel_ptrs_col"], "unique_srcs_and_dests_node_indices": ss["node_indices_col"]}
        order = _k.node_order_by_presence(row_map, dst_map, split=nd if nd < N else None)

        def dense():
            if "dx" in mode:
                _k.rgat_node_backward_dx(0, N, nd, grad_h, loop_wt, g_featc.view(-1, X), Wt, row_map, g_erc, wa_t, dst_map, grad_x, node_order=order)
            if "dw" in mode:
                _k.rows_matmul_backward_dw(offs, None, x[:nd], grad_h, grad_loop.view(1, 1, Kd, X), accumulate=False)
                _k.rows_matmul_backward_dw(rp_row, ss["node_indices_row"], x, g_featc.view(-1, X), grad_W, accumulate=False)
                _k.matmul_backward(d_col, 1, wa_t.view(R, H, 1, Kd), x, g_erc.view(-1, H, 1), None, grad_wa.view(R, H, Kd, 1), True, accumulate=False)

        def edge():
            _k.rgat_backward_compact(ctx.grp, featc, elc, erc, sm[:nd], ret[:nd], gof, g_featc, g_elc, g_erc, slope, fold_attn_l=attn_l,
                                     row_rel_ptrs=rp_row, grad_bias=grad_bias, bias_rows=nd, runs=ctx.runs,
                                     drow_nodes=ss["node_indices_col"], grad_attn_l=grad_attn_l)
        if mode.startswith("edge_only"):
            edge()
        elif mode.startswith("dense_only"):
            dense()
        elif mode.startswith("beside"):
            side.wait_stream(main)
            with th.cuda.stream(side):
                dense()
            edge()
            main.wait_stream(side)
        elif mode.startswith("serial"):
            edge()
            dense()
        grad_W.addcmul_(grad_wa.unsqueeze(-1), attn_r.view(R, H, 1, D))
        grad_attn_r = (W * grad_wa.unsqueeze(-1)).sum(2)
        return None, None, None, None, None, None, None, grad_x, grad_W, grad_attn_l, grad_attn_r, grad_loop, grad_bias
    return staticmethod(bwd)


timeit("shipped (edge pass, then dx || dW)")
for mode in ("edge_only", "dense_only dx dw", "dense_only dx", "dense_only dw", "serial dx dw", "beside dx dw", "beside dx", "beside dw"):
    F.RgatLayerFunction._backward_node_major = probe(mode)
    timeit(mode)
F.RgatLayerFunction._backward_node_major = shipped
timeit("shipped again")
