"""Stress the fused HGT layer's backward: same inputs many times, grad_h of every iteration against the first (differences beyond
float-atomic noise = a race); reports which node types differ.  env: ITERS, HET_RGAT_OVERLAP=0 (no torch side stream), HET_HGT_NODE_DX=0"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_HGTLayerHetero
from het_amd.synth import make_mag_like
dev = "cuda"
H = int(os.environ.get("HEADS", "4"))
bad_total = 0
for trial in range(int(os.environ.get("TRIALS", "6"))):
    g = HetGraph.from_integrated_coo(make_mag_like(scale=1.5e-3))
    g.to_(dev)
    torch.manual_seed(4)
    N, R, T = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes()
    layer = HET_HGTLayerHetero(T, R, 64, 64, num_heads=H, dropout=0.0).to(dev)
    h = (torch.randn(N, 64, device=dev) * 0.5).requires_grad_(True)
    go = torch.randn(N, 64, device=dev)
    offs = g.get_original_node_type_offsets().tolist()
    ref = None
    for it in range(int(os.environ.get("ITERS", "40"))):
        h.grad = None
        layer.zero_grad(set_to_none=True)
        layer(g, h).backward(go)
        gh = h.grad.detach().clone()
        if ref is None:
            ref = gh
            continue
        d = (gh - ref).abs().amax(dim=1)
        bad = torch.nonzero(d > 1e-3 * float(ref.abs().max())).flatten()
        if bad.numel():
            bad_total += 1
            per_type = [int(((bad >= offs[t]) & (bad < offs[t + 1])).sum()) for t in range(T)]
            print(f"trial {trial} iter {it}: {bad.numel()} rows differ (per node type {per_type} of {[offs[t+1]-offs[t] for t in range(T)]}), max {float(d.max()):.3g}", flush=True)
    del g, layer
print("iterations with differing rows:", bad_total)
