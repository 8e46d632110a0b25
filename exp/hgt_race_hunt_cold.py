"""Cold-start variant of exp/hgt_race_hunt.py: a NEW graph per trial, its first forward + backward (groupings, node maps, orders are
built inside) against the second and third on the same graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_HGTLayerHetero
from het_amd.synth import make_mag_like
dev = "cuda"
bad_total = 0
for trial in range(int(os.environ.get("TRIALS", "60"))):
    H = (8, 1, 4, 2)[trial % 4]
    g = HetGraph.from_integrated_coo(make_mag_like(scale=1.5e-3, seed=100 + trial % 3))
    g.to_(dev)
    torch.manual_seed(4)
    N, R, T = g.get_num_nodes(), g.get_num_rels(), g.get_num_ntypes()
    layer = HET_HGTLayerHetero(T, R, 64, 64, num_heads=H, dropout=0.0).to(dev)
    h = (torch.randn(N, 64, device=dev) * 0.5).requires_grad_(True)
    go = torch.randn(N, 64, device=dev)
    offs = g.get_original_node_type_offsets().tolist()
    outs = []
    for it in range(3):
        h.grad = None
        layer.zero_grad(set_to_none=True)
        out = layer(g, h)
        out.backward(go)
        outs.append((out.detach().clone(), h.grad.detach().clone(), {n: p.grad.detach().clone() for n, p in layer.named_parameters() if p.grad is not None}))
    torch.cuda.synchronize()
    for it in (0, 1):
        for name, a, b in [("out", outs[it][0], outs[2][0]), ("grad_h", outs[it][1], outs[2][1])] + [(n, outs[it][2][n], outs[2][2][n]) for n in outs[2][2]]:
            d = (a - b).abs()
            if float(d.max()) > 1e-3 * max(1e-6, float(b.abs().max())):
                bad_total += 1
                extra = ""
                if a.dim() == 2 and a.shape[0] == N:
                    bad = torch.nonzero(d.amax(dim=1) > 1e-3 * float(b.abs().max())).flatten()
                    extra = f" rows per type {[int(((bad >= offs[t]) & (bad < offs[t + 1])).sum()) for t in range(T)]} of {[offs[t+1]-offs[t] for t in range(T)]}"
                print(f"trial {trial} (H={H}) run {it} vs run 2: {name} differs, max {float(d.max()):.3g}{extra}", flush=True)
    del g, layer
    if os.environ.get("CLEAR"):
        import het_amd.plan as plan
        plan.clear()
print("differences:", bad_total)
