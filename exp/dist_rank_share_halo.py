"""What one rank of an 8-way partition computes on the layer's OWN exchange path (forward_with_halo: pack, projection of the halo
rows piece by piece, aggregation; backward ordered around the return, unpack), timed on one GPU with the all-to-all replaced by
slicing (dist.LocalRanks): HET_DIST_CHUNKS=1 (monolithic) against the default pieces -- the piecewise projection must not cost the
rank anything.   python3 exp/dist_rank_share_halo.py   (WORLD=8 FEAT=64)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import het_amd.dist as D
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = torch.device("cuda")
world, feat = int(os.environ.get("WORLD", "8")), int(os.environ.get("FEAT", "64"))
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
torch.manual_seed(0)
layer = HET_RGATLayer(feat, feat, coo.num_rels, 4, self_loop=True, dropout=0.0).to(dev)
for chunks in [int(t) for t in os.environ.get("CHUNKS", "1,4").split(",")]:
    D.CHUNKS = chunks
    lr = D.LocalRanks(coo, world, layer)
    x_own = [torch.nn.Parameter(torch.randn(p.n_own, feat, device=dev) * 0.1) for p in lr.plans]
    go = [torch.randn(p.n_own, feat, device=dev) for p in lr.plans]
    for r, p in enumerate(lr.plans):
        lr.wire.push[r] = D._gather_rows(x_own[r].detach(), p.send_idx)
    ms = []
    for r, p in enumerate(lr.plans):
        back = torch.zeros(p.send_idx.numel(), feat, device=dev)

        def step():
            layer.zero_grad(set_to_none=True)
            x_own[r].grad = None
            D._gather_rows(x_own[r].detach(), p.send_idx)  # the pack of this rank's own sends
            out = layer.forward_with_halo(lr.graphs[r], x_own[r], lr.halos[r])
            out.backward(go[r])
            D._scatter_add_rows(x_own[r].grad, p.send_idx, back)  # unpack of the returned rows

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"world {world} feat {feat} pieces {chunks}: per-rank share (layer's exchange path, no wire) min {min(ms):.3f} max {max(ms):.3f} ms "
          f"[{' '.join(f'{m:.2f}' for m in ms)}]; halo rows max {max(p.n_halo for p in lr.plans)}", flush=True)
    del lr, x_own, go
    import het_amd.plan as plan
    plan.clear()
