"""How much of a small step is host time?  CPU time to enqueue one RGAT step (no sync) vs the GPU time of the step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
for scale in (1.0, 0.125, 0.02):
    coo = make_mag_like(scale=scale)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    g = HetGraph.from_integrated_coo(coo, full=False)
    layer = HET_RGATLayer(64, 64, 4, 4, self_loop=True, dropout=0.0).to(dev)
    x = torch.nn.Parameter(torch.randn(coo.num_nodes, 64, device=dev) * 0.1)
    go = torch.randn(coo.num_nodes, 64, device=dev)
    params = [x] + list(layer.parameters())
    def step():
        for p in params:
            p.grad = None
        layer(g, x).backward(go)
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_enq = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e3
    print(f"scale {scale}: enqueue {t_enq:.3f} ms/step (host), total {t_all:.3f} ms/step", flush=True)
