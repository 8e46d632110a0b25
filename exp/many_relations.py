"""Layers on graphs with many relations (the reference sweep's wikikg2 has 535, fb15k 474, am 133, aifb 104): does any
kernel's per-relation tiling fall over?  Random graph of wikikg2's size by default."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer, HET_EglRelGraphConv_EdgeParallel, HET_HGTLayerHetero
from het_amd.synth import make_random

dev = torch.device("cuda")
N, E = int(os.environ.get("N", 2500604)), int(os.environ.get("E", 16109182))
for R in (int(r) for r in os.environ.get("RELS", "4 104 535").split()):
    coo = make_random(N, R, E, seed=1)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    g = HetGraph.from_integrated_coo(coo, full=True)
    x = torch.nn.Parameter(torch.randn(N, 64, device=dev) * 0.1)
    go = torch.randn(N, 64, device=dev)
    norm = torch.rand(E, 1, device=dev)
    for name, layer, extra in (("rgat heads 1", HET_RGATLayer(64, 64, R, 1, self_loop=True, dropout=0.0).to(dev), ()),
                               ("rgat heads 4", HET_RGATLayer(64, 64, R, 4, self_loop=True, dropout=0.0).to(dev), ()),
                               ("rgat compact", HET_RGATLayer(64, 64, R, 4, self_loop=True, dropout=0.0, compact_as_of_node_flag=True,
                                                              compact_direct_indexing_flag=True).to(dev), ()),
                               ("rgcn", HET_EglRelGraphConv_EdgeParallel(64, 64, R).to(dev), (norm,)),
                               ("hgt heads 1", HET_HGTLayerHetero(1, R, 64, 64, num_heads=1, dropout=0.0).to(dev), ()),
                               ("hgt heads 8", HET_HGTLayerHetero(1, R, 64, 64, num_heads=8, dropout=0.0).to(dev), ())):
        def step():
            layer.zero_grad(set_to_none=True)
            x.grad = None
            layer(g, x, *extra).backward(go)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        print(f"R={R:4d} {name:13s} {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms / step", flush=True)
    del g, coo
