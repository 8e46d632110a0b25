"""RGAT layer step (fwd + bwd) on the ogbn-mag-shaped graph at layer shapes other than the headline's: SHAPES="K:X:H,..."
(default: the reference's experiments/run_het_rgat.sh shape -- 128 -> 8 classes over 8 heads -- and its neighbours)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = th.device("cuda:0")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo)
shapes = os.environ.get("SHAPES", "128:8:8,128:8:1,128:64:8,64:8:8,64:64:8,64:64:4")
for spec in shapes.split(","):
    K, X, H = (int(v) for v in spec.split(":"))
    th.manual_seed(0)
    layer = HET_RGATLayer(K, X, g.get_num_rels(), H, self_loop=True, dropout=0.0).to(dev)
    embed = th.nn.Parameter(th.empty(coo.num_nodes, K, device=dev))
    th.nn.init.xavier_uniform_(embed)
    go = th.randn(coo.num_nodes, X, device=dev)

    def step():
        for q in layer.parameters():
            q.grad = None
        embed.grad = None
        layer(g, embed).backward(go)
    for _ in range(3):
        step()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    th.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"rgat in={K} out={X} heads={H} (D={X // H}): {ms:.2f} ms / step, {coo.num_edges / ms / 1e3:.0f} M edges/s", flush=True)
    del layer, embed, go
    th.cuda.empty_cache()
