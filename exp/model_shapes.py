"""RGCN / HGT layer step (fwd + bwd) on the ogbn-mag-shaped graph at the widths of the reference's experiment scripts
(hrt/experiments/run_het_rgcn.sh: 128 | 32 -> 16 | 8) and CLI defaults: SHAPES="model:K:X:H,...", COMPACT=1 for the compact flags."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from het_amd.graph import HetGraph
from het_amd.layers import HET_EglRelGraphConv_EdgeParallel, HET_HGTLayerHetero
from het_amd.synth import make_mag_like

dev = th.device("cuda:0")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
compact = os.environ.get("COMPACT", "0") == "1"
g = HetGraph.from_integrated_coo(coo, full=True)
E, N = coo.num_edges, coo.num_nodes
shapes = os.environ.get("SHAPES", "rgcn:128:16:1,rgcn:128:8:1,rgcn:32:16:1,rgcn:32:8:1,rgcn:64:64:1,rgcn:128:32:1,hgt:64:8:1,hgt:128:8:1,hgt:128:8:8,hgt:64:64:8")
for spec in shapes.split(","):
    m, K, X, H = spec.split(":")
    K, X, H = int(K), int(X), int(H)
    th.manual_seed(0)
    extra = ()
    if m == "rgcn":
        layer = HET_EglRelGraphConv_EdgeParallel(K, X, g.get_num_rels(), compact_as_of_node_flag=compact, compact_direct_indexing_flag=compact).to(dev)
        extra = (th.rand(E, 1, device=dev),)
    else:
        layer = HET_HGTLayerHetero(g.get_num_ntypes(), g.get_num_rels(), K, X, num_heads=H, dropout=0.0).to(dev)
    embed = th.nn.Parameter(th.empty(N, K, device=dev))
    th.nn.init.xavier_uniform_(embed)
    go = th.randn(N, X, device=dev)

    def step():
        for q in layer.parameters():
            q.grad = None
        embed.grad = None
        layer(g, embed, *extra).backward(go)
    for _ in range(3):
        step()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    th.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{m} in={K} out={X} heads={H} compact={int(compact)}: {ms:.2f} ms / step, {E / ms / 1e3:.0f} M edges/s", flush=True)
    del layer, embed, go, extra
    th.cuda.empty_cache()
