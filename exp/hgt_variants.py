"""Time the HGT layer flag combinations on the full mag-shaped graph (one-off experiment)."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_HGTLayerHetero
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
N, K, H = g.get_num_nodes(), 64, int(os.environ.get("HEADS", "8"))
x = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.1)
go = torch.randn(N, K, device=dev)
for name, kw in [("default", {}), ("fused_attn", dict(hgt_fused_attn_score_flag=True)),
                 ("compact", dict(compact_as_of_node_flag=True)), ("compact_direct", dict(compact_as_of_node_flag=True, compact_direct_indexing_flag=True))]:
    torch.manual_seed(0)
    layer = HET_HGTLayerHetero(g.get_num_ntypes(), g.get_num_rels(), K, K, num_heads=H, dropout=0.0, **kw).to(dev)
    def step():
        x.grad = None
        for p in layer.parameters():
            p.grad = None
        layer(g, x).backward(go)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms/step", flush=True)
    del layer
