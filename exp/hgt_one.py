import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_HGTLayerHetero
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
coo = make_mag_like(scale=1.0)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
N, K, H = g.get_num_nodes(), 64, 8
x = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.1)
go = torch.randn(N, K, device=dev)
kw = dict(compact_as_of_node_flag=True, compact_direct_indexing_flag=os.environ.get("DIRECT", "1") == "1")
layer = HET_HGTLayerHetero(g.get_num_ntypes(), g.get_num_rels(), K, K, num_heads=H, dropout=0.0, **kw).to(dev)
for _ in range(5):
    x.grad = None
    layer(g, x).backward(go)
torch.cuda.synchronize()
