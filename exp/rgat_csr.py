import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
coo = make_mag_like(scale=1.0)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
layer = HET_RGATLayer(64, 64, 4, 4, self_loop=True, dropout=0.0, gat_edge_parallel_flag=False).to(dev)
x = torch.nn.Parameter(torch.randn(coo.num_nodes, 64, device=dev) * 0.1)
go = torch.randn(coo.num_nodes, 64, device=dev)
def step():
    x.grad = None
    layer(g, x).backward(go)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("RGAT, CSR GAT op (gat_edge_parallel_flag off):", round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms/step")
