"""Per-kernel table (torch profiler) of one RGAT layer step at SHAPE=K:X:H on the ogbn-mag-shaped graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from torch.profiler import profile, ProfilerActivity
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = th.device("cuda:0")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo)
K, X, H = (int(v) for v in os.environ.get("SHAPE", "128:8:8").split(":"))
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    th.cuda.synchronize()
print(f"shape in={K} out={X} heads={H}; 3 steps")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
