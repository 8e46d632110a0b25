import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_random
dev = torch.device("cuda")
N, E, R = 2500604, 16109182, int(sys.argv[1])
coo = make_random(N, R, E, seed=1)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
x = torch.nn.Parameter(torch.randn(N, 64, device=dev) * 0.1)
go = torch.randn(N, 64, device=dev)
layer = HET_RGATLayer(64, 64, R, 4, self_loop=True, dropout=0.0).to(dev)
for _ in range(8):
    layer.zero_grad(set_to_none=True); x.grad = None
    layer(g, x).backward(go)
torch.cuda.synchronize()
ss = g.get_separate_unique_node_indices_single_sided()
print("S_row", ss["node_indices_row"].numel(), "S_col", ss["node_indices_col"].numel())
