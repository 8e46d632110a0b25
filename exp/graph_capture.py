"""Can a whole layer step (forward + backward through the C-ABI kernels) be captured in a HIP graph?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
scale = float(os.environ.get("SCALE", "0.01"))
coo = make_mag_like(scale=scale)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
layer = HET_RGATLayer(64, 64, 4, 4, self_loop=True, dropout=0.0).to(dev)
x = torch.nn.Parameter(torch.randn(coo.num_nodes, 64, device=dev) * 0.1)
go = torch.randn(coo.num_nodes, 64, device=dev)
params = [x] + list(layer.parameters())
def step():
    for p in params:
        p.grad = None
    layer(g, x).backward(go)
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", round(timeit(step), 4), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
ref = [p.grad.clone() for p in params]
gr = torch.cuda.CUDAGraph()
for p in params: p.grad = None
with torch.cuda.graph(gr):
    out = layer(g, x)
    out.backward(go)
gr.replay(); torch.cuda.synchronize()
err = max(float((p.grad - r).abs().max()) for p, r in zip(params, ref))
print("captured; max |grad diff| vs eager:", err, flush=True)
print("graph ms/step", round(timeit(gr.replay), 4), flush=True)
