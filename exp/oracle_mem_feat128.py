"""Peak GPU memory of the fp64 oracle RGAT layer (fwd + bwd) on the full ogbn-mag-shaped graph at feat 128, heads 4."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.synth import make_mag_like
from oracle import layers as OL
DEV = "cuda"
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(DEV))
g = HetGraph.from_integrated_coo(coo, full=False)
s = g.get_separate_coo_original()
N, R, H, K, X = g.get_num_nodes(), 4, 4, 128, 128
gen = torch.Generator(device=DEV).manual_seed(1)
mk = lambda *sh: (torch.randn(*sh, device=DEV, generator=gen, dtype=torch.float64) * 0.2).requires_grad_(True)
x, W, al, ar, lw, b = mk(N, K), mk(R, H, K, X // H), mk(R, H, X // H), mk(R, H, X // H), mk(K, X), mk(X)
go = torch.randn(N, X, device=DEV, generator=gen, dtype=torch.float64)
torch.cuda.reset_peak_memory_stats()
out = OL.rgat_layer(x, W, al, ar, s["rel_ptrs"], s["row_indices"], s["col_indices"], N, 0.2, lw, b)
print("after forward: peak GB", torch.cuda.max_memory_allocated() / 2**30, flush=True)
out.backward(go)
torch.cuda.synchronize()
print("after backward: peak GB", torch.cuda.max_memory_allocated() / 2**30, "total GB", torch.cuda.mem_get_info()[1] / 2**30, flush=True)
