"""ms per fwd+bwd step of every layer flag combination on the full mag-shaped graph (catches slow fallbacks)."""
import itertools, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer, HET_EglRelGraphConv_EdgeParallel, HET_HGTLayerHetero
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
x = torch.nn.Parameter(torch.randn(N, 64, device=dev) * 0.1)
go = torch.randn(N, 64, device=dev)
norm = torch.rand(E, 1, device=dev)
def timeit(layer, extra=()):
    def step():
        x.grad = None
        for p in layer.parameters(): p.grad = None
        layer(g, x, *extra).backward(go)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / 5 * 1e3, 2)
for compact, direct, mulfirst, ep in itertools.product([False, True], [False, True], [False, True], [True, False]):
    if direct and not compact: continue
    if compact and not ep: continue
    l = HET_RGATLayer(64, 64, R, 4, self_loop=True, dropout=0.0, compact_as_of_node_flag=compact, compact_direct_indexing_flag=direct,
                      multiply_among_weights_first_flag=mulfirst, gat_edge_parallel_flag=ep).to(dev)
    print(f"RGAT compact={compact} direct={direct} mulfirst={mulfirst} edge_parallel={ep}: {timeit(l)} ms", flush=True)
for compact, direct in [(False, False), (True, False), (True, True)]:
    l = HET_EglRelGraphConv_EdgeParallel(64, 64, R, compact_as_of_node_flag=compact, compact_direct_indexing_flag=direct).to(dev)
    print(f"RGCN compact={compact} direct={direct}: {timeit(l, (norm,))} ms", flush=True)
l = HET_EglRelGraphConv_EdgeParallel(64, 64, R, num_bases=2).to(dev)
print(f"RGCN num_bases=2: {timeit(l, (norm,))} ms", flush=True)
for kw in [{}, dict(hgt_fused_attn_score_flag=True), dict(compact_as_of_node_flag=True), dict(compact_as_of_node_flag=True, compact_direct_indexing_flag=True)]:
    l = HET_HGTLayerHetero(g.get_num_ntypes(), R, 64, 64, num_heads=8, dropout=0.0, **kw).to(dev)
    print(f"HGT {kw}: {timeit(l)} ms", flush=True)
