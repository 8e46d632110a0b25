"""ms per fwd+bwd layer step over a grid of widths and head counts on the ogbn-mag-shaped graph, all three models, default
flags (MODELS=rgat,rgcn,hgt  KS=..  XS=..  HS=..): finds shapes that fall off the row kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer, HET_EglRelGraphConv_EdgeParallel, HET_HGTLayerHetero
from het_amd.synth import make_mag_like

dev = th.device("cuda:0")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
N, E, R = g.get_num_nodes(), g.get_num_edges(), g.get_num_rels()
ints = lambda name, d: [int(v) for v in os.environ.get(name, d).split(",")]
KS, XS, HS = ints("KS", "16,32,64,100,128,256"), ints("XS", "8,16,32,48,64,128,256"), ints("HS", "1,2,4,8")
norm = th.rand(E, 1, device=dev)
for model in os.environ.get("MODELS", "rgat,rgcn,hgt").split(","):
    for K in KS:
        x = th.nn.Parameter(th.randn(N, K, device=dev) * 0.1)
        for X in XS:
            go = th.randn(N, X, device=dev)
            for H in (HS if model != "rgcn" else [1]):
                if X % H:
                    continue
                th.manual_seed(0)
                extra = ()
                try:
                    if model == "rgat":
                        layer = HET_RGATLayer(K, X, R, H, self_loop=True, dropout=0.0).to(dev)
                    elif model == "rgcn":
                        layer, extra = HET_EglRelGraphConv_EdgeParallel(K, X, R).to(dev), (norm,)
                    else:
                        layer = HET_HGTLayerHetero(g.get_num_ntypes(), R, K, X, num_heads=H, dropout=0.0).to(dev)

                    def step():
                        x.grad = None
                        for p in layer.parameters():
                            p.grad = None
                        layer(g, x, *extra).backward(go)
                    step()
                    th.cuda.synchronize()
                    t0 = time.perf_counter()
                    step()
                    th.cuda.synchronize()
                    first = time.perf_counter() - t0
                    n = 3 if first < 0.05 else 1
                    t0 = time.perf_counter()
                    for _ in range(n):
                        step()
                    th.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / n * 1e3
                    print(f"{model} in={K:3d} out={X:3d} heads={H}: {ms:8.2f} ms", flush=True)
                except Exception as ex:  # noqa: BLE001
                    print(f"{model} in={K:3d} out={X:3d} heads={H}: FAILED {type(ex).__name__}: {str(ex)[:120]}", flush=True)
                del layer
                th.cuda.empty_cache()
