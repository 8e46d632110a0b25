"""Scratch: capture + replay the RGAT layer step as a HIP graph several times in one process (fork / join events of the library
inside the capture; found a crash in hipStreamEndCapture when those events were destroyed during the capture)."""
import sys, torch
sys.path.insert(0, ".")
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like
DEV = "cuda:0"
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    coo = make_mag_like(scale=2e-3 * (1 + rep % 3), seed=rep)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(DEV))
    g = HetGraph.from_integrated_coo(coo, full=True)
    layer = HET_RGATLayer(64, 64, 4, 4, self_loop=True, dropout=0.0).to(DEV)
    x = torch.nn.Parameter(torch.randn(coo.num_nodes, 64, device=DEV) * 0.1)
    go = torch.randn(coo.num_nodes, 64, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            x.grad = None
            layer(g, x).backward(go)
    torch.cuda.current_stream().wait_stream(side)
    ref = x.grad.clone()
    x.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        layer(g, x).backward(go)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(x.grad, ref, rtol=1e-3, atol=1e-4)
    print("capture", rep, "ok", flush=True)
    del graph
print("GRAPH_CAPTURE_LOOP_OK")
