"""The cross-stream allocation hazard of the node-major HGT backward, provoked: on a graph seen for the first time the backward
builds its node plan (torch kernels on the main stream) AFTER forking the side stream and BEFORE allocating grad_qw, which the
side stream writes.  With the main stream lagging behind the host (a sleep kernel enqueued in front of the backward) the side
stream's kernel runs while the plan's kernels are still pending, and a recycled block can be written by both.
TRIALS cold graphs; every result is compared with the same backward repeated warm and without lag.  HET_HGT_REFORK=0 brings
the hazard back (the shipped code forks again after the allocations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from het_amd.graph import HetGraph
from het_amd.layers import HET_HGTLayerHetero
from het_amd.synth import make_mag_like

dev = th.device("cuda:0")
# the lag goes in the middle of the plan build (after its host reads of the destination lists, which drain the stream): in
# front of the sort that produces the node order
_argsort = th.argsort


def _lagged_argsort(*a, **k):
    if LAG[0]:
        th.cuda._sleep(int(4e7))
    return _argsort(*a, **k)


LAG = [False]
th.argsort = _lagged_argsort
bad = 0
T = int(os.environ.get("TRIALS", "40"))
for trial in range(T):
    coo = make_mag_like(scale=1.0e-3 * (1 + trial % 7), seed=1000 + trial)
    g = HetGraph.from_integrated_coo(coo)
    g.to_(dev)
    th.manual_seed(trial)
    layer = HET_HGTLayerHetero(g.get_num_ntypes(), g.get_num_rels(), 64, 64, num_heads=4, dropout=0.0).to(dev)
    N = g.get_num_nodes()
    h = (th.randn(N, 64, device=dev) * 0.5).requires_grad_(True)
    go = th.randn(N, 64, device=dev)
    junk = [th.empty(int(n), device=dev) for n in (3000, 7000, 16000, 23000, 50000)]  # small free blocks for the allocator to recycle
    del junk
    out = layer(g, h)
    th.cuda.synchronize()
    LAG[0] = True
    out.backward(go)
    LAG[0] = False
    th.cuda.synchronize()
    cold = [h.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    h.grad = None
    for p in layer.parameters():
        p.grad = None
    layer(g, h).backward(go)  # warm: the plan is cached, nothing lags
    th.cuda.synchronize()
    warm = [h.grad] + [p.grad for p in layer.parameters()]
    diff = max(float((a - b).abs().max()) for a, b in zip(cold, warm))
    if diff > 1e-3:
        bad += 1
        print(f"trial {trial}: cold backward differs from the warm one by {diff:.3e}", flush=True)
print(f"{bad} of {T} cold backwards differ (HET_HGT_REFORK={os.environ.get('HET_HGT_REFORK', '1')})")
