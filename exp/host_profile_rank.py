"""Where the HOST time of one rank's step goes (8-way partition of the ogbn-mag-shaped graph, rank 7, the layer's own exchange path
with the all-to-all replaced by slicing): at 1/8 of the graph a step is ~1.3 ms of GPU work behind ~50 launches, so Python time per
launch decides.  cProfile of 300 steps, top functions by own time.   python3 exp/host_profile_rank.py"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import het_amd.dist as D
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = torch.device("cuda")
world, feat, r = 8, 64, int(os.environ.get("RANK_ID", "7"))
coo = make_mag_like(scale=1.0)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
torch.manual_seed(0)
layer = HET_RGATLayer(feat, feat, coo.num_rels, 4, self_loop=True, dropout=0.0).to(dev)
lr = D.LocalRanks(coo, world, layer)
x_own = [torch.nn.Parameter(torch.randn(p.n_own, feat, device=dev) * 0.1) for p in lr.plans]
go = torch.randn(lr.plans[r].n_own, feat, device=dev)
for q, p in enumerate(lr.plans):
    lr.wire.push[q] = D._gather_rows(x_own[q].detach(), p.send_idx)


def step():
    layer.zero_grad(set_to_none=True)
    x_own[r].grad = None
    out = layer.forward_with_halo(lr.graphs[r], x_own[r], lr.halos[r])
    out.backward(go)


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
t_host = (time.perf_counter() - t0) / n * 1e3   # host time to ENQUEUE a step (no sync inside)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n * 1e3
print(f"rank {r} of {world}: host enqueue {t_host:.3f} ms per step, with the GPU drained {t_all:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
