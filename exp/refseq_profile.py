"""Where the drop-in path spends its time: the reference's op sequence (het_amd/backend/reference_protocol.py = RGAT/models.py:265-385
on reference-named torch_hrt ops only) on the full ogbn-mag shape -- per C-ABI entry point (HIP events) and per torch-side op.
Run under rocprofv3 --kernel-trace --stats for the per-kernel list (profiles/r03/refseq_*)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd import kernels as HK
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = torch.device("cuda")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
torch.manual_seed(0)
VARIANT = os.environ.get("VARIANT", "reference_op_sequence")  # or op_by_op: the reference's model code on het_amd.backend
layer = HET_RGATLayer(64, 64, g.get_num_rels(), 4, self_loop=True, dropout=0.0, reference_op_sequence=VARIANT == "reference_op_sequence").to(dev)
layer.op_by_op = VARIANT == "op_by_op"
x = torch.nn.Parameter(torch.randn(coo.num_nodes, 64, device=dev) * 0.1)
go = torch.randn(coo.num_nodes, 64, device=dev)


def step():
    for p in layer.parameters():
        p.grad = None
    x.grad = None
    layer(g, x).backward(go)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("STEPS", "5"))
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"{VARIANT}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms / step", flush=True)
HK.event_timers["*"] = []
for _ in range(3):
    step()
torch.cuda.synchronize()
acc, cnt = {}, {}
for a, b, name in HK.event_timers.pop("*"):
    acc[name] = acc.get(name, 0.0) + a.elapsed_time(b) / 3
    cnt[name] = cnt.get(name, 0) + 1 / 3
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k[4:]:60s} {v:7.3f} ms  ({cnt[k]:.0f} calls)")
print(f"  (sum of C-ABI calls) {sum(acc.values()):.3f} ms")
if os.environ.get("TORCH_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25))
