"""What one rank of an N-way partition computes, timed on ONE GPU without the exchange: the layer on rank 0's local graph
(owned destinations + halo sources), the pack / unpack kernels around the all-to-all, and the bytes it would move."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd.dist import build_plan
from het_amd.graph import HetGraph
from het_amd.layers import HET_RGATLayer
from het_amd.synth import make_mag_like

dev = torch.device("cuda")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


import het_amd.dist as D
ONLY = os.environ.get("ONLY")  # "world,rank": just that share (for rocprofv3)
BETAS = [float(t) for t in os.environ.get("BETAS", "12").split(",")]
WORLDS = [int(t) for t in os.environ.get("WORLDS", "2,4,8").split(",") if t]  # WORLDS= (empty) with KERNELS=w,r: only that breakdown
for world, beta in ([(1, 0.0)] if 1 in WORLDS or "WORLDS" not in os.environ else []) + [(w, b) for w in WORLDS if w > 1 for b in BETAS]:
    D.NODE_WEIGHT = beta
    rows = []
    for rank in range(world):
        if ONLY and (world, rank) != tuple(int(t) for t in ONLY.split(",")):
            continue
        p = build_plan(coo, rank, world)
        g = HetGraph.from_integrated_coo(p.local, full=False)
        torch.manual_seed(0)
        layer = HET_RGATLayer(64, 64, coo.num_rels, 4, self_loop=True, dropout=0.0).to(dev)
        x_own = torch.nn.Parameter(torch.randn(p.n_own, 64, device=dev) * 0.1)
        halo = torch.randn(p.n_halo, 64, device=dev) * 0.1
        go = torch.randn(p.n_own, 64, device=dev)
        back = torch.zeros(p.send_idx.numel(), 64, device=dev)

        def step():
            layer.zero_grad(set_to_none=True)
            x_own.grad = None
            send = D._gather_rows(x_own, p.send_idx)            # pack (what HaloExchange.forward does around the all-to-all)
            x_local = x_own.new_empty((p.n_own + p.n_halo, 64))
            x_local[: p.n_own].copy_(x_own)
            x_local[p.n_own:].copy_(halo)                         # stands for the receive
            x_local = x_local.detach().requires_grad_(True)
            out = layer(g, x_local, num_dst=p.n_own)
            out.backward(go)
            g_own = x_local.grad[: p.n_own].clone()
            D._scatter_add_rows(g_own, p.send_idx, back)          # unpack of the returned halo gradients
            return send

        ms = timeit(step)
        if os.environ.get("GRAPH") and world > 1:  # the same share replayed as ONE HIP graph: what launch / host overhead costs
            xs = torch.randn(p.n_own + p.n_halo, 64, device=dev).mul_(0.1).requires_grad_(True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    layer.zero_grad(set_to_none=True); xs.grad = None
                    layer(g, xs, num_dst=p.n_own).backward(go)
            torch.cuda.current_stream().wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            layer.zero_grad(set_to_none=True); xs.grad = None
            with torch.cuda.graph(cg):
                layer(g, xs, num_dst=p.n_own).backward(go)
            eager = timeit(lambda: (layer.zero_grad(set_to_none=True), layer(g, xs, num_dst=p.n_own).backward(go)))
            print(f"world {world} rank {rank}: layer fwd+bwd eager {eager:.3f} ms, as one HIP graph {timeit(cg.replay):.3f} ms", flush=True)
        if os.environ.get("PROFILE_RANK") and world == 8 and rank == 7 and beta > 0:
            torch.cuda.synchronize()
            from het_amd import kernels as _k
            _k.event_timers["*"] = []
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            acc = {}
            for a, b, name in _k.event_timers.pop("*"):
                acc[name] = acc.get(name, 0.0) + a.elapsed_time(b) / 5
            print("world 8 rank 7 per C-ABI call (ms):", {k: round(v, 3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}, flush=True)
        rows.append((rank, p.num_local_edges, p.n_own, p.n_halo, int(p.send_idx.numel()), ms))
        del layer, g, p
    if os.environ.get("VERBOSE"):
        for r in rows:
            print(f"world {world} rank {r[0]}: local edges {r[1]:>9} own {r[2]:>8} halo {r[3]:>8} sends {r[4]:>8} rows "
                  f"({r[4] * 256 / 1e6:7.1f} MB out, {r[3] * 256 / 1e6:7.1f} MB in)  compute+pack {r[5]:.2f} ms", flush=True)
    if not rows:
        continue
    ms = [r[5] for r in rows]
    print(f"world {world} node_weight {beta}: compute+pack per rank min {min(ms):.2f} max {max(ms):.2f} ms; "
          f"max rows sent {max(r[4] for r in rows)}, max halo {max(r[3] for r in rows)}; "
          f"=> {coo.num_edges / max(ms) / 1e3:.0f} M edges/s before the exchange", flush=True)

if os.environ.get("KERNELS"):  # per-kernel breakdown of one share: KERNELS="world,rank"
    from het_amd import _lib as HL
    world, rank = (int(t) for t in os.environ["KERNELS"].split(","))
    p = build_plan(coo, rank, world)
    g = HetGraph.from_integrated_coo(p.local, full=False)
    torch.manual_seed(0)
    layer = HET_RGATLayer(64, 64, coo.num_rels, 4, self_loop=True, dropout=0.0).to(dev)
    n_local = p.n_own + p.n_halo
    x = (torch.randn(n_local, 64, device=dev) * 0.1).requires_grad_(True)
    go = torch.randn(p.n_own, 64, device=dev)

    def step():
        layer.zero_grad(set_to_none=True)
        x.grad = None
        layer(g, x, num_dst=p.n_own).backward(go)

    for _ in range(3):
        step()
    HL.kernel_timing(True)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"kernels of rank {rank} of {world} (ms per step):")
    for name in ("HET_rgat_aggregate", "HET_rgat_backward_src_short", "HET_rgat_backward_src_long", "HET_segment_sum", "HET_seg_gemm_mfma<store>",
                 "HET_seg_gemm_mfma<dot>", "HET_seg_gemm_mfma<rmw>", "HET_seg_gemm_mfma<atomic>", "HET_seg_dw_mfma"):
        ms, n = HL.kernel_timing_read(name)
        if n:
            print(f"  {name:32s} {ms / 10:.3f}  ({n / 10:.0f} launches)")
    HL.kernel_timing(False)
