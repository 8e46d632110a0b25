"""Multi-rank path on CPU (gloo, world_size 2 and 3): destination-range partition + halo all-to-all +
gradient reduction, with the CPU oracle as the per-rank layer, against the single-process oracle."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from het_amd.synth import make_mag_like, make_random


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _coo(kind):
    return make_mag_like(scale=5e-4) if kind.startswith("mag") else make_random(97, 3, 800, seed=4)


def _params(R, H, K, D):
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: (torch.randn(*s, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    return dict(W=mk(R, H, K, D), al=mk(R, H, D), ar=mk(R, H, D), lw=mk(K, H * D), b=mk(H * D))


_DIMS = {"mag": (2, 8, 4), "random": (2, 8, 4), "mag128": (4, 128, 32)}  # (heads, in feat, per-head out); mag128 = configs[4]


class _HaloOracleLayer(torch.autograd.Function):
    """A layer that runs the halo exchange itself through dist.HaloContext (as the HIP one-node RGAT layer does on a
    partition): start_push / finish_push around the forward, start_return / finish_return around the backward."""

    @staticmethod
    def forward(ctx, x_own, halo, fn, n_own, *params):
        x_local = halo.start_push(x_own.detach())
        halo.finish_push()
        with torch.enable_grad():
            xl = x_local.requires_grad_(True)
            out = fn(xl)[:n_own]
        ctx.halo, ctx.xl, ctx.out, ctx.params, ctx.n_own = halo, xl, out, params, n_own
        return out.detach()

    @staticmethod
    def backward(ctx, grad_out):
        grads = torch.autograd.grad(ctx.out, (ctx.xl,) + tuple(ctx.params), grad_out)
        ctx.halo.start_return(grads[0])
        g_own = ctx.halo.finish_return(grads[0][: ctx.n_own].clone())
        return (g_own, None, None, None) + tuple(grads[1:])


def _worker(rank, world, port, kind, outdir, use_halo=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 4:
        torch.set_num_threads(1)  # (8 ranks on the 8 cores of the build container)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from het_amd.dist import DistLayer
        from oracle import layers as OL
        coo = _coo(kind)
        H, K, D = _DIMS[kind]
        p = _params(coo.num_rels, H, K, D)

        def layer_fn(g, x, n_own):
            s = g.get_separate_coo_original()
            return OL.rgat_layer(x, p["W"], p["al"], p["ar"], s["rel_ptrs"], s["row_indices"], s["col_indices"],
                                 g.get_num_nodes(), 0.2, p["lw"], p["b"])

        def halo_layer_fn(g, x_own, halo):
            return _HaloOracleLayer.apply(x_own, halo, lambda xl: layer_fn(g, xl, x_own.shape[0]), x_own.shape[0], *p.values())

        dl = DistLayer(coo, layer_fn, p.values(), halo_layer_fn=halo_layer_fn if use_halo else None)
        lo, hi = int(dl.plan.bounds[rank]), int(dl.plan.bounds[rank + 1])
        gen = torch.Generator().manual_seed(2)
        x_full = torch.randn(coo.num_nodes, K, generator=gen, dtype=torch.float64)
        go_full = torch.randn(coo.num_nodes, H * D, generator=gen, dtype=torch.float64)
        mine = dl.plan.node_order[lo:hi]  # original ids of the nodes this rank owns
        x_own = x_full[mine].clone().requires_grad_(True)
        out = dl.forward(x_own)
        out.backward(go_full[mine])
        dl.reduce_param_grads()
        torch.save({"lo": lo, "hi": hi, "mine": mine, "sent": sum(dl.plan.send_counts), "out": out.detach(), "gx": x_own.grad, "gW": p["W"].grad, "gal": p["al"].grad,
                    "glw": p["lw"].grad, "n_halo": dl.plan.n_halo, "edges": dl.plan.num_local_edges,
                    "cut": dl.plan.edge_cut}, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,use_halo", [("mag", 2, False), ("random", 3, False), ("mag128", 2, False), ("mag", 2, True),
                                                 ("random", 3, True), ("mag128", 8, True), ("random", 8, False)])
def test_partitioned_layer_matches_single_process(kind, world, use_halo, monkeypatch):
    """mag128: feat 128, 4 heads -- the shape of BASELINE.json configs[4] (RGAT feat = 128 on a partition; world 8 = its rank
    count, eight gloo processes).  use_halo: the layer drives the exchange through dist.HaloContext (the overlapped form of the
    HIP layer) instead of HaloExchange."""
    from het_amd.graph import HetGraph
    from oracle import layers as OL
    # (the workers are fresh interpreters: the pieces of the exchange come from the environment -- 1 = monolithic, 4 = default)
    monkeypatch.setenv("HET_DIST_CHUNKS", "1" if (kind, world) == ("mag", 2) else ("3" if world == 3 else "4"))
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), kind, d, use_halo), nprocs=world, join=True)
        parts = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    coo = _coo(kind)
    H, K, D = _DIMS[kind]
    p = _params(coo.num_rels, H, K, D)
    g = HetGraph.from_integrated_coo(coo, full=False)
    s = g.get_separate_coo_original()
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(coo.num_nodes, K, generator=gen, dtype=torch.float64).requires_grad_(True)
    go = torch.randn(coo.num_nodes, H * D, generator=gen, dtype=torch.float64)
    ref = OL.rgat_layer(x, p["W"], p["al"], p["ar"], s["rel_ptrs"], s["row_indices"], s["col_indices"], coo.num_nodes,
                        0.2, p["lw"], p["b"])
    ref.backward(go)
    assert sum(q["edges"] for q in parts) == coo.num_edges  # every edge lives on exactly one rank
    if world == 8:
        assert sum(q["sent"] for q in parts) == sum(q["n_halo"] for q in parts) > 0  # every halo row has one sender
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == coo.num_nodes
    for a, b in zip(parts[:-1], parts[1:]):
        assert a["hi"] == b["lo"]
    assert torch.equal(torch.sort(torch.cat([q["mine"] for q in parts])).values, torch.arange(coo.num_nodes))
    for q in parts:
        torch.testing.assert_close(q["out"], ref.detach()[q["mine"]])
        torch.testing.assert_close(q["gx"], x.grad[q["mine"]])
        # after the all-reduce every rank holds the full weight gradients
        torch.testing.assert_close(q["gW"], p["W"].grad)
        torch.testing.assert_close(q["gal"], p["al"].grad)
        torch.testing.assert_close(q["glw"], p["lw"].grad)


def test_partition_bounds_balance_cost():
    """Destination ranges carry ~equal cost = in-edges + node_weight per destination (node_weight 0: in-edges alone)."""
    from het_amd.dist import partition_bounds
    coo = make_mag_like(scale=2e-3)
    indeg = torch.bincount(coo.col, minlength=coo.num_nodes).double()
    hub = indeg.max().item()
    for node_weight in (0.0, 12.0):
        cost = indeg + node_weight * (indeg > 0)
        for world in (2, 4, 8):
            b = partition_bounds(coo.col, coo.num_nodes, world, node_weight)
            assert b[0] == 0 and b[-1] == coo.num_nodes and bool((b[1:] >= b[:-1]).all())
            owner = torch.searchsorted(b[1:].contiguous(), torch.arange(coo.num_nodes), right=True).clamp(max=world - 1)
            per_rank = torch.zeros(world, dtype=torch.float64).index_add_(0, owner, cost)
            # a single hub destination can exceed the ideal share; otherwise within 25 %
            assert float(per_rank.max()) <= float(cost.sum()) / world * 1.25 + hub + node_weight


def test_source_only_nodes_are_dealt_out_evenly():
    """ogbn-mag's authors have no in-edges: their ownership (hence the rows a rank must send) is balanced."""
    from het_amd.dist import build_plan
    coo = make_mag_like(scale=5e-3)
    world = 4
    plans = [build_plan(coo, r, world) for r in range(world)]
    sent = torch.tensor([float(sum(p.send_counts)) for p in plans])
    assert float(sent.max()) <= 1.6 * float(sent.mean())
    assert sum(p.num_local_edges for p in plans) == coo.num_edges
    order = plans[0].node_order
    assert torch.equal(torch.sort(order).values, torch.arange(coo.num_nodes))


class _OracleRGAT:
    """The oracle layer behind the two entry points LocalRanks calls on a layer (HET_RGATLayer's forward / forward_with_halo)."""

    def __init__(self, p, with_halo):
        self.p, self.with_halo = p, with_halo

    def _fn(self, g, x):
        p, s = self.p, g.get_separate_coo_original()
        from oracle import layers as OL
        return OL.rgat_layer(x, p["W"], p["al"], p["ar"], s["rel_ptrs"], s["row_indices"], s["col_indices"], g.get_num_nodes(), 0.2,
                             p["lw"], p["b"])

    def __call__(self, g, x, num_dst=None):
        return self._fn(g, x)

    def forward_with_halo(self, g, x_own, halo):
        if not self.with_halo:
            return None
        return _HaloOracleLayer.apply(x_own, halo, lambda xl: self._fn(g, xl), x_own.shape[0], *self.p.values())


@pytest.mark.parametrize("chunks", [1, 3, 50])
def test_exchange_pieces_cover_every_row_once(chunks, monkeypatch):
    """The halo exchange in pieces (DistPlan.chunks): sender and receiver cut every peer's block at the same places, pieces are
    contiguous ranges of the halo rows / of send_idx, and together they move every row exactly once -- also with more pieces
    than a peer has rows (empty pieces)."""
    import het_amd.dist as D
    coo = make_mag_like(scale=5e-4)
    world = 4
    plans = [D.build_plan(coo, r, world, chunks=chunks) for r in range(world)]
    base = [D.build_plan(coo, r, world, chunks=1) for r in range(world)]
    for p, b in zip(plans, base):
        assert p.chunks == chunks and p.n_halo == b.n_halo and p.send_counts == b.send_counts and p.recv_counts == b.recv_counts
        assert torch.equal(torch.sort(p.halo_global).values, b.halo_global) and torch.equal(torch.sort(p.send_idx).values, torch.sort(b.send_idx).values)
        assert p.halo_chunk_ptr[0] == 0 and p.halo_chunk_ptr[-1] == p.n_halo and p.send_chunk_ptr[-1] == p.send_idx.numel()
        assert [sum(p.recv_splits[c][q] for c in range(chunks)) for q in range(world)] == p.recv_counts
        for c in range(chunks):
            for q in range(world):
                assert p.recv_splits[c][q] == plans[q].send_splits[c][p.rank]
        # same local graph up to the numbering of the halo nodes
        assert p.num_local_edges == b.num_local_edges and torch.equal(p.local.col, b.local.col)
        glob = lambda pl: torch.where(pl.local.row < pl.n_own, pl.local.row + int(pl.bounds[pl.rank]),
                                      pl.halo_global[(pl.local.row - pl.n_own).clamp(min=0)])
        assert torch.equal(glob(p), glob(b))


@pytest.mark.parametrize("world,with_halo,chunks", [(8, True, None), (8, False, None), (5, True, 1), (1, True, None), (3, True, 7)])
def test_local_ranks_rehearsal_matches_single_process(world, with_halo, chunks, monkeypatch):
    """dist.LocalRanks (every rank of the partition as a logical rank of one process, the all-to-all by slicing) with the oracle
    as the layer: outputs, input gradients and the accumulated weight gradients equal the single-process oracle -- the harness
    tests/test_gpu_dist.py runs the HIP layer through at 8 ranks."""
    import het_amd.dist as hd
    from het_amd.dist import LocalRanks
    from het_amd.graph import HetGraph
    from oracle import layers as OL
    if chunks is not None:
        monkeypatch.setattr(hd, "CHUNKS", chunks)
    coo = make_mag_like(scale=1e-3)
    H, K, D = 2, 8, 4
    p = _params(coo.num_rels, H, K, D)
    lr = LocalRanks(coo, world, _OracleRGAT(p, with_halo))
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(coo.num_nodes, K, generator=gen, dtype=torch.float64)
    go = torch.randn(coo.num_nodes, H * D, generator=gen, dtype=torch.float64)
    mine = [lr.owned_nodes(r) for r in range(world)]
    assert torch.equal(torch.sort(torch.cat(mine)).values, torch.arange(coo.num_nodes))
    x_own = [x[m].clone().requires_grad_(True) for m in mine]
    outs = lr.forward(x_own)
    assert lr.took_halo_path == [with_halo] * world
    lr.backward(outs, [go[m] for m in mine], x_own)
    got = {k: v.grad.clone() for k, v in p.items()}
    for v in p.values():
        v.grad = None
    g = HetGraph.from_integrated_coo(coo, full=False)
    s = g.get_separate_coo_original()
    xr = x.clone().requires_grad_(True)
    ref = OL.rgat_layer(xr, p["W"], p["al"], p["ar"], s["rel_ptrs"], s["row_indices"], s["col_indices"], coo.num_nodes, 0.2, p["lw"], p["b"])
    ref.backward(go)
    for r in range(world):
        torch.testing.assert_close(outs[r].detach(), ref.detach()[mine[r]])
        torch.testing.assert_close(x_own[r].grad, xr.grad[mine[r]])
    for k, v in p.items():
        torch.testing.assert_close(got[k], v.grad)


def test_rccl_preflight_logic_on_gloo():
    """rccl_preflight (one float per peer + barrier under a watchdog) on a one-rank gloo group: passes; the watchdog is what a
    dead peer meets (exercised over RCCL by tests/test_gpu_dist.py where the box has the GPUs)."""
    from het_amd.dist import rccl_preflight
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rccl_preflight(None, torch.device("cpu"), timeout_s=30)
    finally:
        dist.destroy_process_group()


def test_bench_dry_run_exchange_eight_ranks():
    """The driver's launch line for N = 8 (python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8) in
    bench.py's --dry-run-exchange mode: rendezvous, the 8-way plan of every rank and one halo exchange each way on host tensors
    (no GPU: a one-GPU box may not hold eight processes on its card), every row verified against the node it belongs to."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HET_FORCE_DIST", "HET_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--scale", "0.01", "--feat", "128",
                        "--dry-run-exchange"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = out["dist"]
    assert out["n_gpus"] == 8 and out["value"] is None and d["ranks"] == 8 and len(d["per_rank"]) == 8
    assert d["all_rows_verified"] and all(q["halo_rows_received"] > 0 and q["halo_rows_sent"] > 0 for q in d["per_rank"])
    assert sum(q["halo_rows_sent"] for q in d["per_rank"]) == sum(q["halo_rows_received"] for q in d["per_rank"])
    assert any(c == 0 for q in d["per_rank"] for i, c in enumerate(q["send_counts"]) if i != q["rank"])  # an empty send list occurs
    assert d["edges_total"] == 211111
