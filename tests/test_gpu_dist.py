"""The multi-rank path with the real HIP layer: two ranks (gloo rendezvous, both computing on the one GPU of the test
box, halo rows staged through the host) against the single-process layer -- outputs, input and weight gradients.
The 8-GPU runs of the driver use the same plan / halo / gradient code with the nccl (RCCL) backend."""
import os
import socket
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, compact, outdir, feat=64):
    if world > 2:
        torch.set_num_threads(2)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from het_amd.dist import DistRGAT
        from het_amd.synth import make_mag_like
        dev = torch.device("cuda", 0)
        coo = make_mag_like(scale=4e-3)
        flags = dict(compact_as_of_node_flag=compact, compact_direct_indexing_flag=compact)
        runner = DistRGAT(coo, feat, feat, 4, dev, **flags)
        plan = runner.dl.plan
        lo, hi = int(plan.bounds[rank]), int(plan.bounds[rank + 1])
        mine = plan.node_order[lo:hi].cpu()
        gen = torch.Generator().manual_seed(2)
        x_full = torch.randn(coo.num_nodes, feat, generator=gen)
        go_full = torch.randn(coo.num_nodes, feat, generator=gen)
        x_own = x_full[mine].to(dev).requires_grad_(True)
        out = runner.dl.forward(x_own)
        out.backward(go_full[mine].to(dev))
        runner.dl.reduce_param_grads()
        torch.save({"mine": mine, "out": out.detach().cpu(), "gx": x_own.grad.cpu(),
                    "grads": {n: p.grad.cpu() for n, p in runner.layer.named_parameters()}},
                   os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("compact,feat,world", [(False, 64, 2), (True, 64, 2), (False, 128, 2), (True, 128, 4)])
def test_ranks_on_one_gpu_match_single_process(compact, feat, world):
    """feat 128 with 4 heads: the layer shape of BASELINE.json configs[4] (RGAT feat = 128 on a partition).  world 4: as many
    processes as may share the test box's GPU beside this one (the box allows 6); the 8-way partition of configs[4] runs as
    logical ranks of one process in test_eight_way_partition_feat128_on_one_gpu."""
    import torch.multiprocessing as mp
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    with tempfile.TemporaryDirectory() as d:
        mp.start_processes(_worker, args=(world, _free_port(), compact, d, feat), nprocs=world, join=True, start_method="spawn")
        parts = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    dev = torch.device("cuda", 0)
    coo = make_mag_like(scale=4e-3)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    g = HetGraph.from_integrated_coo(coo, full=True)
    torch.manual_seed(0)  # DistRGAT seeds its replicated layer the same way
    layer = HET_RGATLayer(feat, feat, coo.num_rels, 4, self_loop=True, dropout=0.0, compact_as_of_node_flag=compact,
                          compact_direct_indexing_flag=compact).to(dev)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(coo.num_nodes, feat, generator=gen).to(dev).requires_grad_(True)
    go = torch.randn(coo.num_nodes, feat, generator=gen).to(dev)
    ref = layer(g, x)
    ref.backward(go)
    assert torch.equal(torch.sort(torch.cat([q["mine"] for q in parts])).values, torch.arange(coo.num_nodes))
    for q in parts:
        torch.testing.assert_close(q["out"], ref.detach().cpu()[q["mine"]], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(q["gx"], x.grad.cpu()[q["mine"]], rtol=2e-4, atol=2e-5)
        for n, p in layer.named_parameters():
            torch.testing.assert_close(q["grads"][n], p.grad.cpu(), rtol=5e-4, atol=1e-4)


@pytest.mark.parametrize("compact", [False, True])
def test_eight_way_partition_feat128_on_one_gpu(compact):
    """BASELINE.json configs[4] as far as one GPU goes: RGAT, feat 128, 4 heads, the graph split 8 ways by destination range --
    every rank's plan, local graph, halo pack / unpack and the HIP layer's overlapped (forward_with_halo) path run as logical
    ranks of this process (dist.LocalRanks; eight processes may not share the box's GPU), against the single-process HIP layer
    AND the fp64 oracle layer on the whole graph: outputs, input gradients, all weight gradients.  The partition shows the
    cases a rank has to survive: uneven splits, an empty send list between two ranks, a rank that owns no hub destination."""
    from het_amd.dist import LocalRanks
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    from oracle import layers as OL
    from tests.util import assert_close, rgat_min_abs_preactivation
    dev = torch.device("cuda", 0)
    world, feat, H = 8, 128, 4
    coo = make_mag_like(scale=4e-3)
    torch.manual_seed(0)
    layer = HET_RGATLayer(feat, feat, coo.num_rels, H, self_loop=True, dropout=0.0, compact_as_of_node_flag=compact,
                          compact_direct_indexing_flag=compact)
    with torch.no_grad():
        layer.h_bias.uniform_(-0.1, 0.1)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(coo.num_nodes, feat, generator=gen) * 0.5
    go = torch.randn(coo.num_nodes, feat, generator=gen)
    g_cpu = HetGraph.from_integrated_coo(coo, full=True)
    s = g_cpu.get_separate_coo_original()
    for _ in range(64):  # (no pre-activation on the leaky-ReLU kink: tests/util.py)
        if rgat_min_abs_preactivation(x, layer.conv_weights, layer.attn_l, layer.attn_r, s) >= 2e-6:
            break
        x = x + 1e-3 * torch.randn(coo.num_nodes, feat, generator=gen)
    names = ["conv_weights", "attn_l", "attn_r", "loop_weight", "h_bias"]
    p64 = {n: t.detach().double().requires_grad_(True) for n, t in layer.named_parameters()}
    x64 = x.double().requires_grad_(True)
    ref = OL.rgat_layer(x64, p64["conv_weights"], p64["attn_l"], p64["attn_r"], s["rel_ptrs"], s["row_indices"], s["col_indices"],
                        coo.num_nodes, 0.2, p64["loop_weight"], p64["h_bias"])
    grads_ref = torch.autograd.grad(ref, [x64] + [p64[n] for n in names], go.double())

    layer = layer.to(dev)
    dcoo = make_mag_like(scale=4e-3)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(dcoo, f, getattr(dcoo, f).to(dev))
    lr = LocalRanks(dcoo, world, layer, full_layouts=compact)
    plans = lr.plans
    indeg = torch.bincount(coo.col, minlength=coo.num_nodes)
    owned_hubs = [int((indeg[lr.owned_nodes(r).cpu()] > 256).sum()) for r in range(world)]
    assert len({p.n_own for p in plans}) > 1 and len({p.num_local_edges for p in plans}) > 1, "uneven splits expected"
    assert any(p.send_counts[q] == 0 for p in plans for q in range(world) if q != p.rank), "an empty send list expected"
    assert min(owned_hubs) == 0 and max(owned_hubs) > 0, owned_hubs
    assert sum(p.num_local_edges for p in plans) == coo.num_edges and plans[0].edge_cut > 0.8 * coo.num_edges
    mine = [lr.owned_nodes(r).cpu() for r in range(world)]
    x_own = [x[m].to(dev).requires_grad_(True) for m in mine]
    outs = lr.forward(x_own)
    assert all(lr.took_halo_path), lr.took_halo_path  # the layer ran the exchange itself (forward_with_halo) on every rank
    lr.backward(outs, [go[m].to(dev) for m in mine], x_own)
    part_grads = {n: q.grad.detach().cpu().clone() for n, q in layer.named_parameters()}
    for r in range(world):
        assert_close(outs[r], ref.detach()[mine[r]], what=f"rank {r} out vs oracle")
        assert_close(x_own[r].grad, grads_ref[0][mine[r]], what=f"rank {r} grad_x vs oracle")
    for n, gr in zip(names, grads_ref[1:]):
        assert_close(part_grads[n], gr, what="grad_" + n + " (sum over the 8 ranks) vs oracle")
    # the single-process HIP layer on the whole graph
    for q in layer.parameters():
        q.grad = None
    g = HetGraph.from_integrated_coo(dcoo, full=True)
    xd = x.to(dev).requires_grad_(True)
    one = layer(g, xd)
    one.backward(go.to(dev))
    for r in range(world):
        torch.testing.assert_close(outs[r].detach().cpu(), one.detach().cpu()[mine[r]], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(x_own[r].grad.cpu(), xd.grad.cpu()[mine[r]], rtol=2e-4, atol=2e-5)
    for n, q in layer.named_parameters():
        torch.testing.assert_close(part_grads[n], q.grad.cpu(), rtol=5e-4, atol=1e-4)


def _bench_line(cmd, env):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_distributed_flow_one_rank(backend):
    """bench.py's multi-GPU code path (DistRGAT: plan, halo exchange inside the layer node, gradient all-reduce) with a
    process group of one rank on the test box's GPU.  backend = nccl: the RCCL communicator itself -- init, the asynchronous
    all_to_all_single of the halo contexts (empty on one rank), the gradient all-reduce and the timing all-reduce run
    through RCCL; two RCCL ranks cannot share one GPU, so this is as far as a one-GPU box goes."""
    import sys
    env = dict(os.environ, HET_FORCE_DIST="1", HET_DIST_BACKEND=backend, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = _bench_line([sys.executable, "bench.py", "--gpus", "1", "--scale", "0.02", "--steps", "3", "--warmup", "1",
                       "--no-cpu-baseline", "--no-variants"], env)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "strong"


@pytest.mark.parametrize("world,feat", [(2, 64), (4, 128)])
def test_bench_ranks_through_torchrun(world, feat):
    """The driver's launch line for N = 2 / 4 (python -m torch.distributed.run ... bench.py --gpus N), the ranks sharing the one
    GPU of the test box with the gloo backend standing in for RCCL (HET_DIST_BACKEND): rank 0 prints the one JSON line, whose
    `dist` object carries every rank's halo volume and the wait per piece of the exchanges.  (N = 8: tests/test_dist_gloo.py::
    test_bench_dry_run_exchange_eight_ranks -- eight processes may not share the box's GPU.)"""
    import sys
    env = dict(os.environ, HET_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HET_FORCE_DIST"):
        env.pop(k, None)
    out = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                       "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--scale", "0.02", "--feat",
                       str(feat), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env)
    assert out["n_gpus"] == world and out["value"] > 0
    assert "RCCL all-to-all" in out["config"]["parallelism"]
    d = out["dist"]
    assert d["ranks"] == world and len(d["per_rank"]) == world and all(q["halo_rows_sent"] > 0 and q["halo_rows_received"] > 0 for q in d["per_rank"])
    assert sum(q["local_edges"] for q in d["per_rank"]) == 422220  # every edge of the 2 % graph on exactly one rank


def _nccl_worker_script():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "capi", "dist_nccl_worker.py")


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_ranks_match_single_process(world):
    """The real thing where the box has the GPUs: `world` ranks, one GPU each, backend nccl (= RCCL over xGMI), launched the
    driver's way (python -m torch.distributed.run); the asynchronous all_to_all_single of the halo contexts with uneven splits,
    device_id binding and the gradient all-reduce -- outputs, input and weight gradients against the single-process layer.
    Skips on a one-GPU box (two RCCL ranks cannot share a GPU)."""
    import subprocess
    import sys
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HET_FORCE_DIST", "HET_DIST_BACKEND"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                            "127.0.0.1", "--master-port", str(_free_port()), _nccl_worker_script(), d], cwd=root, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        parts = [torch.load(os.path.join(d, f"r{q}.pt")) for q in range(world)]
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    dev = torch.device("cuda", 0)
    feat = 64
    coo = make_mag_like(scale=4e-3)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    g = HetGraph.from_integrated_coo(coo, full=True)
    torch.manual_seed(0)
    layer = HET_RGATLayer(feat, feat, coo.num_rels, 4, self_loop=True, dropout=0.0).to(dev)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(coo.num_nodes, feat, generator=gen).to(dev).requires_grad_(True)
    go = torch.randn(coo.num_nodes, feat, generator=gen).to(dev)
    ref = layer(g, x)
    ref.backward(go)
    assert torch.equal(torch.sort(torch.cat([q["mine"] for q in parts])).values, torch.arange(coo.num_nodes))
    for q in parts:
        assert q["async_exchanges"] > 0, "the halo contexts took the synchronous branch"
        torch.testing.assert_close(q["out"], ref.detach().cpu()[q["mine"]], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(q["gx"], x.grad.cpu()[q["mine"]], rtol=2e-4, atol=2e-5)
        for n, p in layer.named_parameters():
            torch.testing.assert_close(q["grads"][n], p.grad.cpu(), rtol=5e-4, atol=1e-4)


def test_bench_rccl_two_ranks():
    """bench.py --gpus 2 over RCCL through the driver's launch line (needs two GPUs): the JSON line carries the rank count, the
    halo volume of every rank and the exposed exchange time."""
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HET_FORCE_DIST", "HET_DIST_BACKEND"):
        env.pop(k, None)
    out = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                       "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--scale", "0.05", "--steps",
                       "3", "--warmup", "1", "--no-cpu-baseline"], env)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["dist"]["ranks"] == 2 and "RCCL" in out["dist"]["backend"]
    assert len(out["dist"]["per_rank"]) == 2 and all(r["halo_rows_sent"] > 0 for r in out["dist"]["per_rank"])


def test_eight_way_partition_feat128_at_full_size():
    """BASELINE.json configs[4] at its REAL size as far as one GPU goes (round 5): RGAT, feat 128, 4 heads, the whole ogbn-mag-shaped
    graph (21.1 M edges) split 8 ways by destination range, default flags -- every rank's plan, local graph, halo pack / unpack and
    the HIP layer's own exchange path (forward_with_halo, exchanges in pieces) as logical ranks of this process (dist.LocalRanks:
    the all-to-all is slicing), against the single-process HIP layer on the whole graph: outputs, input gradients and every
    parameter gradient.  (The single-process layer at this size is held to the fp64 oracle by
    tests/test_gpu_fullsize.py::test_rgat_feat128_layer_matches_the_fp64_oracle_at_full_size.)  What stays unexecuted of configs[4]
    is RCCL between >= 2 ranks."""
    from het_amd import plan as _plan
    from het_amd.dist import LocalRanks
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    from tests.util import rgat_nudge_off_kink
    dev = torch.device("cuda", 0)
    world, feat, H = 8, 128, 4
    coo = make_mag_like(scale=1.0)
    for f in ("row", "col", "rel", "eids", "node_type_offsets"):
        setattr(coo, f, getattr(coo, f).to(dev))
    N, E = coo.num_nodes, coo.num_edges
    torch.manual_seed(0)
    layer = HET_RGATLayer(feat, feat, coo.num_rels, H, self_loop=True, dropout=0.0).to(dev)
    with torch.no_grad():
        layer.h_bias.uniform_(-0.1, 0.1)
    gen = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(N, feat, device=dev, generator=gen) * 0.3
    go = torch.randn(N, feat, device=dev, generator=gen)
    g = HetGraph.from_integrated_coo(coo, full=False)
    # (no pre-activation within 2e-6 of the leaky-ReLU kink: the two runs round el + er differently, tests/util.py)
    x, zmin = rgat_nudge_off_kink(x, layer.conv_weights, layer.attn_l, layer.attn_r, g.get_separate_coo_original())
    assert zmin >= 2e-6, zmin
    # ---- the single-process layer on the whole graph
    xd = x.clone().requires_grad_(True)
    one = layer(g, xd)
    one.backward(go)
    one = one.detach()
    gx_one = xd.grad
    grads_one = {n: q.grad.detach().clone() for n, q in layer.named_parameters()}
    for q in layer.parameters():
        q.grad = None
    del g, xd
    _plan.clear()
    # ---- eight logical ranks
    lr = LocalRanks(coo, world, layer)
    plans = lr.plans
    assert sum(p.num_local_edges for p in plans) == E and plans[0].edge_cut > 0.8 * E
    mine = [lr.owned_nodes(r) for r in range(world)]
    assert sum(m.numel() for m in mine) == N
    x_own = [x[m].clone().requires_grad_(True) for m in mine]
    outs = lr.forward(x_own)
    assert all(lr.took_halo_path), lr.took_halo_path  # the layer ran the exchange itself on every rank
    lr.backward(outs, [go[m] for m in mine], x_own)

    def close(name, got, want, tol_l2=2e-5, tol_max=1e-3):
        d = got.double() - want.double()
        rel_l2 = float(d.norm() / want.double().norm().clamp(min=1e-30))
        worst = float(d.abs().max() / want.double().abs().max().clamp(min=1e-30))
        print(f"[8-way partition at full size vs one process] {name}: rel L2 {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}")
        assert rel_l2 < tol_l2 and worst < tol_max, f"{name}: relative L2 error {rel_l2:.2e}, max |diff| / max |value| {worst:.2e}"

    out_all, gx_all = torch.empty_like(one), torch.empty_like(gx_one)
    for r in range(world):
        out_all[mine[r]] = outs[r].detach()
        gx_all[mine[r]] = x_own[r].grad
    close("out", out_all, one)
    close("grad_x", gx_all, gx_one)
    for n, q in layer.named_parameters():
        close("grad_" + n + " (sum over the 8 ranks)", q.grad, grads_one[n])
    halo = [p.n_halo for p in plans]
    print(f"[8-way partition at full size] owned nodes {[p.n_own for p in plans]}, halo rows {halo}, local edges {[p.num_local_edges for p in plans]}")
    _plan.clear()
