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


@pytest.mark.parametrize("compact,feat", [(False, 64), (True, 64), (False, 128)])
def test_two_ranks_on_one_gpu_match_single_process(compact, feat):
    """feat 128 with 4 heads: the layer shape of BASELINE.json configs[4] (RGAT feat = 128 on a partition)."""
    import torch.multiprocessing as mp
    from het_amd.graph import HetGraph
    from het_amd.layers import HET_RGATLayer
    from het_amd.synth import make_mag_like
    world = 2
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


def test_bench_two_ranks_through_torchrun():
    """The driver's launch line for N = 2 (python -m torch.distributed.run ... bench.py --gpus 2), both ranks on the one GPU
    of the test box with the gloo backend standing in for RCCL (HET_DIST_BACKEND): rank 0 prints the one JSON line."""
    import sys
    env = dict(os.environ, HET_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HET_FORCE_DIST"):
        env.pop(k, None)
    out = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                       "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--scale", "0.02", "--steps",
                       "3", "--warmup", "1", "--no-cpu-baseline"], env)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "RCCL all-to-all" in out["config"]["parallelism"]


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
