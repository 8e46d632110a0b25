"""One RCCL rank of tests/test_gpu_dist.py::test_rccl_ranks_match_single_process (launched by torch.distributed.run: one process
per GPU, backend nccl = RCCL).  usage: dist_nccl_worker.py <output dir>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(outdir):
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    try:
        from het_amd.dist import DistRGAT, HaloContext
        from het_amd.synth import make_mag_like
        feat = 64
        coo = make_mag_like(scale=4e-3)
        runner = DistRGAT(coo, feat, feat, 4, dev)
        # count the exchanges that really went through the asynchronous RCCL branch
        n_async = [0]
        orig = HaloContext._a2a

        def counting(self, recv, send, rc, sc):
            orig(self, recv, send, rc, sc)
            n_async[0] += int(self._work is not None)
        HaloContext._a2a = counting
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
        torch.cuda.synchronize()
        torch.save({"mine": mine, "out": out.detach().cpu(), "gx": x_own.grad.cpu(), "async_exchanges": n_async[0],
                    "grads": {n: p.grad.cpu() for n, p in runner.layer.named_parameters()}}, os.path.join(outdir, f"r{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
