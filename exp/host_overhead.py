"""Host time per torch_hrt op call: the Python registration (het_amd/kernels.py: torch.library + ctypes) against the compiled
registration object (libtorch_hrt.so, csrc/torch_export.cpp), on inputs the size of a sampled block (kernel time ~ 10 us).
usage: python exp/host_overhead.py            (spawns one interpreter per mode)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    import torch
    sys.path.insert(0, ROOT)
    import het_amd.kernels as k
    K = k.K
    torch.manual_seed(0)
    dev = "cuda"
    N, R, E, H, Kd, D = 2000, 4, 20000, 4, 64, 16
    rp = torch.tensor([0, 5000, 10000, 15000, 20000], device=dev)
    row, col = torch.randint(0, N, (E,), device=dev), torch.randint(0, N, (E,), device=dev)
    eids = torch.arange(E, device=dev)
    d = {"separate_coo_rel_ptrs": rp, "separate_coo_node_indices": row, "separate_coo_eids": eids}
    W, x = torch.randn(R, H, Kd, D, device=dev), torch.randn(N, Kd, device=dev)
    feat = torch.zeros(E, H, D, device=dev)
    el, er = torch.randn(E, H, device=dev), torch.randn(E, H, device=dev)
    s_, ex, ret = torch.empty(N, H, device=dev), torch.empty(E, H, device=dev), torch.empty(N, H, D, device=dev)
    ops = {
        "rgnn_relational_matmul": lambda: K.rgnn_relational_matmul(d, 0, W, x, feat, True),
        "relational_fused_gat_separate_coo": lambda: K.relational_fused_gat_separate_coo(eids, rp, row, col, 0, {}, feat, el, er, s_, ex, ret, 0.2),
    }
    out = {}
    for name, f in ops.items():
        for _ in range(50):
            f()
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        t_issue = (time.perf_counter() - t0) / n * 1e6  # host time to ISSUE a call (the queue stays ahead of the GPU or not)
        torch.cuda.synchronize()
        t_total = (time.perf_counter() - t0) / n * 1e6
        out[name] = (round(t_issue, 1), round(t_total, 1))
    print("compiled" if k.COMPILED_LIB else "python", out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for lib in (None, os.path.join(ROOT, "het_amd", "libtorch_hrt.so")):
            env = dict(os.environ)
            if lib:
                env["HET_TORCH_HRT_LIB"] = lib
            subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env, check=True)
