"""Run by tests/test_gpu_torch_hrt.py in a fresh interpreter: the reference's RGAT op sequence (RGAT/models.py:265-385 with the
wrappers of hrt/python/backend/rgnn_layers_and_funcs.py:8-73 and rgat_layers_and_funcs.py:233-316) on NOTHING but the compiled
registration object -- `torch.ops.load_library(libtorch_hrt.so)` as hrt/python/kernels/__init__.py:4-16 does; the het_amd Python
package is never imported -- checked against the fp64 oracle layer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.ops.load_library(os.path.join(ROOT, "het_amd", "libtorch_hrt.so"))
K = torch.ops.torch_hrt
assert "het_amd" not in sys.modules
from oracle import layers as OL  # noqa: E402  (test infrastructure: the checker)

DEV = "cuda"
torch.manual_seed(3)
N, R, E, H, Kd, D = 300, 4, 5000, 4, 64, 16
X = H * D
row, col, rel = torch.randint(0, N, (E,)), torch.randint(0, N, (E,)), torch.randint(0, R, (E,))
# layouts through the compiled layout op (CPU tensors in, as the reference passes them)
rp, r, c, eids = K.convert_integrated_coo_to_separate_coo(row, col, rel, torch.arange(E), N, R)
assert rp[-1] == E and torch.equal(torch.sort(eids).values, torch.arange(E))
# canonical edge ids (position = id), as the reference's graph objects hold them after canonicalize_eids
eids = torch.arange(E)
W = (torch.randn(R, H, Kd, D) * 0.2)
al, ar = torch.randn(R, H, D) * 0.3, torch.randn(R, H, D) * 0.3
loop_w, bias = torch.randn(Kd, X) * 0.1, torch.randn(X) * 0.1
x, go = torch.randn(N, Kd) * 0.5, torch.randn(N, X)

# oracle
p = [t.double().requires_grad_(True) for t in (x, W, al, ar, loop_w, bias)]
ref = OL.rgat_layer(p[0], p[1], p[2], p[3], rp, r, c, N, 0.2, p[4], p[5])
gref = torch.autograd.grad(ref, p, go.double())

d = lambda t: t.to(DEV).contiguous()
rp_d, r_d, c_d, e_d = d(rp), d(r), d(c), d(eids)
by_src = {"separate_coo_rel_ptrs": rp_d, "separate_coo_node_indices": r_d, "separate_coo_eids": e_d}
by_dst = {"separate_coo_rel_ptrs": rp_d, "separate_coo_node_indices": c_d, "separate_coo_eids": e_d}
by_eid = {"separate_coo_rel_ptrs": rp_d, "separate_coo_node_indices": e_d, "separate_coo_eids": e_d}
Wd, ald, ard, xd, god = d(W), d(al), d(ar), d(x), d(go)
z = lambda *s: torch.zeros(*s, device=DEV)
# forward (reference wrappers: zero-filled outputs, "+=" ops)
feat = z(E, H, D); K.rgnn_relational_matmul(by_src, 0, Wd, xd, feat, True)
el = z(E, H, 1); K.rgnn_relational_matmul(by_eid, 0, ald.unsqueeze(-1).contiguous(), feat, el, False)
featd = z(E, H, D); K.rgnn_relational_matmul(by_dst, 0, Wd, xd, featd, True)
er = z(E, H, 1); K.rgnn_relational_matmul(by_eid, 0, ard.unsqueeze(-1).contiguous(), featd, er, False)
s_, ex, ret = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV), torch.empty(N, H, D, device=DEV)
K.relational_fused_gat_separate_coo(e_d, rp_d, r_d, c_d, 0, {}, feat, el.view(E, H), er.view(E, H), s_, ex, ret, 0.2)
out = ret.view(N, X) + xd @ d(loop_w) + d(bias)
# backward
g_feat, g_el, g_er = z(E, H, D), z(E, H), z(E, H)
K.backward_relational_fused_gat_separate_coo(e_d, rp_d, r_d, c_d, 0, {}, feat, el.view(E, H), er.view(E, H), s_, ex, ret,
                                             god.view(N, H, D).contiguous(), g_feat, g_el, g_er, 0.2)
Wt = Wd.transpose(2, 3).contiguous()
# er = <featd, attn_r>
g_featd, g_ar = z(E, H, D), z(R, H, D, 1)
K.backward_rgnn_relational_matmul(by_eid, 0, ard.unsqueeze(-1).transpose(2, 3).contiguous(), featd, g_er.view(E, H, 1).contiguous(), g_featd, g_ar, False)
g_x2, g_W2 = z(N, Kd), z(R, H, Kd, D)
K.backward_rgnn_relational_matmul(by_dst, 0, Wt, xd, g_featd, g_x2, g_W2, True)
# el = <feat, attn_l>
g_feat2, g_al = z(E, H, D), z(R, H, D, 1)
K.backward_rgnn_relational_matmul(by_eid, 0, ald.unsqueeze(-1).transpose(2, 3).contiguous(), feat, g_el.view(E, H, 1).contiguous(), g_feat2, g_al, False)
g_x1, g_W1 = z(N, Kd), z(R, H, Kd, D)
K.backward_rgnn_relational_matmul(by_src, 0, Wt, xd, (g_feat + g_feat2).contiguous(), g_x1, g_W1, True)
g_x = g_x1 + g_x2 + god @ d(loop_w).t()
got = (out, g_x, g_W1 + g_W2, g_al.view(R, H, D), g_ar.view(R, H, D), xd.t() @ god, god.sum(0))
torch.cuda.synchronize()
for name, a, b in zip(("out", "grad_x", "grad_W", "grad_attn_l", "grad_attn_r", "grad_loop_weight", "grad_bias"), got, (ref,) + tuple(gref)):
    err = float((a.cpu().double() - b.detach()).abs().max())
    scale = float(b.detach().abs().max())
    print(f"{name}: max abs err {err:.3e} (scale {scale:.3e})")
    assert err <= 2e-4 * max(1.0, scale), name
print("TORCH_HRT_SEQUENCE_OK")
