"""Scratch: the run sums of het_rgat_aggregate_compact_runs against torch (Q, q relative to ref) and grad_er from them."""
import torch, sys
sys.path.insert(0, ".")
import het_amd.kernels as k
from tests.util import random_graph
from tests.test_gpu_ops import _gat_case
DEV = "cuda:0"
H, D, n, e = 4, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 300, 5000
g = random_graph(seed=27, n=n, r=4, e=e)
s, feat, el, er, go, df, db = _gat_case(g, 4, H, D, seed=11)
N, R = g.get_num_nodes(), g.get_num_rels()
ss = g.get_separate_unique_node_indices_single_sided()
srow = df["edata_idx_to_inverse_idx_row"][s["eids"]].contiguous()
drow = df["edata_idx_to_inverse_idx_col"][s["eids"]].contiguous()
col = s["col_indices"]
z = el[srow] + er[drow]
sv = torch.nn.functional.leaky_relu(z, 0.2).double()
dl = torch.where(z > 0, 1.0, 0.2).double()
w = torch.exp(sv)
Q = torch.zeros(er.shape[0], H, D, dtype=torch.float64).index_add_(0, drow, (w * dl).unsqueeze(-1) * feat[srow].double())
q = torch.zeros(er.shape[0], H, dtype=torch.float64).index_add_(0, drow, w * dl)
grp = k.rgat_compact_groupings(col.to(DEV), srow.to(DEV), drow.to(DEV), N, feat.shape[0], er.shape[0], rel_ptrs=s["rel_ptrs"].to(DEV))
sm, ret = torch.zeros(N, H, device=DEV), torch.zeros(N, H, D, device=DEV)
qr, qs, qf = k.rgat_aggregate_compact(grp, feat.to(DEV), el.to(DEV), er.to(DEV), sm, ret, 0.2, num_rels=R)
sc = torch.exp(qf.double().cpu())
print("q   max err", float((qs.double().cpu() * sc - q).abs().max()), "scale", float(q.abs().max()))
print("Q   max err", float((qr.double().cpu() * sc.unsqueeze(-1) - Q).abs().max()), "scale", float(Q.abs().max()))
print("drow_nodes consistent", bool((ss["node_indices_col"][drow] == col).all()))
