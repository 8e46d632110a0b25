"""The RGAT forward gather pass alone (for counter collection): python3 exp/fwd_only.py [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import het_amd.kernels as k
from het_amd.graph import HetGraph
from het_amd.synth import make_mag_like
dev = torch.device("cuda")
coo = make_mag_like(scale=1.0)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
s = g.get_separate_coo_original()
ss = g.get_separate_unique_node_indices_single_sided()
inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
N, H, D = g.get_num_nodes(), 4, 16
S_row, S_col = ss["node_indices_row"].numel(), ss["node_indices_col"].numel()
grp = k.rgat_compact_groupings(s["col_indices"], inv["inverse_indices_row"], inv["inverse_indices_col"], N, S_row, S_col)
feat = torch.randn(S_row, H, D, device=dev) * 0.1
el, er = torch.randn(S_row, H, device=dev) * 0.1, torch.randn(S_col, H, device=dev) * 0.1
sm, ret = torch.empty(N, H, device=dev), torch.empty(N, H, D, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    k.rgat_aggregate_compact(grp, feat, el, er, sm, ret, 0.2)
torch.cuda.synchronize()
