"""Would blocking the gathers by table range pay?  The RGAT forward gather pass (het_rgat_aggregate_compact) with the
(relation, source) row of every edge folded into the first 1/f of the table: same edges, same destinations, same number of
gathers, but the table they hit shrinks from 0.6 GB to 0.6/f GB (below the 256 MiB Infinity Cache for f >= 4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import het_amd.kernels as k
from het_amd import _lib as HL
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
srow, drow = inv["inverse_indices_row"], inv["inverse_indices_col"]
feat = torch.randn(S_row, H, D, device=dev) * 0.1
el, er = torch.randn(S_row, H, device=dev) * 0.1, torch.randn(S_col, H, device=dev) * 0.1
sm, ret = torch.empty(N, H, device=dev), torch.empty(N, H, D, device=dev)
for fold in (1, 2, 4, 8, 16, 64, 256, 1024, 16384):
    m = (S_row + fold - 1) // fold
    sr = (srow % m).contiguous()
    grp = k.rgat_compact_groupings(s["col_indices"], sr, drow, N, S_row, S_col)
    for _ in range(3):
        k.rgat_aggregate_compact(grp, feat, el, er, sm, ret, 0.2)
    HL.kernel_timing(True)
    for _ in range(10):
        k.rgat_aggregate_compact(grp, feat, el, er, sm, ret, 0.2)
    ms, n = HL.kernel_timing_read("HET_rgat_aggregate")
    HL.kernel_timing(False)
    print(f"table {m * H * D * 4 / 1e6:7.1f} MB (1/{fold}): {ms / n:.3f} ms per launch, {coo.row.numel() * 256 / (ms / n * 1e-3) / 1e12:.2f} TB/s of row gathers", flush=True)
    del grp
    import het_amd.plan as plan
    plan.clear()
