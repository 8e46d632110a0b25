"""Would blocking the gathers by table range pay on the round-3 kernels?  (VERDICT r03 "Next" 2.)

The run-sum forward (HET_rgat_aggregate_runs_packed + _hub_items + _finish_hubs) and the run-sum backward
(HET_rgat_dst_pack + _backward_src_coop + _src_long + _grad_er_runs) of the one-node RGAT layer on the ogbn-mag-shaped graph,
with the per-edge gather folded into the first 1/f of its table: same edges, same segments, same number of 256-byte gathers, but
the rows they touch shrink from 0.61 GB (feat_c, forward) / 0.50 GB (gradout, backward) to 1/f of that.

  python3 exp/locality_r04.py                 # sweep, per-kernel HIP-event times (profiles/r04/locality_fold.txt)
  python3 exp/locality_r04.py one <fold> <n>  # n launches of both ops at one fold (under rocprofv3 --pmc)
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import het_amd.kernels as k
import het_amd.plan as plan
from het_amd import _lib as HL
from het_amd.graph import HetGraph
from het_amd.synth import make_mag_like

dev = torch.device("cuda")
coo = make_mag_like(scale=float(os.environ.get("SCALE", "1.0")))
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=True)
s = g.get_separate_coo_original()
ss = g.get_separate_unique_node_indices_single_sided()
inv = g.get_separate_unique_node_indices_single_sided_inverse_idx()
N, H, D, R = g.get_num_nodes(), 4, 16, g.get_num_rels()
E = s["col_indices"].numel()
S_row, S_col = ss["node_indices_row"].numel(), ss["node_indices_col"].numel()
col, srow, drow, rp = s["col_indices"], inv["inverse_indices_row"], inv["inverse_indices_col"], s["rel_ptrs"]
torch.manual_seed(0)
feat = torch.randn(S_row, H, D, device=dev) * 0.1
el, er = torch.randn(S_row, H, device=dev) * 0.1, torch.randn(S_col, H, device=dev) * 0.1
go = torch.randn(N, H * D, device=dev) * 0.1
attn_l = torch.randn(R, H, D, device=dev) * 0.1
sm, ret = torch.empty(N, H, device=dev), torch.empty(N, H, D, device=dev)
g_featc, g_elc, g_erc = torch.empty_like(feat), torch.empty_like(el), torch.empty_like(er)
true_grp = k.rgat_compact_groupings(col, srow, drow, N, S_row, S_col, rel_ptrs=rp, drow_nodes=ss["node_indices_col"],
                                    drow_rel_ptrs=ss["rel_ptrs_col"])
# (the library's timer labels: csrc/gat_compact.hip HET_KTIME; kernels HET_rgat_aggregate_runs_packed / _hub_items / HET_rgat_finish_hubs,
#  HET_rgat_dst_pack / _backward_src_coop / _backward_src_long / HET_rgat_grad_er_runs)
FWD = ("HET_rgat_aggregate_packs", "HET_rgat_aggregate_hubs", "HET_rgat_aggregate_finish")
BWD = ("HET_rgat_backward_dst_pack", "HET_rgat_backward_src_short", "HET_rgat_backward_src_long", "HET_rgat_backward_er_runs")


def groupings(fold):
    """forward: the feat row of every edge folded into the first S_row / fold rows (payload of the destination groupings);
    backward: the destination whose gradout row an edge gathers folded into the first N / fold nodes (payload of the
    (relation, source) grouping).  Segments, hubs, runs and the er rows are those of the true graph."""
    if fold == 1:
        return true_grp
    ms, mn = (S_row + fold - 1) // fold, (N + fold - 1) // fold
    sr = (srow % ms).contiguous()
    cf = (col % mn).contiguous()
    by_dst = plan.get_grouping(None, col, N, sr, drow)
    by_srow = plan.get_grouping(None, srow, S_row, cf, drow)
    return by_dst, by_srow, None, true_grp[3]


def step(grp):
    runs = k.rgat_aggregate_compact(grp, feat, el, er, sm, ret, 0.2, num_rels=R)
    k.rgat_backward_compact(grp, feat, el, er, sm, ret, go, g_featc, g_elc, g_erc, 0.2, fold_attn_l=attn_l,
                            row_rel_ptrs=ss["rel_ptrs_row"], runs=runs, drow_nodes=ss["node_indices_col"])


if len(sys.argv) > 1 and sys.argv[1] == "one":
    grp = groupings(int(sys.argv[2]))
    for _ in range(int(sys.argv[3])):
        step(grp)
    torch.cuda.synchronize()
    sys.exit(0)

os.environ.setdefault("HET_SIDE_STREAM", "0")  # every launch alone on the chip: per-kernel durations add up
print(f"E {E}  S_row {S_row}  S_col {S_col}  N {N}  (HET_SIDE_STREAM={os.environ['HET_SIDE_STREAM']})")
print("fold  feat MB  gradout MB | " + "  ".join(n[9:] for n in FWD) + " = fwd ms | " + "  ".join(n[9:] for n in BWD) + " = bwd ms")
for fold in (1, 2, 4, 8, 16, 64, 256, 1024):
    grp = groupings(fold)
    for _ in range(3):
        step(grp)
    HL.kernel_timing(True)
    for _ in range(10):
        step(grp)
    t = {}
    for n in FWD + BWD:
        ms, cnt = HL.kernel_timing_read(n)
        t[n] = ms / 10
    HL.kernel_timing(False)
    f, b = sum(t[n] for n in FWD), sum(t[n] for n in BWD)
    print(f"{fold:5d} {S_row * 256 / fold / 1e6:8.1f} {N * 256 / fold / 1e6:8.1f} | " + "  ".join(f"{t[n]:.3f}" for n in FWD) + f" = {f:.3f} | "
          + "  ".join(f"{t[n]:.3f}" for n in BWD) + f" = {b:.3f}   row gathers {E * 256 / f / 1e9:.2f} / {E * 256 / b / 1e9:.2f} TB/s", flush=True)
    del grp
