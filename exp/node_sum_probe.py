"""Where does the node-major pass of the RGCN layer spend its time?  HIP-event times of het_rgcn_layer_forward's two launches
(segment sum / node pass) and of the node pass alone in variants, ogbn-mag-shaped graph, feat 64.
  python exp/node_sum_probe.py            all variants
  python exp/node_sum_probe.py one        10 launches of the forward node pass (for rocprofv3 --pmc)"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from het_amd import kernels as k, _lib
from het_amd.graph import HetGraph
from het_amd.synth import make_mag_like

dev = "cuda"
coo = make_mag_like(scale=1.0)
for f in ("row", "col", "rel", "eids", "node_type_offsets"):
    setattr(coo, f, getattr(coo, f).to(dev))
g = HetGraph.from_integrated_coo(coo, full=False)
s = g.get_separate_coo_original()
N, R, K = g.get_num_nodes(), g.get_num_rels(), 64
plan = k.rgcn_layer_plan(s["rel_ptrs"], s["eids"], s["row_indices"], s["col_indices"], N)
gd, gs, dst_map, dst_order, src_map, src_order = plan
torch.manual_seed(0)
W = torch.randn(R, K, K, device=dev) * 0.1
bias = torch.randn(K, device=dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def node_pass(rows, maps, order, out, nsrc=R, n_begin=0, n_end=N):
    srcs = [(rows, 0, maps[r], W[r]) for r in range(nsrc)]
    k.node_rows_matmul_sum(n_begin, n_end, srcs, out, order)


S_col, S_row = gd.num_segments, gs.num_segments
ssum = torch.randn(S_col, K, device=dev)
gsum = torch.randn(S_row, K, device=dev)
out = torch.empty(N, K, device=dev)
dmap_node = k._grouping_segment_map(gd, s["rel_ptrs"], s["col_indices"], N)
smap_node = k._grouping_segment_map(gs, s["rel_ptrs"], s["row_indices"], N)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    for _ in range(10):
        node_pass(ssum, dmap_node, dst_order, out)
    torch.cuda.synchronize()
    sys.exit(0)
nz = int((dmap_node >= 0).any(0).sum())
print(f"N={N} S_col={S_col} S_row={S_row} nodes with in-edges={nz}")
print("forward node pass, sorted order      %.4f ms" % timed(lambda: node_pass(ssum, dmap_node, dst_order, out)))
print("forward node pass, node-id order     %.4f ms" % timed(lambda: node_pass(ssum, dmap_node, None, out)))
print("backward node pass, sorted order     %.4f ms" % timed(lambda: node_pass(gsum, smap_node, src_order, out)))
print("backward node pass, node-id order    %.4f ms" % timed(lambda: node_pass(gsum, smap_node, None, out)))
# only the positions that have rows (the empty class is in front of the sorted list)
first = N - nz
print("forward, sorted, non-empty tail only %.4f ms  (positions %d..%d)" % (timed(lambda: node_pass(ssum, dmap_node, dst_order, out, n_begin=first)), first, N))
print("forward, sorted, empty head only     %.4f ms" % timed(lambda: node_pass(ssum, dmap_node, dst_order, out, n_end=first)))
for ns in (1, 2):
    print("forward, sorted, %d source(s)          %.4f ms" % (ns, timed(lambda: node_pass(ssum, dmap_node, dst_order, out, nsrc=ns))))
# streaming reference: copy of the same bytes
a_ = torch.empty(S_col + N, K, device=dev)
b_ = torch.empty_like(a_)
print("copy of (S_col + N) rows             %.4f ms" % timed(lambda: b_.copy_(a_)))
print("fill of N rows                       %.4f ms" % timed(lambda: out.fill_(1.0)))
