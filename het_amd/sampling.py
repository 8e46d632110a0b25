"""Mini-batch (sampled block) path: neighbour sampling and per-batch layout regeneration on the GPU.

The reference reaches mini-batch training only through DGL's samplers (``MultiLayerNeighborSampler`` /
``DataLoader``, hrt/python/RGNNUtils/RGNNUtils.py:164-196) and converts every sampled block to its own layouts on the
CPU (hrt/python/utils/mydglgraph_converters.py:18-71).  Here both steps run on the device: uniform sampling of at most
``fanout`` in-edges per destination with vectorised torch ops over the in-CSR, then ``HetGraph.from_integrated_coo``
on the block (the native builders of layouts.hip), so that the same ops and layers run on the block unchanged.

A block follows DGL's convention: its destination nodes are the first ``num_dst`` of its nodes; a layer's output rows
``[:num_dst]`` are the inputs of the next block.  HGT's per-node-type linears need runs of equal node type: with
``by_type=True`` the seeds are ordered by type and every block's nodes are a few type-sorted runs
(``node_type_offsets`` of the block = the run boundaries, ``node_segment_types`` = the type of every run), over which the
layer indexes its per-type weights.
"""
from __future__ import annotations

import contextlib
import dataclasses
from typing import List, Optional

import torch

from .graph import HetGraph
from .synth import IntegratedCOO


@dataclasses.dataclass
class Block:
    graph: HetGraph            # layouts of the sampled bipartite-style subgraph, local node ids
    nodes: torch.Tensor        # [num_nodes] global id of every local node; the first num_dst are the destinations
    num_dst: int
    edge_ids: torch.Tensor     # [E_block] global edge id of every block edge, in the block's separate-COO order
    runs: Optional[tuple] = None  # by_type: (types [n_runs], offsets [n_runs + 1]) of the type-sorted runs of `nodes`


class NeighborSampler:
    """Uniform in-neighbour sampling without replacement, ``fanouts[l]`` edges per destination for layer l
    (-1 or 0: all in-edges), as dgl.dataloading.MultiLayerNeighborSampler does for the reference."""

    def __init__(self, g: HetGraph, fanouts: List[int], seed: int = 0, full_layouts: bool = True, by_type: bool = False):
        t = g.get_in_csr()  # rows = destinations, col_indices = sources
        self.ptr, self.src, self.rel, self.eid = t["row_ptrs"], t["col_indices"], t["rel_types"], t["eids"]
        self.num_nodes, self.num_rels = g.get_num_nodes(), g.get_num_rels()
        self.fanouts = list(fanouts)
        # full_layouts False: blocks get the separate COO only (all the default-flag RGAT / RGCN layers read); True adds the
        # CSRs and the unique (relation, node) lists of the compact flags and the CSR ops
        self.full_layouts = full_layouts
        self.dev = self.ptr.device
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(seed)
        self._map = torch.full((self.num_nodes,), -1, dtype=torch.int64, device=self.dev)
        self.by_type = by_type
        self.type_offsets = g.get_original_node_type_offsets().to(self.dev)  # node ids are type-contiguous in the full graph
        self.num_types = int(self.type_offsets.numel() - 1)

    def _type_counts(self, nodes: torch.Tensor) -> torch.Tensor:
        t = torch.searchsorted(self.type_offsets[1:].contiguous(), nodes, right=True).clamp(max=self.num_types - 1)
        return torch.bincount(t, minlength=self.num_types)

    def _sample_in_edges(self, dst: torch.Tensor, fanout: int):
        deg = self.ptr[dst + 1] - self.ptr[dst]
        total = int(deg.sum())
        B = dst.numel()
        first = torch.cumsum(deg, 0) - deg
        owner = torch.repeat_interleave(torch.arange(B, device=self.dev), deg, output_size=total)
        within = torch.arange(total, device=self.dev) - first[owner]
        pos = self.ptr[dst][owner] + within
        if fanout > 0 and total > 0 and int(deg.max()) > fanout:
            r = torch.rand(total, device=self.dev, generator=self.gen, dtype=torch.float64)
            order = torch.sort(owner.to(torch.float64) + r).indices  # groups stay contiguous, random order inside
            keep = order[within < fanout]                            # the first `fanout` of every group
            keep = torch.sort(keep).values                           # back to CSR order (deterministic layouts)
            pos, owner = pos[keep], owner[keep]
        return pos, owner

    def sample_block(self, dst: torch.Tensor, fanout: int, dst_runs=None) -> Block:
        """``dst_runs`` (by_type): (types [n_runs], offsets [n_runs + 1]) of the type-sorted runs ``dst`` consists of."""
        pos, owner = self._sample_in_edges(dst, fanout)
        src_g = self.src[pos]
        B = dst.numel()
        m = self._map
        m[dst] = torch.arange(B, device=self.dev)
        extra = torch.unique(src_g[m[src_g] < 0])
        m[extra] = B + torch.arange(extra.numel(), device=self.dev)
        nodes = torch.cat([dst, extra])
        rel = self.rel[pos]
        o = torch.sort(rel, stable=True).indices  # relation-major, as the integrated COO of a graph is stored
        runs = None
        if dst_runs is not None:  # `extra` is sorted by global id, i.e. by type: one more run per type
            types = torch.cat([dst_runs[0], torch.arange(self.num_types, device=self.dev)])
            offs = torch.cat([dst_runs[1], B + torch.cumsum(self._type_counts(extra), 0)])
            runs = (types, offs)
        node_offs = torch.tensor([0, int(nodes.numel())], device=self.dev) if runs is None else runs[1]
        coo = IntegratedCOO(int(nodes.numel()), self.num_rels, node_offs,
                            m[src_g][o].contiguous(), owner[o].contiguous(), rel[o].contiguous(),
                            torch.arange(pos.numel(), device=self.dev))
        m[nodes] = -1  # reset the scratch map
        g = HetGraph.from_integrated_coo(coo, full=self.full_layouts)
        if runs is not None:
            g.graph_data["original"]["node_segment_types"] = runs[0]
        return Block(g, nodes, B, self.eid[pos][o].contiguous(), runs)

    def sample_blocks(self, seeds: torch.Tensor) -> List[Block]:
        """Blocks in layer order (first layer first); blocks[-1].nodes[:num_dst] == seeds (by_type: the seeds ordered by
        node type -- read them back from there) and blocks[l].nodes[:blocks[l].num_dst] == blocks[l+1].nodes."""
        blocks: List[Block] = []
        dst, runs = seeds, None
        if self.by_type:
            t = torch.searchsorted(self.type_offsets[1:].contiguous(), seeds, right=True).clamp(max=self.num_types - 1)
            dst = seeds[torch.sort(t, stable=True).indices]
            z = torch.zeros(1, dtype=torch.int64, device=self.dev)
            runs = (torch.arange(self.num_types, device=self.dev), torch.cat([z, torch.cumsum(self._type_counts(dst), 0)]))
        for fanout in reversed(self.fanouts):
            b = self.sample_block(dst, fanout, runs)
            blocks.insert(0, b)
            dst, runs = b.nodes, b.runs
        return blocks


# Blocks below this many edges are used for one forward + backward only and are small: building the ops' groupings
# (a few sorts + host round trips each) costs more than it saves -- 1.8 ms per 1024-seed batch of fanout 25 / 20 on
# ogbn-mag (forward 2.34 -> 0.56 ms on the kernels that need no preprocessing).
ONE_SHOT_MIN_EDGES = 1 << 20


@contextlib.contextmanager
def one_shot_graphs(blocks: Optional[List[Block]] = None, min_edges: int = ONE_SHOT_MIN_EDGES):
    """Wrap one whole training step on sampled blocks -- forward AND backward -- in this: when every block is small the
    ops run on their preprocessing-free kernels (het_amd.plan.forced(False)) for the duration."""
    from . import plan
    small = blocks is None or all(b.graph.get_num_edges() < min_edges for b in blocks)
    # a per-thread override, not a process-wide switch: a full-graph model running elsewhere in the process keeps its
    # groupings, and every autograd node created here carries the choice into its backward (plan.consistent) -- also when
    # loss.backward() runs after the block has been left or on an autograd worker thread
    with plan.forced(False if small else None):
        yield


def run_blocks(layers, blocks: List[Block], h: torch.Tensor, edge_data: Optional[torch.Tensor] = None):
    """``h`` = features of blocks[0].nodes.  Applies layer l to block l and keeps the destination rows.
    ``edge_data`` (e.g. RGCN's norm, indexed by global edge id) is gathered per block."""
    for layer, b in zip(layers, blocks):
        extra = () if edge_data is None else (edge_data[b.edge_ids],)
        h = layer(b.graph, h, *extra, num_dst=b.num_dst)  # destination rows only (self-loop, bias, activation on them)
    return h
