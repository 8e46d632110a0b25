"""Synthetic heterogeneous graphs shaped like the reference's benchmark datasets.

No dataset can be downloaded here (no network, no DGL/OGB), so the bench and the
full-size tests run on graphs that reproduce the *shape* of what the reference
loads: node-type ranges, per-relation edge counts, (src-type -> dst-type)
signature and a skewed degree distribution.

ogbn-mag as the reference builds it (DGL ``to_homogeneous``, no reverse edges,
/root/reference/hrt/python/utils_lite/graphiler_datasets.py:80-82,146-148;
node counts from hrt/python/utils/loaders_from_npy.py:268-269):
node types in alphabetical order author / field_of_study / institution / paper,
relations in alphabetical order affiliated_with / cites / has_topic / writes.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

MAG_NODE_TYPES = ("author", "field_of_study", "institution", "paper")
MAG_NODE_COUNTS = (1_134_649, 59_965, 8_740, 736_389)
# (name, src_type, dst_type, num_edges)
MAG_RELATIONS = (
    ("affiliated_with", "author", "institution", 1_043_998),
    ("cites", "paper", "paper", 5_416_271),
    ("has_topic", "paper", "field_of_study", 7_505_078),
    ("writes", "author", "paper", 7_145_660),
)
SEED = 20240427


@dataclasses.dataclass
class IntegratedCOO:
    """Integrated COO in the reference's "original" orientation: row = src,
    col = dst, relation-major edge order, eids = arange(E)
    (hrt/python/utils/mydglgraph_converters.py:625-680)."""

    num_nodes: int
    num_rels: int
    node_type_offsets: torch.Tensor  # int64 [T+1]
    row: torch.Tensor  # int64 [E] src
    col: torch.Tensor  # int64 [E] dst
    rel: torch.Tensor  # int64 [E]
    eids: torch.Tensor  # int64 [E]

    @property
    def num_edges(self) -> int:
        return int(self.row.numel())


def _skewed_ids(rng: np.random.Generator, n: int, size: int, offset: float) -> np.ndarray:
    """Draw ``size`` ids in [0, n) with p(rank) ~ 1/(rank+offset) (a Zipf(1)-like
    law softened by ``offset``), then hide the rank behind a fixed permutation so
    that popularity does not correlate with the node id."""
    u = rng.random(size)
    lo, hi = np.log(offset), np.log(n + offset)
    rank = np.floor(np.exp(lo + u * (hi - lo)) - offset).astype(np.int64)
    np.clip(rank, 0, n - 1, out=rank)
    perm = rng.permutation(n)
    return perm[rank]


def make_hetero_graph(
    node_counts,
    relations,
    seed: int = SEED,
    scale: float = 1.0,
    edge_order: str = "src",
) -> IntegratedCOO:
    """relations: iterable of (src_type_idx, dst_type_idx, num_edges).

    ``scale`` shrinks node and edge counts together (tests use ~1e-3).
    ``edge_order``: "src_dst" lists each relation's edges in (source, destination) order -- the order of the OGB
    edge lists (the reference's shipped slice hrt/data/ogbn_mag_0.1/*_coo_2.npy: every relation is sorted by one end and
    ascending in the other within equal keys); "src" sorts by source id only (destinations of a source in generation
    order: rounds 1-2), "random" keeps generation order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    counts = [max(2, int(round(c * scale))) for c in node_counts]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rows, cols, rels = [], [], []
    for r, (st, dt, ne) in enumerate(relations):
        ne = max(1, int(round(ne * scale)))
        src = _skewed_ids(rng, counts[st], ne, offset=64.0) + offs[st]
        dst = _skewed_ids(rng, counts[dt], ne, offset=16.0) + offs[dt]
        if edge_order == "src":
            o = np.argsort(src, kind="stable")
            src, dst = src[o], dst[o]
        elif edge_order == "src_dst":
            o = np.lexsort((dst, src))
            src, dst = src[o], dst[o]
        elif edge_order != "random":
            raise ValueError(edge_order)
        rows.append(src)
        cols.append(dst)
        rels.append(np.full(ne, r, dtype=np.int64))
    row = torch.from_numpy(np.concatenate(rows))
    col = torch.from_numpy(np.concatenate(cols))
    rel = torch.from_numpy(np.concatenate(rels))
    return IntegratedCOO(
        num_nodes=int(offs[-1]),
        num_rels=len(relations),
        node_type_offsets=torch.from_numpy(offs.copy()),
        row=row,
        col=col,
        rel=rel,
        eids=torch.arange(row.numel(), dtype=torch.int64),
    )


def make_mag_like(scale: float = 1.0, seed: int = SEED, edge_order: str = "src_dst") -> IntegratedCOO:
    t = {n: i for i, n in enumerate(MAG_NODE_TYPES)}
    rels = [(t[s], t[d], e) for (_, s, d, e) in MAG_RELATIONS]
    return make_hetero_graph(MAG_NODE_COUNTS, rels, seed=seed, scale=scale, edge_order=edge_order)


def make_aifb_like(num_rels: int = 4, seed: int = SEED) -> IntegratedCOO:
    """AIFB-sized graph (BASELINE.json configs[0]): N = 8285, E = 58086, one
    node type; ``num_rels`` 4 (as BASELINE.json states) or 104 (what DGL's AIFB
    yields, hrt/python/test/test_graphiler_load_data.py:18)."""
    n, e = 8285, 58086
    per = [e // num_rels] * num_rels
    per[-1] += e - sum(per)
    return make_hetero_graph([n], [(0, 0, c) for c in per], seed=seed)


def make_random(num_nodes: int, num_rels: int, num_edges: int, seed: int = 0, num_ntypes: int = 1) -> IntegratedCOO:
    """Small uniformly random multigraph for unit tests (ragged relations,
    possibly empty ones)."""
    g = torch.Generator().manual_seed(seed)
    rel = torch.sort(torch.randint(0, num_rels, (num_edges,), generator=g)).values
    row = torch.randint(0, num_nodes, (num_edges,), generator=g)
    col = torch.randint(0, num_nodes, (num_edges,), generator=g)
    cuts = torch.sort(torch.randint(0, num_nodes + 1, (num_ntypes - 1,), generator=g)).values
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), cuts, torch.tensor([num_nodes])])
    return IntegratedCOO(num_nodes, num_rels, offs, row, col, rel, torch.arange(num_edges, dtype=torch.int64))
