"""How the inputs of tests/golden/mag01_full.npz are derived (shared by make_golden.py, which runs the reference's Python on
them in the build container, and by the tests, which regenerate them instead of reading them from the fixture).

The fixture holds the topology the reference ships (hrt/data/ogbn_mag_0.1/*_coo_2.npy, the arrays its own round-trip test
loads whole: hrt/src/test_hyb.cu.cc:26-36) as ONE [3, E] int32 array in file order, the reference builders' layouts as SHA-256
digests (+ the small pointer arrays in full) and the reference's float outputs.  Everything else -- the shuffled integrated COO,
the edge numbering, the float inputs -- comes from numpy's PCG64 streams below (stable across platforms by numpy's
compatibility policy; the fixture also stores digests of the regenerated float inputs, checked first by every test that uses
them, so a drifted stream fails as "inputs differ" rather than as a parity error)."""
import hashlib

import numpy as np
import torch

MAG01_RELATIONS = ("cited", "citing", "has", "is-about", "writing", "written-by")  # relation id = position
SEED_PERM, SEED_EIDS = 7, 8
H, D, SLOPE = 2, 4, 0.2
SEED_GAT, SEED_GATB, SEED_GATC = 2, 12, 22
SEED_GATK1 = 32  # round 5: el / er on the two-sided unique (relation, node) list (CompactAsOfNodeKind 1 and 2)
# the typed view used by the layer tests (HGT needs canonical edge types): ids of the files are local to their node type;
# (source type, destination type) per relation, types 0 paper / 1 author / 2 field of study
MAG01_REL_TYPES = ((0, 0), (0, 0), (2, 0), (0, 2), (1, 0), (0, 1))


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def normal(seed, *shape):
    return torch.from_numpy(_rng(seed).standard_normal(shape, dtype=np.float32))


def digest(t) -> str:
    """SHA-256 of the values as little-endian int64 (indices) or float32 (inputs), C order."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    a = np.ascontiguousarray(a.astype("<i8") if a.dtype.kind in "iu" else a.astype("<f4"))
    return hashlib.sha256(a.tobytes()).hexdigest()


def integrated_coo(coo3):
    """(row, col, rel, eids, num_nodes, num_rels) int64: the file-order edges shuffled (so the integrated COO is not
    relation-sorted) and numbered by a second permutation (edge data row != position), one node-id space of max id + 1 nodes as
    the reference's test builds it (test_hyb.cu.cc:38-52)."""
    a = torch.as_tensor(np.asarray(coo3)).to(torch.int64)
    E = a.shape[1]
    perm = torch.from_numpy(_rng(SEED_PERM).permutation(E))
    eids = torch.from_numpy(_rng(SEED_EIDS).permutation(E))
    row, col, rel = a[0][perm].contiguous(), a[1][perm].contiguous(), a[2][perm].contiguous()
    return row, col, rel, eids, int(max(row.max(), col.max())) + 1, len(MAG01_RELATIONS)


def typed_coo(coo3):
    """The same edges with the node ids of the three node types laid end to end (paper, author, field of study) in file
    (relation-major) order with eids = arange: (row, col, rel, node_type_offsets).  A heterogeneous graph with canonical edge
    types, as HGT needs; the pinned layouts are those of integrated_coo() -- this view only feeds layer tests against the oracle."""
    a = torch.as_tensor(np.asarray(coo3)).to(torch.int64)
    size = [0, 0, 0]
    for r, (ts, td) in enumerate(MAG01_REL_TYPES):
        m = a[2] == r
        size[ts] = max(size[ts], int(a[0][m].max()) + 1)
        size[td] = max(size[td], int(a[1][m].max()) + 1)
    off = torch.tensor([0, size[0], size[0] + size[1], size[0] + size[1] + size[2]])
    st = torch.tensor([t[0] for t in MAG01_REL_TYPES])[a[2]]
    dt = torch.tensor([t[1] for t in MAG01_REL_TYPES])[a[2]]
    return (a[0] + off[st]).contiguous(), (a[1] + off[dt]).contiguous(), a[2].contiguous(), off


def gat_inputs(E, n, s_row, s_col):
    """Float inputs of the three reference runs: kind-0 forward (el, er per edge), its backward (gradout per node), kind-4
    forward (el per (relation, source) row, er per (relation, destination) row)."""
    return {"gat_el": normal(SEED_GAT, E, H), "gat_er": normal(SEED_GAT + 100, E, H),
            "gatb_gradout": normal(SEED_GATB, n, H, D),
            "gatc_el": normal(SEED_GATC, s_row, H), "gatc_er": normal(SEED_GATC + 100, s_col, H)}


def two_sided_inverse(ts_inverse_indices, sep_rel_ptrs):
    """(row side, col side) [E] of the two-sided inverse index in separate-COO position order.  The reference builder
    (mydgl_graph_methods.py:104-157) returns it as [rows of relation 0, cols of relation 0, rows of relation 1, ...]: for position
    i of relation r the row side sits at 2 * rel_ptrs[r] + (i - rel_ptrs[r]), the col side n_r entries further."""
    rows, cols = [], []
    R = sep_rel_ptrs.numel() - 1
    for r in range(R):
        a, b = int(sep_rel_ptrs[r]), int(sep_rel_ptrs[r + 1])
        rows.append(ts_inverse_indices[2 * a: 2 * a + (b - a)])
        cols.append(ts_inverse_indices[2 * a + (b - a): 2 * b])
    return torch.cat(rows), torch.cat(cols)


def gat_inputs_round5(ts_rows):
    """el / er per row of the two-sided unique list: the kind-1 forward (dual wrapper fed the two-sided inverse index) and the
    kind-2 forward (single-list wrapper) of the reference."""
    return {"gatk1_el": normal(SEED_GATK1, ts_rows, H), "gatk1_er": normal(SEED_GATK1 + 100, ts_rows, H)}


def canonical_csr(row_ptrs, col, rel, eids):
    """The entries of every CSR row sorted by (col, rel, eid): the reference's coo2csr argsort is not stable, so a CSR is pinned
    up to the order inside a row."""
    n = row_ptrs.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n), row_ptrs[1:] - row_ptrs[:-1])
    order = np.lexsort((eids.numpy(), rel.numpy(), col.numpy(), rows.numpy()))
    order = torch.from_numpy(order)
    return col[order], rel[order], eids[order]
