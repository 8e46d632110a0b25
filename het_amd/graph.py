"""Graph container and layout builders for the relational-GNN hot path.

Host-side mirror of the duck-typed graph argument ``g`` that the reference's
backend wrappers take (``MyDGLGraph``, /root/reference/hrt/python/utils/
mydgl_graph.py:81-1074; getters :281-419): a nested dict of int64 tensors
with the same keys and the same getter names, so the ``backend`` functions of
this package and the reference's model code see the same object.

Layouts:
  original    integrated COO, row = src, col = dst, rel_types, eids
  transposed  in-CSR: rows = dst, col_indices = src (a real transpose; the
              reference's graphiler path copies instead)
  separate/coo/original   edges bucketed by relation (``rel_ptrs``), inside a
              bucket sorted by eid (hrt/python/utils/coo_sorters.py:155-170)
  separate/unique_node_indices[_single_sided]   sorted unique (relation, node)
              lists + inverse indices (hrt/python/utils_lite/
              mydgl_graph_methods.py:10-157)

Builders: on GPU tensors the native device-side builders of libhet_amd.so
(het_amd/csrc/layouts.hip: one hipCUB radix sort + gather / boundary / run-length
kernels per conversion); on CPU tensors the same conversions as vectorised torch
code (the reference's run on CPU with C++ vector-of-vector bucketing and Python
loops).  Both produce the reference builders' results exactly (tests/golden).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import ctypes as _C

from . import _lib
from .synth import IntegratedCOO

_I64 = torch.int64


def _p(t):
    return None if t is None else _C.c_void_p(t.data_ptr())


def _native(name, ref, *args):
    """Call a het_layout_* entry point on the device / current stream of ``ref``."""
    with torch.cuda.device(ref.device):
        _lib.call(name, *args, _C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream))


def _c64(t):
    return t.to(_I64).contiguous()


def _stable_argsort(x: torch.Tensor) -> torch.Tensor:
    return torch.sort(x, stable=True).indices


def _ptrs_from_sorted(keys: torch.Tensor, n: int) -> torch.Tensor:
    counts = torch.bincount(keys, minlength=n)
    ptrs = torch.zeros(n + 1, dtype=_I64, device=keys.device)
    torch.cumsum(counts, 0, out=ptrs[1:])
    return ptrs


def coo_to_csr(row, col, rel, eids, num_rows: int):
    """Integrated COO -> CSR over ``row`` (stable, so ties keep COO order;
    reference: hrt/python/utils_lite/sparse_matrix_converters.py:6-39, which
    uses an unstable argsort)."""
    if row.is_cuda:
        row, col, rel, eids = _c64(row), _c64(col), _c64(rel), _c64(eids)
        E = row.numel()
        ptrs = torch.empty(num_rows + 1, dtype=_I64, device=row.device)
        c, r, e = torch.empty_like(col), torch.empty_like(rel), torch.empty_like(eids)
        _native("het_layout_coo_to_csr", row, _p(row), _p(col), _p(rel), _p(eids), E, num_rows, _p(ptrs), _p(c), _p(r), _p(e))
        return ptrs, c, r, e
    o = _stable_argsort(row)
    return _ptrs_from_sorted(row[o], num_rows), col[o], rel[o], eids[o]


def csr_to_coo_rows(row_ptrs: torch.Tensor) -> torch.Tensor:
    n = row_ptrs.numel() - 1
    return torch.repeat_interleave(
        torch.arange(n, dtype=_I64, device=row_ptrs.device), row_ptrs[1:] - row_ptrs[:-1]
    )


def transpose_csr(row_ptrs, col_indices, eids, rel_types):
    """CSR -> CSR of the transposed adjacency (reference op ``transpose_csr``,
    hrt/include/DGLHackKernel/OpExport/DataConverters.inc.h:283-344; restated
    in hrt/python/testing/adjacency_manipulation.py:72-107)."""
    n = max(int(row_ptrs.numel() - 1), int(col_indices.max().item()) + 1 if col_indices.numel() else 0)
    if row_ptrs.is_cuda:
        row_ptrs, col_indices, eids, rel_types = _c64(row_ptrs), _c64(col_indices), _c64(eids), _c64(rel_types)
        E = col_indices.numel()
        ptrs = torch.empty(n + 1, dtype=_I64, device=row_ptrs.device)
        c, e, r = torch.empty_like(col_indices), torch.empty_like(eids), torch.empty_like(rel_types)
        _native("het_layout_transpose_csr", row_ptrs, _p(row_ptrs), _p(col_indices), _p(eids), _p(rel_types),
                row_ptrs.numel() - 1, E, n, _p(ptrs), _p(c), _p(e), _p(r))
        return ptrs, c, e, r
    rows = csr_to_coo_rows(row_ptrs)
    ptrs, c, r, e = coo_to_csr(col_indices, rows, rel_types, eids, n)
    return ptrs, c, e, r


def integrated_coo_to_separate_coo(row, col, rel, eids, num_rels: int):
    """Bucket edges by relation, each bucket sorted by eid (reference:
    ``convert_integrated_coo_to_separate_coo`` hrt/include/MyHyb/MyHyb.h:1047-1096
    followed by ``sort_coo_by_etype_eids_torch_tensors``)."""
    if row.is_cuda:
        row, col, rel, eids = _c64(row), _c64(col), _c64(rel), _c64(eids)
        E = row.numel()
        rp = torch.empty(num_rels + 1, dtype=_I64, device=row.device)
        r, c, e = torch.empty_like(row), torch.empty_like(col), torch.empty_like(eids)
        bound = int(eids.max().item()) + 1 if E else 1
        _native("het_layout_separate_coo", row, _p(row), _p(col), _p(rel), _p(eids), E, num_rels, bound, _p(rp), _p(r), _p(c), _p(e))
        return rp, r, c, e
    o = _stable_argsort(eids)
    o = o[_stable_argsort(rel[o])]
    return _ptrs_from_sorted(rel[o], num_rels), row[o], col[o], eids[o]


def _unique_per_relation(rel_ptrs: torch.Tensor, nodes: torch.Tensor, num_nodes: int):
    """Sorted unique node ids inside every relation bucket, concatenated.
    Returns (node_indices[U], rel_ptrs_u[R+1], inverse[len(nodes)])."""
    R = rel_ptrs.numel() - 1
    if nodes.is_cuda:
        return _unique_rel_nodes_native(rel_ptrs, nodes, None, num_nodes)
    rel_of = torch.repeat_interleave(
        torch.arange(R, dtype=_I64, device=nodes.device), rel_ptrs[1:] - rel_ptrs[:-1]
    )
    key = rel_of * int(num_nodes) + nodes
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    return uniq % int(num_nodes), _ptrs_from_sorted(uniq // int(num_nodes), R), inv


def _unique_rel_nodes_native(rel_ptrs, nodes_a, nodes_b, num_nodes: int):
    """(node_indices[U], rel_ptrs_u[R+1], inverse) through het_layout_unique_rel_nodes; with ``nodes_b`` the dual
    list over both arrays, inverse in the reference's per-relation [rows..., cols...] order."""
    rel_ptrs, nodes_a = _c64(rel_ptrs), _c64(nodes_a)
    nodes_b = None if nodes_b is None else _c64(nodes_b)
    R, E = rel_ptrs.numel() - 1, nodes_a.numel()
    total = E * (2 if nodes_b is not None else 1)
    out_nodes = torch.empty(max(total, 1), dtype=_I64, device=nodes_a.device)
    out_ptrs = torch.empty(R + 1, dtype=_I64, device=nodes_a.device)
    inv = torch.empty(total, dtype=_I64, device=nodes_a.device)
    count = _C.c_int64(0)
    _native("het_layout_unique_rel_nodes", nodes_a, _p(rel_ptrs), R, _p(nodes_a), _p(nodes_b), E, int(num_nodes),
            _p(out_nodes), _p(out_ptrs), _p(inv), _C.byref(count))
    return out_nodes[: count.value].clone(), out_ptrs, inv


class HetGraph:
    def __init__(self):
        self.graph_data: Dict = {}
        self.sequential_eids_format: Optional[str] = None
        self._num_nodes = 0
        self._num_rels = 0
        # opaque cache for device-side plans (het_amd.plan), keyed by name
        self._plans: Dict = {}

    # ---- construction -------------------------------------------------
    @classmethod
    def from_integrated_coo(cls, coo: IntegratedCOO, full: bool = True) -> "HetGraph":
        g = cls()
        g._num_nodes, g._num_rels = int(coo.num_nodes), int(coo.num_rels)
        g.graph_data["original"] = {
            "row_indices": coo.row,
            "col_indices": coo.col,
            "rel_types": coo.rel,
            "eids": coo.eids,
            "node_type_offsets": coo.node_type_offsets,
        }
        g.generate_separate_coo_adj_for_each_etype(transposed_flag=False)
        g.canonicalize_eids("separate_coo")
        if full:
            g.generate_separate_unique_node_indices_for_each_etype()
            g.generate_separate_unique_node_indices_single_sided_for_each_etype()
            g.generate_csrs()
        return g

    # ---- sizes ----------------------------------------------------------
    def get_num_nodes(self) -> int:
        return self._num_nodes

    def get_num_rels(self) -> int:
        return self._num_rels

    def get_num_edges(self) -> int:
        return int(self.graph_data["original"]["eids"].numel())

    def get_num_ntypes(self) -> int:
        return int(self.graph_data["original"]["node_type_offsets"].numel() - 1)

    def get_device(self):
        return self.graph_data["original"]["eids"].device

    # ---- device movement ------------------------------------------------
    def apply_to_each_tensor(self, func):
        def rec(d):
            for k, v in d.items():
                if isinstance(v, dict):
                    rec(v)
                elif isinstance(v, torch.Tensor):
                    d[k] = func(v)

        rec(self.graph_data)
        self._plans.clear()
        return self

    def to_(self, device):
        return self.apply_to_each_tensor(lambda t: t.to(device))

    def cuda_(self):
        return self.to_("cuda")

    def cpu_(self):
        return self.to_("cpu")

    def contiguous_(self):
        return self.apply_to_each_tensor(lambda t: t.contiguous())

    def __getitem__(self, key):
        return self.graph_data[key]

    def __contains__(self, key):
        return key in self.graph_data

    def save_to_disk(self, filename):
        torch.save(
            {"graph_data": self.graph_data, "num_nodes": self._num_nodes, "num_rels": self._num_rels,
             "sequential_eids_format": self.sequential_eids_format},
            filename,
        )

    def load_from_disk(self, filename):
        blob = torch.load(filename)
        self.graph_data = blob["graph_data"]
        self._num_nodes, self._num_rels = blob["num_nodes"], blob["num_rels"]
        self.sequential_eids_format = blob["sequential_eids_format"]
        self._plans.clear()
        return self

    # ---- builders -----------------------------------------------------------
    @torch.no_grad()
    def generate_separate_coo_adj_for_each_etype(self, transposed_flag: bool = False, rel_eid_sorted_flag: bool = True):
        if transposed_flag:
            raise NotImplementedError("only the original orientation is used by the hot path")
        o = self.graph_data["original"]
        rp, r, c, e = integrated_coo_to_separate_coo(
            o["row_indices"], o["col_indices"], o["rel_types"], o["eids"], self._num_rels
        )
        self.graph_data.setdefault("separate", {}).setdefault("coo", {})["original"] = {
            "rel_ptrs": rp, "row_indices": r, "col_indices": c, "eids": e,
        }

    @torch.no_grad()
    def canonicalize_eids(self, target_sequential_eids_format: str = "separate_coo"):
        """Renumber eids so that the separate COO's are arange(E): edge data is
        then stored in separate-COO order (hrt/python/utils/mydgl_graph.py:765-823)."""
        if target_sequential_eids_format == self.sequential_eids_format:
            return
        if target_sequential_eids_format != "separate_coo":
            raise NotImplementedError(target_sequential_eids_format)
        old = self.graph_data["separate"]["coo"]["original"]["eids"]
        mapping = torch.empty(int(old.max().item()) + 1 if old.numel() else 0, dtype=_I64, device=old.device)
        mapping[old] = torch.arange(old.numel(), dtype=_I64, device=old.device)

        def remap(d):
            if "eids" in d:
                d["eids"] = mapping[d["eids"]]

        remap(self.graph_data["separate"]["coo"]["original"])
        remap(self.graph_data["original"])
        if "transposed" in self.graph_data:
            remap(self.graph_data["transposed"])
        self.sequential_eids_format = target_sequential_eids_format
        self._plans.clear()

    @torch.no_grad()
    def generate_separate_unique_node_indices_for_each_etype(self, produce_inverse_idx: bool = True):
        if produce_inverse_idx:
            self.canonicalize_eids("separate_coo")
        s = self.graph_data["separate"]["coo"]["original"]
        E = s["row_indices"].numel()
        # per relation: unique(concat(rows, cols))  (mydgl_graph_methods.py:104-157)
        R = self._num_rels
        if s["row_indices"].is_cuda:
            nodes, ptrs, inv = _unique_rel_nodes_native(s["rel_ptrs"], s["row_indices"], s["col_indices"], self._num_nodes)
            d = {"node_indices": nodes, "rel_ptrs": ptrs}
            if produce_inverse_idx:
                d["inverse_indices"] = inv
            self.graph_data["separate"]["unique_node_indices"] = d
            return
        rel_of = torch.repeat_interleave(torch.arange(R, dtype=_I64, device=s["rel_ptrs"].device),
                                         s["rel_ptrs"][1:] - s["rel_ptrs"][:-1])
        key = torch.cat([rel_of * self._num_nodes + s["row_indices"], rel_of * self._num_nodes + s["col_indices"]])
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        d = {"node_indices": uniq % self._num_nodes, "rel_ptrs": _ptrs_from_sorted(uniq // self._num_nodes, R)}
        if produce_inverse_idx:
            # reference order: per relation [rows of r ..., cols of r ...]
            inv_row, inv_col = inv[:E], inv[E:]
            parts = []
            rp = s["rel_ptrs"].tolist()
            for r in range(R):
                parts += [inv_row[rp[r]:rp[r + 1]], inv_col[rp[r]:rp[r + 1]]]
            d["inverse_indices"] = torch.cat(parts) if parts else inv
        self.graph_data["separate"]["unique_node_indices"] = d

    @torch.no_grad()
    def generate_separate_unique_node_indices_single_sided_for_each_etype(self, produce_inverse_idx: bool = True):
        if produce_inverse_idx:
            self.canonicalize_eids("separate_coo")
        s = self.graph_data["separate"]["coo"]["original"]
        nr, pr, ir = _unique_per_relation(s["rel_ptrs"], s["row_indices"], self._num_nodes)
        nc, pc, ic = _unique_per_relation(s["rel_ptrs"], s["col_indices"], self._num_nodes)
        d = {"node_indices_row": nr, "rel_ptrs_row": pr, "node_indices_col": nc, "rel_ptrs_col": pc}
        if produce_inverse_idx:
            d["inverse_indices_row"], d["inverse_indices_col"] = ir, ic
        self.graph_data["separate"]["unique_node_indices_single_sided"] = d

    @torch.no_grad()
    def generate_csrs(self):
        """Integrated out-CSR (rows = src) under "original" and a true in-CSR
        (rows = dst, col_indices = src) under "transposed"."""
        o = self.graph_data["original"]
        ptrs, c, r, e = coo_to_csr(o["row_indices"], o["col_indices"], o["rel_types"], o["eids"], self._num_nodes)
        self.graph_data["out_csr"] = {"row_ptrs": ptrs, "col_indices": c, "rel_types": r, "eids": e}
        ptrs, c, r, e = coo_to_csr(o["col_indices"], o["row_indices"], o["rel_types"], o["eids"], self._num_nodes)
        self.graph_data["transposed"] = {"row_ptrs": ptrs, "col_indices": c, "rel_types": r, "eids": e}

    # ---- getters (names as in the reference) --------------------------------------
    def get_original_coo(self):
        o = self.graph_data["original"]
        return {k: o[k] for k in ("rel_types", "row_indices", "col_indices", "eids")}

    def get_out_csr(self):
        if "out_csr" not in self.graph_data:
            self.generate_csrs()
        return dict(self.graph_data["out_csr"])

    def get_in_csr(self):
        if "transposed" not in self.graph_data:
            self.generate_csrs()
        return dict(self.graph_data["transposed"])

    def get_original_node_type_offsets(self):
        return self.graph_data["original"]["node_type_offsets"]

    def get_rel_node_types(self):
        """(src_type [R], dst_type [R]) int64: the node type of the sources / destinations of every relation, for graphs
        whose relations are canonical edge types (one source type, one destination type each) -- what the reference's HGT
        receives as src_ / dst_node_type_per_canonical_edge_type (hrt/python/HGT/models.py:31-52).  Read off the first
        edge of each relation (an empty relation gets type 0); raises when a relation mixes node types."""
        hit = self._plans.get("rel_node_types")
        if hit is None:
            s = self.graph_data["separate"]["coo"]["original"]
            offs = self.graph_data["original"]["node_type_offsets"]
            rp = s["rel_ptrs"]
            first = rp[:-1].clamp(max=max(0, s["row_indices"].numel() - 1))
            empty = rp[1:] == rp[:-1]
            run = lambda nodes: torch.searchsorted(offs[1:].contiguous(), nodes, right=True).clamp(max=offs.numel() - 2)
            seg_types = self.graph_data["original"].get("node_segment_types")  # sampled block: runs of equal type (sampling.py)
            typ = run if seg_types is None else (lambda nodes: seg_types[run(nodes)])
            E = s["row_indices"].numel()
            if E == 0:
                z = torch.zeros(rp.numel() - 1, dtype=_I64, device=rp.device)
                hit = (z, z.clone())
            else:
                st, dt = typ(s["row_indices"][first]), typ(s["col_indices"][first])
                st, dt = torch.where(empty, torch.zeros_like(st), st), torch.where(empty, torch.zeros_like(dt), dt)
                rel = torch.repeat_interleave(torch.arange(rp.numel() - 1, device=rp.device), rp[1:] - rp[:-1], output_size=E)
                if not (bool((typ(s["row_indices"]) == st[rel]).all()) and bool((typ(s["col_indices"]) == dt[rel]).all())):
                    raise ValueError("get_rel_node_types: a relation mixes node types (not a canonical edge type)")
                hit = (st.contiguous(), dt.contiguous())
            self._plans["rel_node_types"] = hit
        return hit

    def get_separate_coo_original(self):
        return dict(self.graph_data["separate"]["coo"]["original"])

    def get_separate_unique_node_indices(self):
        d = self.graph_data["separate"]["unique_node_indices"]
        return {"rel_ptrs": d["rel_ptrs"], "node_indices": d["node_indices"]}

    def get_separate_unique_node_indices_inverse_idx(self):
        d = self.graph_data["separate"]["unique_node_indices"]
        return {"rel_ptrs": d["rel_ptrs"], "inverse_indices": d["inverse_indices"]}

    def get_separate_unique_node_indices_single_sided(self):
        d = self.graph_data["separate"]["unique_node_indices_single_sided"]
        return {k: d[k] for k in ("node_indices_row", "rel_ptrs_row", "node_indices_col", "rel_ptrs_col")}

    def get_separate_unique_node_indices_single_sided_inverse_idx(self):
        d = self.graph_data["separate"]["unique_node_indices_single_sided"]
        return {k: d[k] for k in ("rel_ptrs_row", "inverse_indices_row", "inverse_indices_col")}


MyDGLGraph = HetGraph  # the reference's class name
