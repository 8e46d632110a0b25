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

All builders are vectorised torch code and run on whatever device the tensors
live on (the reference's run on CPU with Python loops).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .synth import IntegratedCOO

_I64 = torch.int64


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
    rows = csr_to_coo_rows(row_ptrs)
    n = max(int(row_ptrs.numel() - 1), int(col_indices.max().item()) + 1 if col_indices.numel() else 0)
    ptrs, c, r, e = coo_to_csr(col_indices, rows, rel_types, eids, n)
    return ptrs, c, e, r


def integrated_coo_to_separate_coo(row, col, rel, eids, num_rels: int):
    """Bucket edges by relation, each bucket sorted by eid (reference:
    ``convert_integrated_coo_to_separate_coo`` hrt/include/MyHyb/MyHyb.h:1047-1096
    followed by ``sort_coo_by_etype_eids_torch_tensors``)."""
    o = _stable_argsort(eids)
    o = o[_stable_argsort(rel[o])]
    return _ptrs_from_sorted(rel[o], num_rels), row[o], col[o], eids[o]


def _unique_per_relation(rel_ptrs: torch.Tensor, nodes: torch.Tensor, num_nodes: int):
    """Sorted unique node ids inside every relation bucket, concatenated.
    Returns (node_indices[U], rel_ptrs_u[R+1], inverse[len(nodes)])."""
    R = rel_ptrs.numel() - 1
    rel_of = torch.repeat_interleave(
        torch.arange(R, dtype=_I64, device=nodes.device), rel_ptrs[1:] - rel_ptrs[:-1]
    )
    key = rel_of * int(num_nodes) + nodes
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    return uniq % int(num_nodes), _ptrs_from_sorted(uniq // int(num_nodes), R), inv


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
