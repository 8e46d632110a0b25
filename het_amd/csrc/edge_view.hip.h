// A position-indexed view of an edge list that covers both layouts the ops take:
// separate COO (src/dst arrays, relation by searching rel_ptrs) and integrated
// CSR (one endpoint by searching row_ptrs, relation given per edge).
#pragma once
#include "common.hip.h"

struct EdgeView {
  idx_t E = 0;
  int64_t N = 0;
  const idx_t* eids = nullptr;
  const idx_t* src = nullptr;       // per position, or NULL -> rows of src_ptrs
  const idx_t* dst = nullptr;       // per position, or NULL -> rows of dst_ptrs
  const idx_t* src_ptrs = nullptr;  // [N+1] CSR over sources
  const idx_t* dst_ptrs = nullptr;  // [N+1] CSR over destinations
  const idx_t* rel_ptrs = nullptr;  // [R+1] relation buckets over positions, or
  const idx_t* rel_types = nullptr; // per position
  int R = 0;
};

__device__ __forceinline__ idx_t ev_src(const EdgeView& v, idx_t i) {
  return v.src ? v.src[i] : (idx_t)find_segment(v.src_ptrs, (int)v.N, i);
}
__device__ __forceinline__ idx_t ev_dst(const EdgeView& v, idx_t i) {
  return v.dst ? v.dst[i] : (idx_t)find_segment(v.dst_ptrs, (int)v.N, i);
}
__device__ __forceinline__ int ev_rel(const EdgeView& v, idx_t i) {
  return v.rel_types ? (int)v.rel_types[i] : find_segment(v.rel_ptrs, v.R, i);
}

// Row maps of the compact (relation, node) tensors on the source and destination side.
struct RowMaps {
  int kind = 0;
  const idx_t *ra = nullptr, *rb = nullptr, *ca = nullptr, *cb = nullptr;
};

__device__ __forceinline__ void ev_rows(const EdgeView& v, const RowMaps& m, idx_t i, idx_t eid, idx_t s, idx_t d,
                                        idx_t& srow, idx_t& drow) {
  if (m.kind == HET_KIND_DISABLED) {
    srow = drow = eid;
  } else if (m.kind == HET_KIND_DUAL_LIST_DIRECT_INDEX) {
    srow = m.ra[eid];
    drow = m.ca[eid];
  } else {
    const int r = ev_rel(v, i);
    srow = compact_row(HET_KIND_ENABLED, m.ra, m.rb, r, s, eid);
    drow = compact_row(HET_KIND_ENABLED, m.ca, m.cb, r, d, eid);
  }
}
