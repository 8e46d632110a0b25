// het_grouping: positions of an index list sorted by (relation, key), with the
// segments (runs of equal (relation, key)) and a list of wave-sized work items.
#pragma once
#include <vector>

#include "common.hip.h"

// Segments longer than this are split into several work items so that no single
// wave serialises a hub node (degree skew: a few destinations own 1e5+ edges).
constexpr int HET_ITEM_MAX = 256;

struct het_grouping {
  uint64_t serial = 0;     // unique per object for the life of the process (an address can come back after a destroy)
  // Streams (round 5).  The arrays below come from the caller's allocator when one is installed (het_set_allocator): a stream-
  // ordered pool hands a freed block out again at once to its creation stream, without waiting for kernels of OTHER streams that
  // still read it.  `home` is the stream the grouping was built on (its blocks belong to it); `used` the other streams a binding
  // reported through het_grouping_note_stream.  het_grouping_destroy makes `home` wait for them before it releases anything.
  hipStream_t home = nullptr;
  mutable std::vector<hipStream_t> used;
  int64_t E = 0;           // positions
  int64_t S = 0;           // segments
  int64_t num_items = 0;   // work items (>= S)
  int64_t num_split = 0;   // segments that were split into > 1 item
  int64_t key_bound = 0;
  int R = 0;               // 0: grouped by key only
  int32_t* perm = nullptr;       // [E]   position at sorted rank j (stable)
  int32_t* seg_ptr = nullptr;    // [S+1] sorted-rank range of segment s
  int32_t* seg_key = nullptr;    // [S]   key of segment s
  int32_t* seg_rel_ptr = nullptr;// [R+1] segment range of relation r (R > 0)
  idx_t* seg_key64 = nullptr;    // [S]   seg_key as int64 (gather / scatter list of the segment GEMMs)
  idx_t* seg_rel_ptr64 = nullptr;// [R+1] seg_rel_ptr as int64
  int32_t* item_seg = nullptr;   // [num_items] segment of the item
  int32_t* item_begin = nullptr; // [num_items+1] sorted-rank range [item_begin[t], item_end[t])
  int32_t* item_end = nullptr;
  int32_t* split_seg = nullptr;  // [num_split] segments that own several items
  int32_t* p0 = nullptr;         // [E] payload0[perm[j]] or NULL
  int32_t* p1 = nullptr;         // [E] payload1[perm[j]] or NULL
  int64_t p0_max = 0, p1_max = 0;  // largest payload value (0 without the payload): bounds the tables a kernel indexes with them
  bool p0_contiguous = false;    // p0[j] == j for every rank: the list already was in (relation, key) order
  mutable int32_t* seg_of_rank = nullptr;  // [E] segment of sorted rank j; built on first use (segment broadcast)
  // "Packs" (grouping_packs): the SHORT segments (<= HET_PACK_T positions) gathered into runs of whole segments of about
  // HET_PACK_T consecutive ranks (< 2 * HET_PACK_T) -- the work unit of the lane-group-per-pack kernels; a few edges per
  // (relation, source) row is the common case, and a work unit per segment would spend its time in dependent
  // prologues.  A longer segment is a pack of its own with bit 31 of pack_ptr set: the pack kernels skip it and the
  // wave-per-item kernels take its work items (long_items: indices into item_seg / item_begin / item_end).
  mutable int32_t* pack_ptr = nullptr;     // [num_packs+1] first rank of pack k (| 1u<<31: a long segment)
  mutable int32_t* key_of_rank = nullptr;  // [E+1] seg_key of the segment of rank j; sentinel -1 at E
  mutable int32_t* long_items = nullptr;   // [num_long_items] work items of segments with more than HET_PACK_T positions
  mutable int64_t num_packs = 0, num_long_items = 0;
  // A second set of packs for ONE other threshold (grouping_pack_view: the RGAT backward walks segments of up to 64 positions in
  // packs), kept beside the default set, so that the packs an op gets do not depend on which op touched the grouping first
  // (until round 5 the first user's threshold decided for everybody, and with it the fp32 summation order).
  mutable int32_t* alt_pack_ptr = nullptr;
  mutable int32_t* alt_long_items = nullptr;
  mutable int64_t alt_num_packs = 0, alt_num_long_items = 0;
  mutable int alt_pack_t = 0;              // 0: not built
  // Packed ids (grouping_packed_ids): one vector load per edge instead of one per list -- the gather passes are bound by the
  // number of their vector-memory instructions.
  mutable int2* p01 = nullptr;    // [E]   {p0[j], p1[j]}
  mutable int4* kp01 = nullptr;   // [E+1] {key_of_rank[j], p0[j], p1[j], tag}; sentinel key -1 at E (needs the packs)
  // tag (grouping_tag_kp01; 0 until a user asks): what a lane group that walks the ranks of a pack in order would otherwise derive
  // from compares against the neighbouring records and a search of relation boundaries, per edge --
  //   HET_TAG_FIRST_KEY / _LAST_KEY  first / last rank of its segment;  HET_TAG_LAST_RUN  last rank of a run of equal p1 inside it
  //   tag >> HET_TAG_REL_SHIFT  relation of the rank: number of relation boundaries <= its key (tag_which 0) or its p0 (tag_which 1)
  mutable int tag_which = -1;     // -1: not tagged
  mutable int tag_thr[7] = {0, 0, 0, 0, 0, 0, 0};
  mutable const idx_t* tag_dev_src = nullptr;  // grouping_tag_kp01_dev: the device array the thresholds were read from, its length - 1
  mutable int tag_dev_R = 0;
  // Hub items (grouping_hub_items): for a grouping by key * R + relation, the work items (ascending) whose key belongs to a
  // segment of more than hub_min positions in the twin grouping by key alone (het_rgat_aggregate_compact_runs).
  mutable int32_t* hub_items = nullptr;
  mutable int64_t num_hub_items = -1;         // -1: not built
  mutable int32_t* hub_order = nullptr;       // [num_hub_items] indices into hub_items, by the first payload0 of the twin in the item
  mutable int32_t* hub_segs = nullptr;        // the twin's segments of more than hub_min positions (ascending)
  mutable int4* hub_rec = nullptr;            // [num_hub_segs] {first run (segment of this grouping), one past the last, first record in
                                              // hub_items, key of the hub}: what a finishing pass needs per hub, searched ONCE here
                                              // instead of by five dependent binary searches per hub and launch (round 5)
  mutable int64_t num_hub_segs = 0;
  mutable int hub_min = 0;
  mutable uint64_t hub_twin_serial = 0;       // serial of the twin the lists were built against
  mutable std::vector<int32_t*> retired;      // hub lists replaced by a rebuild (grouping_hub_items): freed with the grouping
  // An order of the rows of a caller's list by the list's values (grouping_value_order): ONLY a locality hint -- any permutation of
  // [0, n) gives the same results -- so it is cached by the identity (array, length) of the list it was built from.
  mutable int32_t* val_order = nullptr;
  mutable const idx_t* val_order_src = nullptr;
  mutable int64_t val_order_n = 0;
};

constexpr int HET_PACK_T = 32;
// Builds g->pack_ptr / key_of_rank / long_items (threshold HET_PACK_T) once (thread-safe; synchronises `s` before publishing them).
int grouping_packs(const het_grouping* g, hipStream_t s);
// The packs of a grouping for a threshold: HET_PACK_T (or <= 0) = the default set above; any other value = the second set
// (built on first use; a grouping keeps one other threshold -- asking for a third retires the second).  key_of_rank is shared.
struct PackView {
  const int32_t* pack_ptr = nullptr;
  const int32_t* long_items = nullptr;
  int64_t num_packs = 0, num_long_items = 0;
};
int grouping_pack_view(const het_grouping* g, hipStream_t s, int pack_t, PackView* out);
// Builds g->p01 (with_keys == false) or g->kp01 (true; builds the packs first) once, thread-safe, published after a sync.
int grouping_packed_ids(const het_grouping* g, bool with_keys, hipStream_t s);
constexpr int HET_TAG_FIRST_KEY = 1, HET_TAG_LAST_RUN = 2, HET_TAG_LAST_KEY = 4, HET_TAG_REL_SHIFT = 8;
// Fills the fourth word of g->kp01 (builds kp01 first; thread-safe; synchronises `s`).  thr[0..6]: ascending first values of relations
// 1 .. 7 (INT_MAX beyond the last relation), or NULL: the caller reads the segment / run flags only (any tagging will do);
// which = 0: the relation follows the key, 1: payload0.  A grouping tagged with other thresholds before is re-tagged in place (the
// flags are rewritten with the same values; the thresholds of a graph's unique lists do not change between calls).
int grouping_tag_kp01(const het_grouping* g, int which, const int* thr, hipStream_t s);
// The same with the relation boundaries read on the device: rel_ptrs_dev [R+1] (int64), any R < 2^23 (a search above 8).  No host copy,
// so "tagged already" is decided by the identity of (array, R): the boundaries of a row list belong to the list the grouping sorts,
// a caller does not hand the same grouping different boundaries.
int grouping_tag_kp01_dev(const het_grouping* g, int which, const idx_t* rel_ptrs_dev, int R, hipStream_t s);
// Builds g_rel->hub_items once (thread-safe, published after a sync): g_rel groups the same positions as `twin` by
// key * R + relation.  Rebuilt when g_rel was paired with another twin object (or threshold) before.
int grouping_hub_items(const het_grouping* g_rel, const het_grouping* twin, int R, int hub_min, hipStream_t s);
// g->val_order [n]: the indices of values[0..n) (device, 0 <= values < 2^31) in ascending order of the value, ties in index order;
// built once per (array, n), thread-safe, published after a sync.  A pass over the rows of a (relation, node) list in this order
// visits the rows of one node together (het_rgat_backward_compact_runs: the two rows of a destination read the same gradout /
// ret rows).
int grouping_value_order(const het_grouping* g, const idx_t* values, int64_t n, hipStream_t s);
