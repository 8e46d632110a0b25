// Segment GEMM on the matrix cores: C[cs(i), :] (+)= scale(i) * A[ga(i), :] . B_r
//
// Workgroup = 4 waves; the relation's K x X weight is staged once in LDS and
// reused for a chunk of up to 512 rows (4 steps of 128 rows, 32 rows per wave).
// Per wave and step: 32 gathered rows are loaded coalesced (K/4 lanes x float4
// per row) into a padded LDS tile, read back as MFMA A fragments with
// ds_read_b128, and multiplied with v_mfma_f32_32x32x2_f32 into X/32
// accumulators.  The k index is permuted between the two lane halves
// (half h owns k in [h*K/2, (h+1)*K/2)) -- A and B use the same permutation, so
// the sum is unchanged while every lane's fragment is one contiguous half row.
#include <stdlib.h>

#include "seg_gemm_mfma.hip.h"
#include "coop.hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChunkRows = 2048;  // rows of one relation per workgroup at most (16 tiles per wave)
// Smaller inputs (a rank's share of a partitioned graph, a sampled block) get smaller chunks so that the launch still
// has ~2000 workgroups: 128 rows = one 32-row tile per wave is the floor.
inline int chunk_rows_for(int64_t num_rows) {
  static const int64_t wgs = [] { const char* v = getenv("HET_GEMM_WGS"); return v && atoi(v) > 0 ? (int64_t)atoi(v) : 2048; }();  // A/B switch
  int64_t c = ((num_rows / wgs) + 127) / 128 * 128;
  return (int)(c < 128 ? 128 : (c > kChunkRows ? kChunkRows : c));
}

// DOT (plain stores only): additionally dot_out[cs(i), h] = < C row (h, :), dot_w[r, h, :] > from the row pieces the
// epilogue already holds -- the attention-vector product of RGAT without re-reading the tensor just written.
// RMW (plain stores only): C row += the product, read-modify-write WITHOUT atomics -- for launches whose rows hit
// distinct C rows (one relation of a unique (relation, node) list): 256-byte rows added at the plain load / store rate
// instead of the float-atomic rate (1.3 TB/s chip-wide).  The old C rows are requested before the MFMAs of the tile.
// DOTL: 0 = no dot; 4 = heads of 16 floats (4 lanes of the store mapping: the sum over a head is two DPP adds and the head index a
// shift -- the RGAT shape); -1 = any power-of-two head width (shuffles, runtime lane counts).
template <int K, int NT, bool ATOMIC, int DOTL = 0, bool RMW = false>
__device__ __forceinline__ void seg_gemm_mfma_body(const MfmaGemmArgs& a, int chunk_rows) {
  constexpr bool DOT = DOTL != 0;
  constexpr int X = NT * 32, KH = K / 2;
  constexpr int LDA = K + 4, LPRA = K / 4, RPIA = 64 / LPRA, NITA = 32 / RPIA;  // A tile: rows per load instr
  constexpr int LDC = X + 4, LPRC = X / 4, RPIC = 64 / LPRC, NITC = 32 / RPIC;  // C tile: rows per store instr
  constexpr int WREG = 32 * (LDA > LDC ? LDA : LDC);                            // per-wave LDS floats
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk_rows, blockIdx.x, r, rb, re)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // The relation's weight is staged once per workgroup in LDS ([K][X], read conflict-free: 32 consecutive
  // floats per lane half).
  // (keeping small weights in registers instead was measured slower: 201 VGPRs, 2 waves per SIMD, 2.22 ms vs
  //  2.08 ms with the weight in LDS at 3 waves per SIMD on the 64x64 forward projection of ogbn-mag)
  constexpr bool B_REGS = false;
  float* Bs = smem;                                           // [K][X] (unused when B_REGS)
  float* Ws = smem + (B_REGS ? 0 : K * X) + wave * WREG;      // wave-private: A tile [32][LDA], then C tile [32][LDC]
  const float* __restrict__ Bm = a.B + (int64_t)r * a.b_rel_stride;
  const int KF = a.b_kfull ? a.b_kfull : K, XF = a.b_xfull ? a.b_xfull : X;  // the weight's full size (see b_k0 / b_n0)
  auto b_elem = [&](int kw, int nw) -> float {
    const int k = kw + a.b_k0, n = nw + a.b_n0;
    if (a.b_headcat == 1) {
      const int Dh = a.headcat_d, h = n / Dh, d = n - h * Dh;
      return Bm[(int64_t)h * KF * Dh + (int64_t)k * Dh + d];
    }
    if (a.b_headcat == 2) {  // block diagonal: per-head [Kh x Dh] blocks, A and C rows are [H*Kh] / [H*Dh]
      const int Dh = a.headcat_d, Kh = a.blockdiag_k, hk = k / Kh, hn = n / Dh;
      return hk == hn ? Bm[((int64_t)hk * Kh + (k - hk * Kh)) * Dh + (n - hn * Dh)] : 0.f;
    }
    return Bm[(int64_t)k * XF + n];
  };
  float breg[B_REGS ? KH * NT : 1];
  if (B_REGS) {
#pragma unroll
    for (int s = 0; s < KH; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        breg[B_REGS ? s * NT + nt : 0] = b_elem((lane >> 5) * KH + s, nt * 32 + (lane & 31));
  } else {
    for (int e = tid; e < K * X; e += 256) Bs[e] = b_elem(e / X, e % X);
    __syncthreads();
  }
  // From here on the four waves are independent: each walks its own 32-row tiles with a private LDS
  // region (LDS instructions of one wave execute in order, so no barrier is needed).

  const int row = lane & 31, half = lane >> 5;
  const int ra = lane / LPRA, ca = (lane % LPRA) * 4;  // A-load mapping
  const int rc = lane / LPRC, cc = (lane % LPRC) * 4;  // C-store mapping
  float4 dotw = make_float4(0.f, 0.f, 0.f, 0.f);
  if (DOT) dotw = *reinterpret_cast<const float4*>(a.dot_w + (int64_t)r * X + cc);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!ATOMIC && a.bias) bias4 = *reinterpret_cast<const float4*>(a.bias + cc);
  const int dot_dl = DOTL == 4 ? 4 : (DOT ? a.headcat_d >> 2 : 1), dot_h = DOTL == 4 ? cc >> 4 : (DOT ? cc / a.headcat_d : 0);
  const int dot_H = DOTL == 4 ? X / 16 : (DOT ? X / a.headcat_d : 1);

  // Global loads are software-pipelined two tiles deep and issued in branch-free phases of independent
  // instructions (out-of-range rows clamp to the last row and are masked afterwards):
  //   iteration t:  wait rows(t) -> LDS;  issue rows(t+1) (their row ids arrived an iteration ago);
  //                 issue row ids of tile t+2 (A gather list and C scatter list);  MFMAs(t);  stores(t)
  // vmcnt counts loads and stores in issue order, so a wait for a load also waits for every older store:
  // with this order everything consumed in iteration t+1 was requested BEFORE the stores of tile t, and
  // no wait in the loop ever covers a store that was just issued.
  float4 areg[NITA];
  int ar_next[NITA];    // A row ids of the next tile (32-bit: row counts < 2^31 are checked by the callers)
  int crow_next[NITC];  // C row ids of the next tile, -1 = no row
  // No branch inside the loop (the waitcnt pass is conservative at block boundaries): a missing gather /
  // scatter list is read as seg_ptrs[0] (always valid) and the loaded value is discarded by a select.
  const bool has_g = a.gather != nullptr, has_s = a.scatter != nullptr;
  const idx_t* __restrict__ gp = has_g ? a.gather : a.seg_ptrs;
  const idx_t* __restrict__ sp = has_s ? a.scatter : a.seg_ptrs;
  // load_ids only ISSUES loads (raw 32-bit halves of the int64 ids, no arithmetic on them): the values are
  // first touched one iteration later, so no wait for them is placed in front of the MFMAs.
  auto load_ids = [&](idx_t wb, int (&ar)[NITA], int (&cr)[NITC]) {
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      const idx_t i = wb + it * RPIA + ra, ic = i < re ? i : re - 1;
      ar[it] = reinterpret_cast<const int*>(gp + (has_g ? ic : 0))[0];
    }
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      const idx_t i = wb + it * RPIC + rc, ic = i < re ? i : re - 1;
      cr[it] = reinterpret_cast<const int*>(sp + (has_s ? ic : 0))[0];
    }
  };
  // rows of the tile starting at wb, ids as loaded by load_ids for that tile
  auto load_rows = [&](idx_t wb, const int (&ar)[NITA]) {
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      const idx_t i = wb + it * RPIA + ra, ic = i < re ? i : re - 1;
#ifdef HET_ABL_NOLOAD
      const int64_t r64 = ra;
#else
      const int64_t r64 = has_g ? (int64_t)ar[it] : (int64_t)ic;
#endif
      areg[it] = *reinterpret_cast<const float4*>(a.A + r64 * a.a_ld + ca);
    }
  };

  // One tile: rows(t) -> LDS, request tile t+1 / ids of t+2, MFMAs, epilogue.  The body is instantiated twice
  // (first tile peeled, then the loop) so that both entries of the loop header see the same queue of pending
  // memory operations [ids, rows, stores] and the compiler's counted waits stay exact.
  int crow_cur[NITC];
  auto tile = [&](idx_t wb) {
    // Rows past the end of the segment (last tile only) are clamped to its last row everywhere, so that all
    // lanes run the same unpredicated code (stores the compiler can count): when storing they recompute and
    // rewrite that row's values (same bytes), when accumulating they add zeros to it.
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
      const bool in = !ATOMIC || wb + it * RPIA + ra < re;
      *reinterpret_cast<float4*>(&Ws[(it * RPIA + ra) * LDA + ca]) = in ? areg[it] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int crow[NITC];
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      const idx_t i = wb + it * RPIC + rc, ic = i < re ? i : re - 1;
      crow[it] = has_s ? crow_cur[it] : (int)ic;
    }
    // (ids beyond the segment clamp to its last row: always valid addresses, results never used)
    int ar_cur[NITA];
#pragma unroll
    for (int it = 0; it < NITA; ++it) ar_cur[it] = ar_next[it];
#pragma unroll
    for (int it = 0; it < NITC; ++it) crow_cur[it] = crow_next[it];
    load_ids(wb + 256, ar_next, crow_next);
    load_rows(wb + 128, ar_cur);
    float4 cold[RMW ? NITC : 1];
    if (RMW) {
#pragma unroll
      for (int it = 0; it < NITC; ++it) cold[RMW ? it : 0] = *reinterpret_cast<const float4*>(a.C + (int64_t)crow[it] * a.c_ld + cc);
    }
    __builtin_amdgcn_sched_barrier(0);  // the prefetch stays above the MFMAs

    float af[KH];
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(&Ws[row * LDA + half * KH + q * 4]);
      af[4 * q + 0] = t.x; af[4 * q + 1] = t.y; af[4 * q + 2] = t.z; af[4 * q + 3] = t.w;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    // PAIRED (two 32-wide column tiles, plain stores): a lane reads the weight columns (2*row, 2*row + 1) of k-step s with
    // ONE ds_read_b64 instead of two ds_read_b32 -- tile nt then holds the columns 2*n + nt, a relabelling the LDS transpose
    // of the epilogue undoes (64 instead of 128 LDS reads per 32-row tile; the atomic epilogue keeps the contiguous
    // mapping its 128-byte-segment adds need).
    constexpr bool PAIRED = NT == 2 && !ATOMIC && !B_REGS;
#pragma unroll
    for (int s = 0; s < KH; ++s) {
      if (PAIRED) {
        const float2 b2 = *reinterpret_cast<const float2*>(&Bs[(half * KH + s) * X + 2 * row]);
#ifdef HET_ABL_NOMFMA
        acc[0][s & 15] += af[s] * b2.x; acc[NT - 1][s & 15] += af[s] * b2.y;
#else
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b2.x, acc[0], 0, 0, 0);
        acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b2.y, acc[NT - 1], 0, 0, 0);
#endif
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float b = B_REGS ? breg[B_REGS ? s * NT + nt : 0] : Bs[(half * KH + s) * X + nt * 32 + row];
#ifdef HET_ABL_NOMFMA  // diagnostic builds only (exp/mfma_bench.hip)
          acc[nt][s & 15] += af[s] * b;
#else
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b, acc[nt], 0, 0, 0);
#endif
        }
      }
    }
    // Epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    if (ATOMIC) {
      // straight from the accumulators: one atomic instruction adds two 128-byte row segments, the shape
      // float atomics run at full rate with (MI355X_MICROARCH.md, Global float atomics); C-row ids go through
      // the wave's LDS region (all lanes of a store row group hold the same id)
#pragma unroll
      for (int it = 0; it < NITC; ++it) reinterpret_cast<int*>(Ws)[it * RPIC + rc] = crow[it];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int cr = reinterpret_cast<const int*>(Ws)[(reg & 3) + 8 * (reg >> 2) + 4 * half];
        float* p = a.C + (int64_t)cr * a.c_ld + row;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) atomicAdd(p + nt * 32, acc[nt][reg]);
      }
    } else {
      // transpose through the wave's LDS region so that every output row leaves as whole 16-byte pieces
      // (X/4 lanes x float4 per row) instead of 32 row-strided dword stores
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          Ws[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LDC + (PAIRED ? 2 * row + nt : nt * 32 + row)] = acc[nt][reg];
#pragma unroll
      for (int it = 0; it < NITC; ++it) {
        float4 v = *reinterpret_cast<const float4*>(&Ws[(it * RPIC + rc) * LDC + cc]);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        bool real = true;
        if (RMW) {
          // rows past the end of the segment are clamped copies of its last row: only the real one adds and stores (a
          // copy storing anything could land after the real row's store)
          real = wb + it * RPIC + rc < re;
          const float4 c = cold[RMW ? it : 0];
          v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
        }
#ifdef HET_ABL_NOSTORE
        if (a.num_rows < 0)
#endif
        if (!RMW || real) *reinterpret_cast<float4*>(a.C + (int64_t)crow[it] * a.c_ld + cc) = v;
        if (DOT) {
          float p = v.x * dotw.x + v.y * dotw.y + v.z * dotw.z + v.w * dotw.w;
          if (DOTL == 4) p = quad_sum(p);  // (coop.hip.h: the four lanes of a head are a DPP quad)
          else for (int off = dot_dl >> 1; off > 0; off >>= 1) p += __shfl_xor(p, off);
          // all lanes of a head store the same value to the same word (no lane predicate: see the store note above).
          // Measured alternatives, both no faster: constant-offset butterflies; staging the tile's dots in LDS and
          // storing them with one or two instructions per tile (4.8 -> 5.0 ms per two launches).
          a.dot_out[(int64_t)crow[it] * dot_H + dot_h] = p;
        }
      }
    }
  };

  idx_t wb = rb + wave * 32;
  if (wb >= re) return;
  load_ids(wb, ar_next, crow_cur);
  {
    int ar0[NITA];
#pragma unroll
    for (int it = 0; it < NITA; ++it) ar0[it] = ar_next[it];
    load_ids(wb + 128, ar_next, crow_next);
    load_rows(wb, ar0);
  }
  tile(wb);
  for (wb += 128; wb < re; wb += 128) tile(wb);
}

template <int K, int NT, bool ATOMIC, int DOTL = 0, bool RMW = false>
__global__ __launch_bounds__(256) void HET_seg_gemm_mfma(MfmaGemmArgs a, int chunk_rows) {
  seg_gemm_mfma_body<K, NT, ATOMIC, DOTL, RMW>(a, chunk_rows);
}
// The RGAT projection (K <= 64 into 64 columns, heads of 16, dot epilogue) held to the register budget of three waves per SIMD --
// what the plain-store instance needs anyway (168); without the bound the dot's few extra values cost a whole wave per SIMD (180).
template <int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void HET_seg_gemm_mfma_dot16(MfmaGemmArgs a, int chunk_rows) {
  seg_gemm_mfma_body<K, 2, false, 4, false>(a, chunk_rows);
}

// ---- weight gradient -----------------------------------------------------------------------------
// One workgroup = 4 waves owns a chunk of rows of one relation; every wave walks a quarter of the
// chunk two rows per MFMA step (lane half h reads row 2s+h: 32 consecutive floats of the A row as
// the MFMA A operand -- A is used transposed, feature index on the M axis -- and 32 consecutive
// floats of the G row as the B operand), accumulating the full K x X product in KT*NT 32x32
// accumulators.  The four partial products are summed through LDS and flushed with one atomic
// add per element and workgroup.
// CS: additionally colsum[n] += SUM_rows G[row, n] (the bias gradient of a layer whose output gradient this launch streams
// anyway: one VALU add per loaded G value instead of another pass over the rows); the workgroups of the first K block add it.
template <int KT, int NT, bool CS = false>
__global__ __launch_bounds__(256) void HET_seg_dw_mfma(MfmaDwArgs a, int chunk) {
  // blockIdx.y selects a (KT*32) x (NT*32) block of the K x X product when K or X exceed 64 (each block re-reads its
  // column slices of the A and G rows)
  const int Kf = a.K, Xf = a.X, nbn = Xf / (NT * 32);
  const int kbase = ((int)blockIdx.y / nbn) * KT * 32, nbase = ((int)blockIdx.y % nbn) * NT * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][KT*NT*16][64] partials of waves 1..3
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, chunk, blockIdx.x, r, rb, re)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const idx_t n = re - rb, q = ((n + 3) / 4 + 1) & ~(idx_t)1;  // rows per wave, even
  const idx_t wb = rb + wave * q, we = (wb + q < re) ? wb + q : re;
  f32x16 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[kt][nt][e] = 0.f;
  float cs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) cs[nt] = 0.f;
  // Rows are consumed in batches of SB two-row MFMA steps with a two-deep, branch-free software pipeline (the same
  // scheme as the GEMM kernel above): iteration b issues the row loads of batch b+1 (their ids arrived one batch
  // ago) and the id loads of batch b+2, then runs the MFMAs of batch b.  Out-of-range rows clamp to the last row
  // (valid addresses) and enter the product as zeros.
  if (wb < we) {
    constexpr int SB = 4;  // same-box A/B: 8 -> 0.353 ms, 4 -> 0.306 ms per launch (fewer registers, 3 -> 4 waves per SIMD)
    const bool has_g = a.gather != nullptr, has_gg = a.g_gather != nullptr;
    const idx_t* __restrict__ gp = has_g ? a.gather : a.seg_ptrs;
    const idx_t* __restrict__ ggp = has_gg ? a.g_gather : a.seg_ptrs;
    int ar[2][SB], gr[2][SB];
    float av[2][SB][KT], gv[2][SB][NT];
    auto load_ids = [&](idx_t base, int (&A)[SB], int (&G)[SB]) {
#pragma unroll
      for (int st = 0; st < SB; ++st) {
        const idx_t i = base + 2 * st + half, ic = i < we ? i : we - 1;
        // the raw loaded words: the choice between them and the row's own index is made where the row is fetched (one
        // batch later) -- a select here would make the wave wait for the id loads it has just issued, and, the counter
        // being in-order, for every row load before them (measured: 0.305 -> see DESIGN.md section 4.1)
        A[st] = reinterpret_cast<const int*>(gp + (has_g ? ic : 0))[0];
        G[st] = reinterpret_cast<const int*>(ggp + (has_gg ? ic : 0))[0];
      }
    };
    // Operand loads: with two 32-wide tiles along K (or X) a lane fetches the float PAIR (2*col, 2*col + 1) of its row with
    // one 8-byte load -- tile t then holds the features 2*m + t instead of t*32 + m, a relabelling of the product's rows
    // (columns) that the epilogue undoes.  Half the vector-memory instructions of one dword per tile (the kernel was
    // bound by their number).
    auto load_rows = [&](idx_t base, const int (&Ai)[SB], const int (&Gi)[SB], float (&AV)[SB][KT], float (&GV)[SB][NT]) {
      int A[SB], G[SB];
#pragma unroll
      for (int st = 0; st < SB; ++st) {
        const idx_t i = base + 2 * st + half, ic = i < we ? i : we - 1;
        A[st] = has_g ? Ai[st] : (int)ic;
        G[st] = has_gg ? Gi[st] : (int)ic;
      }
#pragma unroll
      for (int st = 0; st < SB; ++st) {
        if (KT == 2) {
          const float2 t = *reinterpret_cast<const float2*>(a.A + (int64_t)A[st] * a.a_ld + kbase + 2 * col);
          AV[st][0] = t.x; AV[st][KT - 1] = t.y;
        } else {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) AV[st][kt] = a.A[(int64_t)A[st] * a.a_ld + kbase + kt * 32 + col];
        }
        if (NT == 2) {
          const float2 t = *reinterpret_cast<const float2*>(a.G + (int64_t)G[st] * a.g_ld + nbase + 2 * col);
          GV[st][0] = t.x; GV[st][NT - 1] = t.y;
        } else {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) GV[st][nt] = a.G[(int64_t)G[st] * a.g_ld + nbase + nt * 32 + col];
        }
      }
    };
    auto mma = [&](idx_t base, const float (&AV)[SB][KT], const float (&GV)[SB][NT]) {
#pragma unroll
      for (int st = 0; st < SB; ++st) {
        const bool in = base + 2 * st + half < we;
        if (CS) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) cs[nt] += in ? GV[st][nt] : 0.f;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const float x = in ? AV[st][kt] : 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, GV[st][nt], acc[kt][nt], 0, 0, 0);
        }
      }
    };
    // Batches of B = 2*SB rows.  At batch j: the rows of batch j+1 are fetched (their ids were issued TWO batches ago, so
    // waiting for them leaves the rows of batch j in flight), the ids of batch j+3 are issued, batch j multiplies.  An id
    // consumed one batch after its issue drained the whole load queue at every batch (the counter is in-order).
    constexpr int B = 2 * SB;
    int ar2[SB], gr2[SB];
    load_ids(wb, ar[0], gr[0]);
    load_ids(wb + B, ar[1], gr[1]);
    load_ids(wb + 2 * B, ar2, gr2);
    load_rows(wb, ar[0], gr[0], av[0], gv[0]);
    auto shift_ids = [&](int (&Anew)[SB], int (&Gnew)[SB]) {  // ring: ar[1] <- ar2 <- Anew (ar[0] is refilled from ar[1] by the caller)
#pragma unroll
      for (int st = 0; st < SB; ++st) { ar[1][st] = ar2[st]; gr[1][st] = gr2[st]; ar2[st] = Anew[st]; gr2[st] = Gnew[st]; }
    };
    for (idx_t b = wb; b < we; b += 2 * B) {
      int an[SB], gn[SB];
      load_rows(b + B, ar[1], gr[1], av[1], gv[1]);
      load_ids(b + 3 * B, an, gn);
      __builtin_amdgcn_sched_barrier(0);
      mma(b, av[0], gv[0]);
      __builtin_amdgcn_sched_barrier(0);  // (keeps the address arithmetic of the next loads behind the MFMAs)
      shift_ids(an, gn);                  // ar[1] = ids of batch j+2, ar2 = ids of batch j+3
      load_rows(b + 2 * B, ar[1], gr[1], av[0], gv[0]);
      load_ids(b + 4 * B, an, gn);
      __builtin_amdgcn_sched_barrier(0);
      mma(b + B, av[1], gv[1]);
      __builtin_amdgcn_sched_barrier(0);
      shift_ids(an, gn);
    }
  }
  constexpr int NACC = KT * NT * 16;
#ifdef HET_ABL_DW_NOEPI  // experiment builds only: the whole flush behind a condition that is never true
  if (a.num_rows >= 0) return;
#endif
  // Flush: the four partial products are summed through LDS by ALL four waves -- wave q owns the quarter [q*Q, (q+1)*Q) of the
  // NACC accumulator registers, the three others hand it theirs -- and every wave adds its quarter to the output.  (Until round 5
  // wave 0 summed and added everything while the other twelve wave slots of the workgroup's LDS share sat empty: the flush
  // without its atomics cost as much as with them, without the LDS pass 0.045 ms less per launch on ogbn-mag: profiles/r05/ab_dense.txt.)
  constexpr int Q = NACC / 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (wave != q) {
      const int sr = wave < q ? wave : wave - 1;
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const int f = q * Q + j;
        smem[((q * 3 + sr) * Q + j) * 64 + lane] = acc[f / (NT * 16)][(f / 16) % NT][f % 16];
      }
    }
  }
  float* csm = smem + 3 * NACC * 64;  // [3][NT][32] column sums of waves 1..3 (CS only)
  if (CS && kbase == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      cs[nt] += __shfl_xor(cs[nt], 32);  // the two rows of a step
      if (wave > 0 && half == 0) csm[((wave - 1) * NT + nt) * 32 + col] = cs[nt];
    }
  }
  __syncthreads();
  if (CS && kbase == 0 && wave == 0 && half == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      atomicAdd(a.colsum + nbase + (NT == 2 ? 2 * col + nt : nt * 32 + col),
                cs[nt] + csm[nt * 32 + col] + csm[(NT + nt) * 32 + col] + csm[(2 * NT + nt) * 32 + col]);
  }
  float* __restrict__ out = a.dW + (int64_t)r * a.dw_rel_stride;
  const int Dh = a.headcat_d;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (wave == q) {
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const int f = q * Q + j, kt = f / (NT * 16), nt = (f / 16) % NT, e = f % 16;
        const int o = (q * 3 * Q + j) * 64 + lane;
        const float v = acc[kt][nt][e] + smem[o] + smem[Q * 64 + o] + smem[2 * Q * 64 + o];
        const int m = (e & 3) + 8 * (e >> 2) + 4 * half;  // row of the 32x32 product tile, column = col
        const int k = kbase + (KT == 2 ? 2 * m + kt : kt * 32 + m), nn = nbase + (NT == 2 ? 2 * col + nt : nt * 32 + col);
        int64_t off;
        if (a.headcat == 1) {
          const int h = nn / Dh, d = nn - h * Dh;
          off = (int64_t)h * Kf * Dh + (int64_t)k * Dh + d;
        } else if (a.headcat == 2) {  // keep the per-head diagonal blocks of the full product only
          const int Kh = a.blockdiag_k, hk = k / Kh, hn = nn / Dh;
          if (hk != hn) continue;
          off = ((int64_t)hk * Kh + (k - hk * Kh)) * Dh + (nn - hn * Dh);
        } else {
          off = (int64_t)k * Xf + nn;
        }
        atomicAdd(out + off, v);
      }
    }
  }
}

template <int KT, int NT>
int launch_dw_kx(const MfmaDwArgs& a, hipStream_t s) {
  const size_t lds = sizeof(float) * (3 * KT * NT * 16 * 64 + (a.colsum ? 3 * NT * 32 : 0));
  HET_REQUIRE(!a.row_scale, "segment dW (MFMA): row scales are applied by the segment-sum pre-pass, not here");
  HET_REQUIRE(a.a_ld % 2 == 0 && a.g_ld % 2 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.G) & 7) == 0,
              "segment dW (MFMA): rows must be 8-byte aligned");
  // 48 KiB of LDS per workgroup -> 3 resident per CU, 768 on the chip: aim for about 4 rounds of them
  static const int64_t n_chunks = [] { const char* v = getenv("HET_DW_CHUNKS"); return v ? (int64_t)atoi(v) : 1536; }();  // A/B switch (same box: 768 0.325, 1536 0.316, 3072 0.354, 6144 0.380 ms per launch on ogbn-mag: every workgroup ends with K*X atomic adds)
  int64_t chunk = ceil_div64(a.num_rows, n_chunks);
  if (chunk < 512) chunk = 512;
  chunk = (chunk + 7) & ~7ll;
  const int64_t gx = ceil_div64(a.num_rows, chunk) + a.num_segs;
  const unsigned gy = (unsigned)((a.K / (KT * 32)) * (a.X / (NT * 32)));
  if (a.colsum) {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_dw_mfma<KT, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HET_KTIME("HET_seg_dw_mfma", s);
    hipLaunchKernelGGL((HET_seg_dw_mfma<KT, NT, true>), dim3((unsigned)gx, gy), dim3(256), lds, s, a, (int)chunk);
  } else {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_dw_mfma<KT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HET_KTIME("HET_seg_dw_mfma", s);
    hipLaunchKernelGGL((HET_seg_dw_mfma<KT, NT>), dim3((unsigned)gx, gy), dim3(256), lds, s, a, (int)chunk);
  }
  HET_LAUNCH_CHECK("HET_seg_dw_mfma");
  return HET_OK;
}

template <int K, int NT>
int launch_kx(const MfmaGemmArgs& a, hipStream_t s) {
  constexpr int X = NT * 32, LDA = K + 4, LDC = X + 4;
  constexpr bool B_REGS = false;
  const size_t lds = sizeof(float) * ((B_REGS ? 0 : K * X) + 4 * 32 * (LDA > LDC ? LDA : LDC));
  const int chunk = chunk_rows_for(a.num_rows);
  const int64_t gx = ceil_div64(a.num_rows, chunk) + a.num_segs;
  HET_REQUIRE(gx < (1ll << 31), "segment GEMM: too many row chunks");
  dim3 grid((unsigned)gx), block(256);
  HET_KTIME(a.dot_w ? "HET_seg_gemm_mfma<dot>" : (a.atomic == 2 ? "HET_seg_gemm_mfma<rmw>" : (a.atomic ? "HET_seg_gemm_mfma<atomic>" : "HET_seg_gemm_mfma<store>")), s);
  if (a.atomic == 2) {
    HET_REQUIRE(!a.dot_w && !a.bias, "segment GEMM (MFMA): the read-modify-write epilogue takes no dot / bias");
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, false, 0, true>), grid, block, lds, s, a, chunk);
  } else if (a.dot_w && a.headcat_d == 16 && NT == 2 && K <= 64) {
    if constexpr (NT == 2 && K <= 64) {
      HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma_dot16<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((HET_seg_gemm_mfma_dot16<K>), grid, block, lds, s, a, chunk);
    }
  } else if (a.dot_w && a.headcat_d == 16) {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, false, 4>), grid, block, lds, s, a, chunk);
  } else if (a.dot_w) {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, false, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, false, -1>), grid, block, lds, s, a, chunk);
  } else if (a.atomic) {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, true>), grid, block, lds, s, a, chunk);
  } else {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, false>), grid, block, lds, s, a, chunk);
  }
  HET_LAUNCH_CHECK("HET_seg_gemm_mfma");
  return HET_OK;
}

template <int K>
int launch_k(const MfmaGemmArgs& a, hipStream_t s) {
  switch (a.X) {
    case 32: return launch_kx<K, 1>(a, s);
    case 64: return launch_kx<K, 2>(a, s);
    default: return launch_kx<K, 4>(a, s);
  }
}

}  // namespace

bool mfma_shape_supported(int K, int X) {
  return (K == 32 || K == 64 || K == 128 || K == 256) && (X == 32 || X == 64 || X == 128 || X == 256);
}

int launch_seg_gemm_mfma_rmw_per_segment(const MfmaGemmArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  HET_REQUIRE(a.num_segs <= kRmwMaxSegments && a.K <= 128 && a.X <= 128 && !a.dot_w && !a.bias,
              "segment GEMM (MFMA): per-segment read-modify-write needs few segments and K, X <= 128");
  for (int r = 0; r < a.num_segs; ++r) {
    MfmaGemmArgs m = a;
    m.seg_ptrs = a.seg_ptrs + r;  // the one segment [seg_ptrs[r], seg_ptrs[r+1]); row indices stay absolute
    m.num_segs = 1;
    m.B = a.B + (int64_t)r * a.b_rel_stride;
    m.atomic = 2;
    if (int rc = launch_seg_gemm_mfma(m, s)) return rc;
  }
  return HET_OK;
}

int launch_seg_gemm_mfma(const MfmaGemmArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  HET_REQUIRE(mfma_shape_supported(a.K, a.X), "segment GEMM (MFMA): unsupported shape K=%d X=%d", a.K, a.X);
  if (a.K > 128 || a.X > 128) {
    // 256-wide sides as 128-wide slabs of the weight (one launch each; K slabs after the first add atomically)
    HET_REQUIRE(!a.dot_w && !a.bias && !a.b_k0 && !a.b_n0, "segment GEMM (MFMA): the dot / bias epilogues need K, X <= 128");
    for (int n0 = 0; n0 < a.X; n0 += 128) {
      bool first = true;  // the first window of a column slab stores (unless the caller accumulates), the others add
      for (int k0 = 0; k0 < a.K; k0 += 128) {
        MfmaGemmArgs w = a;
        w.K = a.K - k0 < 128 ? a.K - k0 : 128;
        w.X = a.X - n0 < 128 ? a.X - n0 : 128;
        if (a.b_headcat == 2) {  // block diagonal: a window that no head's block touches is all zeros
          const int Kh = a.blockdiag_k, Dh = a.headcat_d;
          if ((k0 + w.K - 1) / Kh < n0 / Dh || k0 / Kh > (n0 + w.X - 1) / Dh) continue;
        }
        w.A = a.A + k0; w.C = a.C + n0;
        w.b_k0 = k0; w.b_n0 = n0; w.b_kfull = a.K; w.b_xfull = a.X;
        w.atomic = a.atomic || !first;
        first = false;
        if (int rc = launch_seg_gemm_mfma(w, s)) return rc;
      }
    }
    return HET_OK;
  }
  // (Measured and dropped in round 4: K = 128 as two 64-deep passes, the second adding into the rows the first stored -- the
  //  64-deep kernel runs two workgroups per CU against one for the 128-deep one, but reading and writing C once more costs more
  //  than the occupancy gives: RGAT at feat 128 10.7 -> 11.5 ms per step.)
  // K <= 64 into X = 128 (HGT's k' | m rows): two launches of the X = 64 kernel beat the one with four column tiles per
  // wave (0.60 -> 0.55 ms for 2.4 M rows), although the A rows are read twice.  HET_GEMM_XSLAB64=0: A/B switch
  static const bool xslab64 = [] { const char* v = getenv("HET_GEMM_XSLAB64"); return !(v && v[0] == '0'); }();
  if (xslab64 && a.X == 128 && a.K <= 64 && !a.dot_w && !a.bias && !a.atomic) {
    for (int n0 = 0; n0 < 128; n0 += 64) {  // two 64-wide column slabs (the A rows are read twice)
      MfmaGemmArgs w = a;
      // (a may itself be a window of a wider weight -- the 128-wide slabs above: offsets compose, the full size is kept)
      w.X = 64; w.C = a.C + n0; w.b_n0 = a.b_n0 + n0;
      w.b_kfull = a.b_kfull ? a.b_kfull : a.K; w.b_xfull = a.b_xfull ? a.b_xfull : a.X;
      if (int rc = launch_seg_gemm_mfma(w, s)) return rc;
    }
    return HET_OK;
  }
  HET_REQUIRE(a.a_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0, "segment GEMM (MFMA): A rows must be 16-byte aligned");
  HET_REQUIRE(a.c_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0, "segment GEMM (MFMA): C rows must be 16-byte aligned");
  HET_REQUIRE(!a.row_scale, "segment GEMM (MFMA): row scales are applied by the segment-sum pre-pass, not here");
  if (a.dot_w) {
    const int Dh = a.headcat_d;
    HET_REQUIRE(a.dot_out && !a.atomic && a.b_headcat == 1 && Dh >= 4 && (Dh & (Dh - 1)) == 0 && a.X % Dh == 0 &&
                    (reinterpret_cast<uintptr_t>(a.dot_w) & 15) == 0,
                "segment GEMM (MFMA): the dot epilogue needs plain stores, head-concatenated weights and a power-of-two D >= 4");
  }
  switch (a.K) {
    case 32: return launch_k<32>(a, s);
    case 64: return launch_k<64>(a, s);
    default: return launch_k<128>(a, s);
  }
}

bool mfma_dw_supported(int K, int X) {
  return (K == 32 || K == 64 || K == 128 || K == 256) && (X == 32 || X == 64 || X == 128 || X == 256);
}

int launch_seg_dw_mfma(const MfmaDwArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  HET_REQUIRE(mfma_dw_supported(a.K, a.X), "segment dW (MFMA): unsupported shape K=%d X=%d", a.K, a.X);
  // 64-wide blocks of the product per workgroup; K or X = 128 are covered by 2 (or 4) blocks along blockIdx.y
  if (a.K == 32) return a.X == 32 ? launch_dw_kx<1, 1>(a, s) : launch_dw_kx<1, 2>(a, s);
  return a.X == 32 ? launch_dw_kx<2, 1>(a, s) : launch_dw_kx<2, 2>(a, s);
}
