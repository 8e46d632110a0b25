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
#include "seg_gemm_mfma.hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChunkRows = 2048;  // rows of one relation per workgroup (16 tiles per wave)

template <int K, int NT, bool ATOMIC>
__global__ __launch_bounds__(256) void HET_seg_gemm_mfma(MfmaGemmArgs a) {
  constexpr int X = NT * 32, KH = K / 2;
  constexpr int LDA = K + 4, LPRA = K / 4, RPIA = 64 / LPRA, NITA = 32 / RPIA;  // A tile: rows per load instr
  constexpr int LDC = X + 4, LPRC = X / 4, RPIC = 64 / LPRC, NITC = 32 / RPIC;  // C tile: rows per store instr
  constexpr int WREG = 32 * (LDA > LDC ? LDA : LDC);                            // per-wave LDS floats
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int r;
  idx_t rb, re;
  if (!tile_to_relation(a.seg_ptrs, a.num_segs, kChunkRows, blockIdx.x, r, rb, re)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Small weights (K*X <= 64*64) live in registers: every lane keeps the KH*NT B-operand values it feeds
  // to the MFMAs, so the MFMA loop issues back to back with no LDS read (and no barrier) in it.  Larger
  // weights are staged once per workgroup in LDS.
  constexpr bool B_REGS = KH * NT <= 64;
  float* Bs = smem;                                           // [K][X] (unused when B_REGS)
  float* Ws = smem + (B_REGS ? 0 : K * X) + wave * WREG;      // wave-private: A tile [32][LDA], then C tile [32][LDC]
  const float* __restrict__ Bm = a.B + (int64_t)r * a.b_rel_stride;
  auto b_elem = [&](int k, int n) -> float {
    if (a.b_headcat == 1) {
      const int Dh = a.headcat_d, h = n / Dh, d = n - h * Dh;
      return Bm[(int64_t)h * K * Dh + (int64_t)k * Dh + d];
    }
    if (a.b_headcat == 2) {  // block diagonal: per-head [Kh x Dh] blocks, A and C rows are [H*Kh] / [H*Dh]
      const int Dh = a.headcat_d, Kh = a.blockdiag_k, hk = k / Kh, hn = n / Dh;
      return hk == hn ? Bm[((int64_t)hk * Kh + (k - hk * Kh)) * Dh + (n - hn * Dh)] : 0.f;
    }
    return Bm[k * X + n];
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

  // Loads are issued in phases of independent instructions (all row indices, then all rows) and are
  // branch-free -- out-of-range rows are clamped to the last row and masked afterwards -- so that the
  // compiler can keep a whole phase in flight instead of waiting after every dependent pair.
  float4 areg[NITA];
  auto load_tile = [&](idx_t wb) {
    idx_t ar[NITA];
    if (a.gather) {
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const idx_t i = wb + it * RPIA + ra;
        ar[it] = a.gather[i < re ? i : re - 1];
      }
    } else {
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const idx_t i = wb + it * RPIA + ra;
        ar[it] = i < re ? i : re - 1;
      }
    }
#ifdef HET_ABL_NOLOAD
#pragma unroll
    for (int it = 0; it < NITA; ++it) ar[it] = ra;
#endif
#pragma unroll
    for (int it = 0; it < NITA; ++it) areg[it] = *reinterpret_cast<const float4*>(a.A + ar[it] * a.a_ld + ca);
    if (a.row_scale) {
      idx_t si[NITA];
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const idx_t i = wb + it * RPIA + ra, ic = i < re ? i : re - 1;
        si[it] = a.scale_idx ? a.scale_idx[ic] : ic;
      }
#pragma unroll
      for (int it = 0; it < NITA; ++it) {
        const float sc = a.row_scale[si[it]];
        areg[it].x *= sc; areg[it].y *= sc; areg[it].z *= sc; areg[it].w *= sc;
      }
    }
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      if (wb + it * RPIA + ra >= re) areg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  };

  idx_t wb = rb + wave * 32;
  if (wb < re) load_tile(wb);
  for (; wb < re; wb += 128) {
#pragma unroll
    for (int it = 0; it < NITA; ++it)
      *reinterpret_cast<float4*>(&Ws[(it * RPIA + ra) * LDA + ca]) = areg[it];
    // C rows of this tile (needed in the epilogue) and the A rows of the next tile: both stay in
    // flight while the MFMAs below run
    idx_t crow[NITC];
    if (a.scatter) {
#pragma unroll
      for (int it = 0; it < NITC; ++it) {
        const idx_t i = wb + it * RPIC + rc;
        crow[it] = a.scatter[i < re ? i : re - 1];
      }
    } else {
#pragma unroll
      for (int it = 0; it < NITC; ++it) crow[it] = wb + it * RPIC + rc;
    }
#pragma unroll
    for (int it = 0; it < NITC; ++it)
      if (wb + it * RPIC + rc >= re) crow[it] = -1;
    if (wb + 128 < re) load_tile(wb + 128);

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
#pragma unroll
    for (int s = 0; s < KH; ++s) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float b = B_REGS ? breg[B_REGS ? s * NT + nt : 0] : Bs[(half * KH + s) * X + nt * 32 + row];
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b, acc[nt], 0, 0, 0);
      }
    }
    // Epilogue: transpose through the wave's LDS region so that every output row leaves as whole
    // 16-byte pieces (X/4 lanes x float4 per row) instead of 32 row-strided dword stores.
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    if (ATOMIC) {
      // straight from the accumulators: one atomic instruction adds two 128-byte row segments, the shape
      // float atomics run at full rate with (MI355X_MICROARCH.md, Global float atomics)
      // C-row ids go through the wave's LDS region (all lanes of a store row group hold the same id)
#pragma unroll
      for (int it = 0; it < NITC; ++it) reinterpret_cast<idx_t*>(Ws)[it * RPIC + rc] = crow[it];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const idx_t cr = reinterpret_cast<const idx_t*>(Ws)[(reg & 3) + 8 * (reg >> 2) + 4 * half];
        if (cr < 0) continue;
        float* p = a.C + cr * a.c_ld + row;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) atomicAdd(p + nt * 32, acc[nt][reg]);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          Ws[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LDC + nt * 32 + row] = acc[nt][reg];
#pragma unroll
      for (int it = 0; it < NITC; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(&Ws[(it * RPIC + rc) * LDC + cc]);
        if (crow[it] < 0) continue;
#ifdef HET_ABL_NOSTORE
        if (a.num_rows < 0)
#endif
        *reinterpret_cast<float4*>(a.C + crow[it] * a.c_ld + cc) = v;
      }
    }
  }
}

// ---- weight gradient -----------------------------------------------------------------------------
// One workgroup = 4 waves owns a chunk of rows of one relation; every wave walks a quarter of the
// chunk two rows per MFMA step (lane half h reads row 2s+h: 32 consecutive floats of the A row as
// the MFMA A operand -- A is used transposed, feature index on the M axis -- and 32 consecutive
// floats of the G row as the B operand), accumulating the full K x X product in KT*NT 32x32
// accumulators.  The four partial products are summed through LDS and flushed with one atomic
// add per element and workgroup.
template <int KT, int NT>
__global__ __launch_bounds__(256) void HET_seg_dw_mfma(MfmaDwArgs a, int chunk) {
  constexpr int K = KT * 32, X = NT * 32;
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
#pragma unroll 2
  for (idx_t i0 = wb; i0 < we; i0 += 2) {
    const idx_t i = i0 + half;
    float av[KT], gv[NT];
    if (i < we) {
      const idx_t ar = a.gather ? a.gather[i] : i, gr = a.g_gather ? a.g_gather[i] : i;
      const float sc = a.row_scale ? a.row_scale[a.scale_idx ? a.scale_idx[i] : i] : 1.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[kt] = a.A[ar * a.a_ld + kt * 32 + col] * sc;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) gv[nt] = a.G[gr * a.g_ld + nt * 32 + col];
    } else {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[kt] = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) gv[nt] = 0.f;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kt], gv[nt], acc[kt][nt], 0, 0, 0);
  }
  constexpr int NACC = KT * NT * 16;
  if (wave > 0) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) smem[((wave - 1) * NACC + (kt * NT + nt) * 16 + e) * 64 + lane] = acc[kt][nt][e];
  }
  __syncthreads();
  if (wave == 0) {
    float* __restrict__ out = a.dW + (int64_t)r * a.dw_rel_stride;
    const int Dh = a.headcat_d;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int o = ((kt * NT + nt) * 16 + e) * 64 + lane;
          const float v = acc[kt][nt][e] + smem[o] + smem[NACC * 64 + o] + smem[2 * NACC * 64 + o];
          const int k = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half, nn = nt * 32 + col;
          int64_t off;
          if (a.headcat == 1) {
            const int h = nn / Dh, d = nn - h * Dh;
            off = (int64_t)h * K * Dh + (int64_t)k * Dh + d;
          } else if (a.headcat == 2) {  // keep the per-head diagonal blocks of the full product only
            const int Kh = a.blockdiag_k, hk = k / Kh, hn = nn / Dh;
            if (hk != hn) continue;
            off = ((int64_t)hk * Kh + (k - hk * Kh)) * Dh + (nn - hn * Dh);
          } else {
            off = (int64_t)k * X + nn;
          }
          atomicAdd(out + off, v);
        }
  }
}

template <int KT, int NT>
int launch_dw_kx(const MfmaDwArgs& a, hipStream_t s) {
  const size_t lds = sizeof(float) * 3 * KT * NT * 16 * 64;
  int64_t chunk = ceil_div64(a.num_rows, 1024);  // about 1024 workgroups
  if (chunk < 512) chunk = 512;
  chunk = (chunk + 7) & ~7ll;
  const int64_t gx = ceil_div64(a.num_rows, chunk) + a.num_segs;
  HET_HIP(hipFuncSetAttribute((const void*)HET_seg_dw_mfma<KT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((HET_seg_dw_mfma<KT, NT>), dim3((unsigned)gx), dim3(256), lds, s, a, (int)chunk);
  HET_LAUNCH_CHECK("HET_seg_dw_mfma");
  return HET_OK;
}

template <int K, int NT>
int launch_kx(const MfmaGemmArgs& a, hipStream_t s) {
  constexpr int X = NT * 32, LDA = K + 4, LDC = X + 4;
  constexpr bool B_REGS = (K / 2) * NT <= 64;
  const size_t lds = sizeof(float) * ((B_REGS ? 0 : K * X) + 4 * 32 * (LDA > LDC ? LDA : LDC));
  const int64_t gx = ceil_div64(a.num_rows, kChunkRows) + a.num_segs;
  HET_REQUIRE(gx < (1ll << 31), "segment GEMM: too many row chunks");
  dim3 grid((unsigned)gx), block(256);
  if (a.atomic) {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, true>), grid, block, lds, s, a);
  } else {
    HET_HIP(hipFuncSetAttribute((const void*)HET_seg_gemm_mfma<K, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((HET_seg_gemm_mfma<K, NT, false>), grid, block, lds, s, a);
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
  return (K == 32 || K == 64 || K == 128) && (X == 32 || X == 64 || X == 128);
}

int launch_seg_gemm_mfma(const MfmaGemmArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  HET_REQUIRE(mfma_shape_supported(a.K, a.X), "segment GEMM (MFMA): unsupported shape K=%d X=%d", a.K, a.X);
  HET_REQUIRE(a.a_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0, "segment GEMM (MFMA): A rows must be 16-byte aligned");
  HET_REQUIRE(a.c_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0, "segment GEMM (MFMA): C rows must be 16-byte aligned");
  switch (a.K) {
    case 32: return launch_k<32>(a, s);
    case 64: return launch_k<64>(a, s);
    default: return launch_k<128>(a, s);
  }
}

bool mfma_dw_supported(int K, int X) { return (K == 32 || K == 64) && (X == 32 || X == 64); }

int launch_seg_dw_mfma(const MfmaDwArgs& a, hipStream_t s) {
  if (a.num_rows == 0) return HET_OK;
  HET_REQUIRE(mfma_dw_supported(a.K, a.X), "segment dW (MFMA): unsupported shape K=%d X=%d", a.K, a.X);
  if (a.K == 32) return a.X == 32 ? launch_dw_kx<1, 1>(a, s) : launch_dw_kx<1, 2>(a, s);
  return a.X == 32 ? launch_dw_kx<2, 1>(a, s) : launch_dw_kx<2, 2>(a, s);
}
