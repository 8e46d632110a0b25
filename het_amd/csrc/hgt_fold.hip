// HGT: the per-step folding of the layer's parameters into one weight per relation, forward and backward, as two launches.
//
// The one-node HGT attention (het_amd/backend/hgt_fused_layer.py) projects the distinct (relation, source) rows with
//   w_kv[r] = [ K_st(r) . att'[r] . pri[r] / sqrt(dk)  |  V_st(r) . msg[r] ]        [in, 2 H dk]
// (the reference composes the same factors edge by edge: HGT/models.py:159-262, k = K_linear(h), then per relation
// k . relation_att, . relation_pri / sqrt_dk, v . relation_msg).  Written with torch ops that is ~14 launches of a few
// microseconds each per step and ~24 more for its autograd backward -- 0.3 ms of a 6.4 ms step on ogbn-mag
// (profiles/r04/hgt_timeline.txt: 51 launches under 20 us).  The tensors are tiny ([R, 64, 128] on ogbn-mag): one thread per output.
#include "common.hip.h"

namespace {
constexpr int kBlock = 256;

// A[r,h,d,e]: rel_att as stored (fused score: s = <k . att, q>) or transposed (s = <q . att, k> = <k . att^T, q>)
__device__ __forceinline__ float att_at(const float* __restrict__ att, int64_t rh, int dk, int d, int e, int transpose) {
  return att[rh * dk * dk + (transpose ? e * dk + d : d * dk + e)];
}

__global__ __launch_bounds__(kBlock) void HET_hgt_fold_weights(const float* __restrict__ k_lin, const float* __restrict__ v_lin,
                                                               const float* __restrict__ att, const float* __restrict__ msg,
                                                               const float* __restrict__ pri, const int64_t* __restrict__ src_type,
                                                               int R, int H, int dk, int K_in, int transpose, float* __restrict__ w_kv) {
  const int X = H * dk;
  const int64_t total = (int64_t)R * K_in * 2 * X;
  const float inv_sqrt = rsqrtf((float)dk);
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(t % (2 * X));
    const int64_t ri = t / (2 * X);
    const int i = (int)(ri % K_in), r = (int)(ri / K_in);
    const int part = c / X, hc = c % X, h = hc / dk, e = hc % dk;
    const int64_t st = src_type[r], rh = (int64_t)r * H + h;
    const float* lin = (part ? v_lin : k_lin) + (st * K_in + i) * X + h * dk;
    float s = 0.f;
    if (part == 0) {
      for (int d = 0; d < dk; ++d) s = fmaf(lin[d], att_at(att, rh, dk, d, e, transpose), s);
      s *= pri[rh] * inv_sqrt;
    } else {
      for (int d = 0; d < dk; ++d) s = fmaf(lin[d], msg[rh * dk * dk + d * dk + e], s);
    }
    w_kv[t] = s;
  }
}

// grad_k_lin / grad_v_lin [T, K_in, X]: thread per (part, t, i, h*dk + d); relations of source type t are summed in order
__global__ __launch_bounds__(kBlock) void HET_hgt_fold_backward_lin(const float* __restrict__ g, const float* __restrict__ att,
                                                                    const float* __restrict__ msg, const float* __restrict__ pri,
                                                                    const int64_t* __restrict__ src_type, int T, int R, int H, int dk,
                                                                    int K_in, int transpose, float* __restrict__ grad_k,
                                                                    float* __restrict__ grad_v) {
  const int X = H * dk;
  const int64_t per = (int64_t)T * K_in * X, total = 2 * per;
  const float inv_sqrt = rsqrtf((float)dk);
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int part = (int)(t / per);
    const int64_t u = t % per;
    const int hc = (int)(u % X), h = hc / dk, d = hc % dk;
    const int i = (int)((u / X) % K_in), ty = (int)(u / ((int64_t)X * K_in));
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
      if (src_type[r] != ty) continue;
      const int64_t rh = (int64_t)r * H + h;
      const float* gr = g + ((int64_t)r * K_in + i) * 2 * X + part * X + h * dk;
      float a = 0.f;
      if (part == 0) {
        for (int e = 0; e < dk; ++e) a = fmaf(gr[e], att_at(att, rh, dk, d, e, transpose), a);
        a *= pri[rh] * inv_sqrt;
      } else {
        for (int e = 0; e < dk; ++e) a = fmaf(gr[e], msg[rh * dk * dk + d * dk + e], a);
      }
      s += a;
    }
    (part ? grad_v : grad_k)[u] = s;
  }
}

// workgroup per (relation, head): Tk[d,e] = SUM_i k_lin[st,i,h,d] G_k[r,i,h,e], Tm likewise with v_lin / G_m;
//   grad_att = mu Tk (stored transposed when the score is the unfused one), grad_msg = Tm, grad_pri = SUM_de A Tk / sqrt(dk)
__global__ __launch_bounds__(kBlock) void HET_hgt_fold_backward_rel(const float* __restrict__ g, const float* __restrict__ k_lin,
                                                                    const float* __restrict__ v_lin, const float* __restrict__ att,
                                                                    const float* __restrict__ pri, const int64_t* __restrict__ src_type,
                                                                    int H, int dk, int K_in, int transpose, float* __restrict__ grad_att,
                                                                    float* __restrict__ grad_msg, float* __restrict__ grad_pri) {
  __shared__ float red[kBlock];
  const int X = H * dk;
  const int64_t rh = blockIdx.x;
  const int r = (int)(rh / H), h = (int)(rh % H);
  const int64_t st = src_type[r];
  const float inv_sqrt = rsqrtf((float)dk), mu = pri[rh] * inv_sqrt;
  float acc = 0.f;
  for (int p = threadIdx.x; p < dk * dk; p += kBlock) {
    const int d = p / dk, e = p % dk;
    float tk = 0.f, tm = 0.f;
    for (int i = 0; i < K_in; ++i) {
      const float* gr = g + ((int64_t)r * K_in + i) * 2 * X + h * dk + e;
      const int64_t li = (st * K_in + i) * X + h * dk + d;
      tk = fmaf(k_lin[li], gr[0], tk);
      tm = fmaf(v_lin[li], gr[X], tm);
    }
    grad_att[rh * dk * dk + (transpose ? e * dk + d : d * dk + e)] = mu * tk;
    grad_msg[rh * dk * dk + p] = tm;
    acc = fmaf(att_at(att, rh, dk, d, e, transpose), tk, acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) grad_pri[rh] = red[0] * inv_sqrt;
}

inline unsigned grid_for(int64_t total) {
  const int64_t b = ceil_div64(total, kBlock);
  return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}
}  // namespace

extern "C" int het_hgt_fold_source_weights(const float* k_lin, const float* v_lin, const float* rel_att, const float* rel_msg,
                                           const float* rel_pri, const int64_t* src_type, int64_t num_types, int64_t num_rels,
                                           int64_t H, int64_t dk, int64_t K_in, int transpose_att, float* w_kv, het_stream stream) {
  const char* op = "het_hgt_fold_source_weights";
  HET_REQUIRE(k_lin && v_lin && rel_att && rel_msg && rel_pri && src_type && w_kv, "%s: null argument", op);
  HET_REQUIRE(num_types > 0 && num_rels > 0 && H > 0 && dk > 0 && K_in > 0 && num_rels * K_in * 2 * H * dk < (1ll << 40), "%s: bad sizes", op);
  hipLaunchKernelGGL(HET_hgt_fold_weights, dim3(grid_for(num_rels * K_in * 2 * H * dk)), dim3(kBlock), 0, (hipStream_t)stream, k_lin, v_lin,
                     rel_att, rel_msg, rel_pri, src_type, (int)num_rels, (int)H, (int)dk, (int)K_in, transpose_att, w_kv);
  HET_LAUNCH_CHECK("HET_hgt_fold_weights");
  return HET_OK;
}

extern "C" int het_hgt_fold_source_weights_backward(const float* grad_w_kv, const float* k_lin, const float* v_lin, const float* rel_att,
                                                    const float* rel_msg, const float* rel_pri, const int64_t* src_type, int64_t num_types,
                                                    int64_t num_rels, int64_t H, int64_t dk, int64_t K_in, int transpose_att,
                                                    float* grad_k_lin, float* grad_v_lin, float* grad_att, float* grad_msg, float* grad_pri,
                                                    het_stream stream) {
  const char* op = "het_hgt_fold_source_weights_backward";
  HET_REQUIRE(grad_w_kv && k_lin && v_lin && rel_att && rel_msg && rel_pri && src_type && grad_k_lin && grad_v_lin && grad_att && grad_msg &&
                  grad_pri, "%s: null argument", op);
  HET_REQUIRE(num_types > 0 && num_rels > 0 && H > 0 && dk > 0 && K_in > 0, "%s: bad sizes", op);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(HET_hgt_fold_backward_lin, dim3(grid_for(2 * num_types * K_in * H * dk)), dim3(kBlock), 0, s, grad_w_kv, rel_att, rel_msg,
                     rel_pri, src_type, (int)num_types, (int)num_rels, (int)H, (int)dk, (int)K_in, transpose_att, grad_k_lin, grad_v_lin);
  HET_LAUNCH_CHECK("HET_hgt_fold_backward_lin");
  hipLaunchKernelGGL(HET_hgt_fold_backward_rel, dim3((unsigned)(num_rels * H)), dim3(kBlock), 0, s, grad_w_kv, k_lin, v_lin, rel_att, rel_pri,
                     src_type, (int)H, (int)dk, (int)K_in, transpose_att, grad_att, grad_msg, grad_pri);
  HET_LAUNCH_CHECK("HET_hgt_fold_backward_rel");
  return HET_OK;
}
