// BatchNormalization bookkeeping kernels (SURVEY 8a row a11 and its part of a15).
// The heavy passes live in the producers/consumers (GEMM epilogue statistics, depthwise kernels
// applying scale/shift on load); what is left here are the per-channel finalisations, which reduce
// the per-tile partial sums in a FIXED order (double accumulation) so results are bit-reproducible.
#include "internal.h"

namespace {

// finalise kernels: 256 threads = FIN_CG channels x FIN_RG row groups; up to 256 partial rows are summed
// directly (<= 16 per thread; more rows with only C/16 workgroups is latency-bound: 2048 rows took 35 us), row group r takes rows r, r+FIN_RG, ... and the groups are combined in order
constexpr int FIN_CG = 16, FIN_RG = 16;
#ifndef KWS_FIN_U
#define KWS_FIN_U 16
#endif
constexpr int FIN_U = KWS_FIN_U;   // rows per thread and trip

__device__ float g_zero_fin[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

// part[n_tiles][2][C] -> bn[4C] = scale | shift | mean | rstd ; moving stats update in place.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int n_tiles,
                                                                double inv_count, int C,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                float one_minus_momentum, float* moving_mean,
                                                                float* moving_var, float* __restrict__ bn) {
  __shared__ double red[2][FIN_RG][FIN_CG];
  const int cg = threadIdx.x % FIN_CG, rg = threadIdx.x / FIN_CG;
  const int c = blockIdx.x * FIN_CG + cg;
  double s = 0.0, ss = 0.0;
  if (c < C) {
    // FIN_U rows' loads in flight per trip - with <= 256 partial rows (the producers' caps) ONE trip: the kernel is a chain
    // of memory round trips (the rows were just written by other XCDs) and little else.  Rows past the end read a zero
    // buffer (address select, not a branch); the additions stay in ascending row order.
    for (int t = rg; t < n_tiles; t += FIN_U * FIN_RG) {
      float a[FIN_U], b[FIN_U];
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int tt = t + u * FIN_RG;
        const float* src = tt < n_tiles ? part + (int64_t)tt * 2 * C + c : g_zero_fin;
        a[u] = src[0];
        b[u] = src[tt < n_tiles ? C : 1];
      }
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        s += (double)a[u];
        ss += (double)b[u];
      }
    }
  }
  red[0][rg][cg] = s;
  red[1][rg][cg] = ss;
  __syncthreads();
  if (rg == 0 && c < C) {
    s = 0.0;
    ss = 0.0;
    for (int r = 0; r < FIN_RG; ++r) {
      s += red[0][r][cg];
      ss += red[1][r][cg];
    }
    const double mean = s * inv_count;
    double var = ss * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean, varf = (float)var;
    const float scale = gamma[c] * rstd;
    bn[c] = scale;
    bn[C + c] = beta[c] - meanf * scale;
    bn[2 * C + c] = meanf;
    bn[3 * C + c] = rstd;
    if (moving_mean) {
      // AssignMovingAvg: m -= (m - batch) * (1 - momentum); biased variance
      moving_mean[c] = moving_mean[c] - (moving_mean[c] - meanf) * one_minus_momentum;
      moving_var[c] = moving_var[c] - (moving_var[c] - varf) * one_minus_momentum;
    }
  }
}

// Stage 1 of the two-stage partial-sum reduction: in[n][W] -> out[S][W], slice s sums rows
// [s*rows_per, (s+1)*rows_per) in ascending order (fixed order => bit-reproducible).  One thread per
// (column, slice): coalesced across columns, thousands of threads, so the per-tile partials of the big
// early layers (3000+ tiles) are folded at memory speed instead of by a handful of serial loops.
__global__ __launch_bounds__(256) void slice_reduce_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int n, int W, int rows_per) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (w >= W) return;
  const int r0 = s * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > n) r1 = n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += in[(int64_t)r * W + w];
    a1 += in[(int64_t)(r + 1) * W + w];
    a2 += in[(int64_t)(r + 2) * W + w];
    a3 += in[(int64_t)(r + 3) * W + w];
  }
  for (; r < r1; ++r) a0 += in[(int64_t)r * W + w];
  out[(int64_t)s * W + w] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void bn_infer_prepare_kernel(const float* gamma, const float* beta,
                                                               const float* mm, const float* mv, float eps, int C,
                                                               float* bn) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(mv[c] + eps);
  const float scale = gamma[c] * rstd;
  bn[c] = scale;
  bn[C + c] = beta[c] - mm[c] * scale;
  bn[2 * C + c] = mm[c];
  bn[3 * C + c] = rstd;
}

// the inference tables of SEVERAL BatchNorm layers in one launch (round 4: a predict call prepared its twelve layers with twelve
// 4.5 us launches); blockIdx.y = layer, same arithmetic per channel as bn_infer_prepare_kernel
struct BnInferBatch {
  const float* gamma[KWS_BN_INFER_BATCH];
  const float* beta[KWS_BN_INFER_BATCH];
  const float* mm[KWS_BN_INFER_BATCH];
  const float* mv[KWS_BN_INFER_BATCH];
  float* bn[KWS_BN_INFER_BATCH];
  int C[KWS_BN_INFER_BATCH];
  float eps;
};
__global__ __launch_bounds__(256) void bn_infer_prepare_batch_kernel(BnInferBatch b) {
  const int l = blockIdx.y, C = b.C[l];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(b.mv[l][c] + b.eps);
  const float scale = b.gamma[l][c] * rstd;
  float* bn = b.bn[l];
  bn[c] = scale;
  bn[C + c] = b.beta[l][c] - b.mm[l][c] * scale;
  bn[2 * C + c] = b.mm[l][c];
  bn[3 * C + c] = rstd;
}

__global__ __launch_bounds__(256) void bn_relu6_apply_kernel(const float* __restrict__ y, const float* __restrict__ bn,
                                                             float* __restrict__ out, int64_t n4, int C, int relu6) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const float4 v = reinterpret_cast<const float4*>(y)[i];
  const float4 sc = *reinterpret_cast<const float4*>(bn + c);
  const float4 sh = *reinterpret_cast<const float4*>(bn + C + c);
  float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
  if (relu6) o = make_float4(relu6f(o.x), relu6f(o.y), relu6f(o.z), relu6f(o.w));
  reinterpret_cast<float4*>(out)[i] = o;
}

// part[n_parts][5][C] -> dgamma, dbeta, dw[3][C], coef[2C] = (sum g / n, sum g*xhat / n)
__global__ __launch_bounds__(256) void dw_bwd_finalize_kernel(const float* __restrict__ part, int n_parts,
                                                              double inv_count, int C, float* dw, float* dgamma,
                                                              float* dbeta, float* coef) {
  __shared__ double red[5][FIN_RG][FIN_CG];
  const int cg = threadIdx.x % FIN_CG, rg = threadIdx.x / FIN_CG;
  const int c = blockIdx.x * FIN_CG + cg;
  double s[5] = {0, 0, 0, 0, 0};
  if (c < C) {
    for (int t = rg; t < n_parts; t += FIN_U * FIN_RG) {   // as in bn_stats_finalize_kernel: 5 FIN_U loads in flight per trip
      float a[FIN_U][5];
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int tt = t + u * FIN_RG;
        const bool ok = tt < n_parts;
        const float* src = ok ? part + (int64_t)tt * 5 * C + c : g_zero_fin;
#pragma unroll
        for (int q = 0; q < 5; ++q) a[u][q] = src[ok ? q * C : q];
      }
#pragma unroll
      for (int u = 0; u < FIN_U; ++u)
#pragma unroll
        for (int q = 0; q < 5; ++q) s[q] += (double)a[u][q];
    }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) red[q][rg][cg] = s[q];
  __syncthreads();
  if (rg == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      double a = 0.0;
      for (int r = 0; r < FIN_RG; ++r) a += red[q][r][cg];
      s[q] = a;
    }
    if (dbeta) dbeta[c] = (float)s[0];
    if (dgamma) dgamma[c] = (float)s[1];
    if (coef) {
      coef[c] = (float)(s[0] * inv_count);
      coef[C + c] = (float)(s[1] * inv_count);
    }
    if (dw) {
      dw[c] = (float)s[2];
      dw[C + c] = (float)s[3];
      dw[2 * C + c] = (float)s[4];
    }
  }
}

// The depthwise weight gradients of SEVERAL layers in one launch (round 4, net_logmfcc.hip): a block's FIRST depthwise
// convolution reads a materialised activation, so its backward leaves only dw partial rows - nothing on the dependency chain
// needs their fold.  Each layer keeps its rows in a region of its own until this launch.  Same summation order per channel as
// dw_bwd_finalize_kernel (row groups r, r + FIN_RG, ...; groups combined in order; double): bit-identical gradients.
struct DwFinBatch {
  const float* part[KWS_DW_FIN_BATCH];
  float* dw[KWS_DW_FIN_BATCH];
  int n_parts[KWS_DW_FIN_BATCH], C[KWS_DW_FIN_BATCH], blk_end[KWS_DW_FIN_BATCH];
  int n;
};
__global__ __launch_bounds__(256) void dw_grad_finalize_batch_kernel(DwFinBatch b) {
  __shared__ double red[3][FIN_RG][FIN_CG];
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.blk_end[m]) ++m;
  const int blk = blockIdx.x - (m ? b.blk_end[m - 1] : 0);
  const int C = b.C[m], n_parts = b.n_parts[m];
  const float* part = b.part[m];
  const int cg = threadIdx.x % FIN_CG, rg = threadIdx.x / FIN_CG;
  const int c = blk * FIN_CG + cg;
  double s[3] = {0, 0, 0};
  if (c < C) {
    for (int t = rg; t < n_parts; t += FIN_U * FIN_RG) {
      float a[FIN_U][3];
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int tt = t + u * FIN_RG;
        const bool ok = tt < n_parts;
        const float* src = ok ? part + (int64_t)tt * 5 * C + c : g_zero_fin;
#pragma unroll
        for (int q = 0; q < 3; ++q) a[u][q] = src[ok ? (q + 2) * C : q];
      }
#pragma unroll
      for (int u = 0; u < FIN_U; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) s[q] += (double)a[u][q];
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) red[q][rg][cg] = s[q];
  __syncthreads();
  if (rg == 0 && c < C) {
    float* dw = b.dw[m];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double a = 0.0;
      for (int r = 0; r < FIN_RG; ++r) a += red[q][r][cg];
      dw[q * C + c] = (float)a;
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ bn, const float* __restrict__ gamma,
                                                           const float* __restrict__ coef, int64_t n4, int C,
                                                           unsigned* amax) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const float4 gv = reinterpret_cast<const float4*>(g)[i];
  const float4 yv = reinterpret_cast<const float4*>(y)[i];
  const float4 mean = *reinterpret_cast<const float4*>(bn + 2 * C + c);
  const float4 rstd = *reinterpret_cast<const float4*>(bn + 3 * C + c);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  const float4 c1 = *reinterpret_cast<const float4*>(coef + c);
  const float4 c2 = *reinterpret_cast<const float4*>(coef + C + c);
  float4 o;
  o.x = ga.x * rstd.x * (gv.x - c1.x - (yv.x - mean.x) * rstd.x * c2.x);
  o.y = ga.y * rstd.y * (gv.y - c1.y - (yv.y - mean.y) * rstd.y * c2.y);
  o.z = ga.z * rstd.z * (gv.z - c1.z - (yv.z - mean.z) * rstd.z * c2.z);
  o.w = ga.w * rstd.w * (gv.w - c1.w - (yv.w - mean.w) * rstd.w * c2.w);
  reinterpret_cast<float4*>(g)[i] = o;
  if (amax) kws_absmax_commit(amax, kws_abs4max(0.f, o));
}

// Folds n partial rows of width W into at most KWS_REDUCE_SLICES rows in `scratch` when that pays.
#ifndef KWS_PRE_REDUCE_MIN
#define KWS_PRE_REDUCE_MIN (8 * KWS_REDUCE_SLICES)   // 256
#endif
int pre_reduce(const float* part, int n, int W, float* scratch, hipStream_t st, const float** out_part, int* out_n) {
  if (scratch == nullptr || n <= KWS_PRE_REDUCE_MIN) {      // up to this many rows the finalise kernels sum them directly
    *out_part = part;
    *out_n = n;
    return KWS_OK;
  }
  const int rows_per = ceil_div(n, KWS_REDUCE_SLICES);
  const int S = ceil_div(n, rows_per);
  hipLaunchKernelGGL(slice_reduce_kernel, dim3((unsigned)ceil_div(W, 256), (unsigned)S), dim3(256), 0, st, part, scratch,
                     n, W, rows_per);
  KWS_LAUNCH_CHECK("slice_reduce_kernel");
  *out_part = scratch;
  *out_n = S;
  return KWS_OK;
}

}  // namespace

// internal (net_logmfcc.hip): dw[i][3][C_i] = column sums 2..4 of part[i][n_parts_i][5][C_i], count <= KWS_DW_FIN_BATCH layers, one launch
int kws_dw_grad_finalize_batch(const float* const* part, const int* n_parts, const int* C, float* const* dw, int count,
                               hipStream_t stream) {
  KWS_REQUIRE(part && n_parts && C && dw && count > 0 && count <= KWS_DW_FIN_BATCH, "dw_grad_finalize_batch: bad arguments (count=%d)", count);
  DwFinBatch b;
  int blocks = 0;
  double bytes = 0;
  for (int i = 0; i < count; ++i) {
    KWS_REQUIRE(part[i] && dw[i] && n_parts[i] > 0 && n_parts[i] <= 8 * KWS_REDUCE_SLICES && C[i] > 0,
                "dw_grad_finalize_batch: bad entry %d (rows %d)", i, n_parts[i]);
    b.part[i] = part[i]; b.dw[i] = dw[i]; b.n_parts[i] = n_parts[i]; b.C[i] = C[i];
    blocks += ceil_div(C[i], FIN_CG);
    b.blk_end[i] = blocks;
    bytes += 12.0 * n_parts[i] * C[i];
  }
  b.n = count;
  KwsProfScope prof("bn_finalize", 0.0, bytes, stream);
  hipLaunchKernelGGL(dw_grad_finalize_batch_kernel, dim3((unsigned)blocks), dim3(FIN_CG * FIN_RG), 0, stream, b);
  KWS_LAUNCH_CHECK("dw_grad_finalize_batch_kernel");
  return KWS_OK;
}

extern "C" {

int kws_bn_stats_finalize(const float* stats_part, int n_tiles, int64_t count, int C, const float* gamma,
                          const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                          float* bn, float* scratch, void* stream) {
  KWS_REQUIRE(stats_part && gamma && beta && bn, "bn_stats_finalize: NULL pointer");
  KWS_REQUIRE(n_tiles > 0 && count > 0 && C > 0, "bn_stats_finalize: bad sizes");
  KWS_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_stats_finalize: moving stats must both be set");
  const float omm = (float)(1.0 - (double)momentum);
  KwsProfScope prof("bn_finalize", 0.0, 8.0 * n_tiles * C, (hipStream_t)stream);
#ifdef KWS_ABL_NO_FIN   // timing-only ablation (wrong results): what the fold launches on the dependency chain cost the step
  return KWS_OK;       // (round 5: 4.414 -> 4.233 ms, 181 us for 25 launches; profiles/r05_ablation_no_fold_launches.txt)
#endif
  KWS_TRY(pre_reduce(stats_part, n_tiles, 2 * C, scratch, (hipStream_t)stream, &stats_part, &n_tiles));
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((unsigned)ceil_div(C, FIN_CG)), dim3(FIN_CG * FIN_RG), 0, (hipStream_t)stream,
                     stats_part, n_tiles, 1.0 / (double)count, C, gamma, beta, eps, omm, moving_mean, moving_var, bn);
  KWS_LAUNCH_CHECK("bn_stats_finalize_kernel");
  return KWS_OK;
}

int kws_bn_infer_prepare(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                         float eps, int C, float* bn, void* stream) {
  KWS_REQUIRE(gamma && beta && moving_mean && moving_var && bn && C > 0, "bn_infer_prepare: bad arguments");
  hipLaunchKernelGGL(bn_infer_prepare_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     gamma, beta, moving_mean, moving_var, eps, C, bn);
  KWS_LAUNCH_CHECK("bn_infer_prepare_kernel");
  return KWS_OK;
}

// internal: kws_bn_infer_prepare for count <= KWS_BN_INFER_BATCH layers in one launch
int kws_bn_infer_prepare_batch(const float* const* gamma, const float* const* beta, const float* const* mm, const float* const* mv,
                               float eps, const int* C, float* const* bn, int count, hipStream_t stream) {
  KWS_REQUIRE(gamma && beta && mm && mv && C && bn && count > 0 && count <= KWS_BN_INFER_BATCH, "bn_infer_prepare_batch: bad arguments (count=%d)", count);
  BnInferBatch b;
  int maxC = 0;
  for (int i = 0; i < count; ++i) {
    KWS_REQUIRE(gamma[i] && beta[i] && mm[i] && mv[i] && bn[i] && C[i] > 0, "bn_infer_prepare_batch: bad entry %d", i);
    b.gamma[i] = gamma[i]; b.beta[i] = beta[i]; b.mm[i] = mm[i]; b.mv[i] = mv[i]; b.bn[i] = bn[i]; b.C[i] = C[i];
    maxC = C[i] > maxC ? C[i] : maxC;
  }
  b.eps = eps;
  hipLaunchKernelGGL(bn_infer_prepare_batch_kernel, dim3((unsigned)ceil_div(maxC, 256), (unsigned)count), dim3(256), 0, stream, b);
  KWS_LAUNCH_CHECK("bn_infer_prepare_batch_kernel");
  return KWS_OK;
}

int kws_bn_relu6_apply(const float* y, const float* bn, float* out, int64_t rows, int C, int relu6, void* stream) {
  KWS_REQUIRE(y && bn && out && rows > 0 && C > 0 && C % 4 == 0, "bn_relu6_apply: bad arguments");
  const int64_t n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_relu6_apply_kernel, dim3((unsigned)ceil_div64(n4, 256)), dim3(256), 0, (hipStream_t)stream, y,
                     bn, out, n4, C, relu6);
  KWS_LAUNCH_CHECK("bn_relu6_apply_kernel");
  return KWS_OK;
}

int kws_dw_bwd_finalize(const float* part, int n_parts, int64_t count, int C, float* dw, float* dgamma, float* dbeta,
                        float* coef, float* scratch, void* stream) {
  KWS_REQUIRE(part && n_parts > 0 && count > 0 && C > 0, "dw_bwd_finalize: bad arguments");
  KwsProfScope prof("bn_finalize", 0.0, 20.0 * n_parts * C, (hipStream_t)stream);
#ifdef KWS_ABL_NO_FIN
  return KWS_OK;
#endif
  KWS_TRY(pre_reduce(part, n_parts, 5 * C, scratch, (hipStream_t)stream, &part, &n_parts));
  hipLaunchKernelGGL(dw_bwd_finalize_kernel, dim3((unsigned)ceil_div(C, FIN_CG)), dim3(FIN_CG * FIN_RG), 0, (hipStream_t)stream, part,
                     n_parts, 1.0 / (double)count, C, dw, dgamma, dbeta, coef);
  KWS_LAUNCH_CHECK("dw_bwd_finalize_kernel");
  return KWS_OK;
}

int kws_bn_bwd_apply(float* g, const float* y, const float* bn, const float* gamma, const float* coef, int64_t rows,
                     int C, void* stream) {
  return kws_bn_bwd_apply_amax(g, y, bn, gamma, coef, rows, C, nullptr, (hipStream_t)stream);
}

// internal: as kws_bn_bwd_apply, and the |dy| maximum into amax (may be NULL)
int kws_bn_bwd_apply_amax(float* g, const float* y, const float* bn, const float* gamma, const float* coef, int64_t rows,
                          int C, unsigned* amax, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  KWS_REQUIRE(g && y && bn && gamma && coef && rows > 0 && C > 0 && C % 4 == 0, "bn_bwd_apply: bad arguments");
  const int64_t n4 = rows * C / 4;
  KwsProfScope prof("bn_bwd_apply", 6.0 * rows * C, 12.0 * rows * C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div64(n4, 256)), dim3(256), 0, (hipStream_t)stream, g, y,
                     bn, gamma, coef, n4, C, amax);
  KWS_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return KWS_OK;
}

}  // extern "C"
