// The per-channel folds of partial rows (BatchNorm statistics; depthwise-backward sums) as device functions of a block index,
// shared by bn.hip's own finalise kernels and - round 5 - by the CONSUMER kernels of their results (dwconv.hip): the fold runs on
// the first few workgroups of the consumer's grid, the others wait for a flag, and the launch between producer and consumer (a
// drain, a launch and a ramp-up on the step's dependency chain: 7.5 us each, 181 us per step by -DKWS_ABL_NO_FIN) is gone.
// Same code, same order of every sum: bit-identical to the stand-alone launches.
#pragma once
#include "internal.h"

namespace kws_fold {

// 256 threads = FIN_CG channels x FIN_RG row groups; up to 256 partial rows are summed directly (<= 16 per thread; more rows
// with only C/16 workgroups is latency-bound: 2048 rows took 35 us), row group r takes rows r, r+FIN_RG, ... and the groups are
// combined in order
constexpr int FIN_CG = 16, FIN_RG = 16;
#ifndef KWS_FIN_U
#define KWS_FIN_U 16
#endif
constexpr int FIN_U = KWS_FIN_U;   // rows per thread and trip
constexpr int FOLD_THREADS = FIN_CG * FIN_RG;

static __device__ float g_zero_fin[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct BnFold {      // part[n_tiles][2][C] -> bn[4C] = scale | shift | mean | rstd ; moving stats update in place
  const float* part; int n_tiles; double inv_count; int C;
  const float* gamma; const float* beta; float eps; float one_minus_momentum;
  float* moving_mean; float* moving_var; float* bn;
};
struct DwFold {      // part[n_parts][5][C] -> dgamma, dbeta, dw[3][C], coef[2C] = (sum g / n, sum g*xhat / n)
  const float* part; int n_parts; double inv_count; int C;
  float* dw; float* dgamma; float* dbeta; float* coef;
};
constexpr int bn_fold_red_doubles() { return 2 * FIN_RG * FIN_CG; }
constexpr int dw_fold_red_doubles() { return 5 * FIN_RG * FIN_CG; }

// A result that workgroups of the SAME launch read (fused form): stored at agent scope, i.e. written through this XCD's L2, so
// that the signal needs no L2 write-back (a __threadfence() per fold wave walked the L2: + 60 us per step, round 5).  SHARED =
// false: the stand-alone kernels' plain stores.
template <bool SHARED>
__device__ __forceinline__ void fold_store(float* p, float v) {
  if (SHARED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// `blk` = which group of FIN_CG channels; the first FOLD_THREADS threads of the workgroup work, every thread keeps the barrier
template <bool SHARED>
__device__ __forceinline__ void bn_stats_fold(const BnFold& a, const int blk, double* const red_) {
  double (*red)[FIN_RG][FIN_CG] = reinterpret_cast<double (*)[FIN_RG][FIN_CG]>(red_);
  const float* __restrict__ part = a.part;
  const int n_tiles = a.n_tiles, C = a.C;
  const bool on = threadIdx.x < FOLD_THREADS;
  const int cg = threadIdx.x % FIN_CG, rg = (threadIdx.x / FIN_CG) % FIN_RG;
  const int c = blk * FIN_CG + cg;
  double s = 0.0, ss = 0.0;
  if (on && c < C) {
    // FIN_U rows' loads in flight per trip - with <= 256 partial rows (the producers' caps) ONE trip: the kernel is a chain
    // of memory round trips (the rows were just written by other XCDs) and little else.  Rows past the end read a zero
    // buffer (address select, not a branch); the additions stay in ascending row order.
    for (int t = rg; t < n_tiles; t += FIN_U * FIN_RG) {
      float va[FIN_U], vb[FIN_U];
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int tt = t + u * FIN_RG;
        const float* src = tt < n_tiles ? part + (int64_t)tt * 2 * C + c : g_zero_fin;
        va[u] = src[0];
        vb[u] = src[tt < n_tiles ? C : 1];
      }
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        s += (double)va[u];
        ss += (double)vb[u];
      }
    }
  }
  if (on) {
    red[0][rg][cg] = s;
    red[1][rg][cg] = ss;
  }
  __syncthreads();
  if (on && rg == 0 && c < C) {
    s = 0.0;
    ss = 0.0;
    for (int r = 0; r < FIN_RG; ++r) {
      s += red[0][r][cg];
      ss += red[1][r][cg];
    }
    const double mean = s * a.inv_count;
    double var = ss * a.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean, varf = (float)var;
    const float scale = a.gamma[c] * rstd;
    fold_store<SHARED>(a.bn + c, scale);
    fold_store<SHARED>(a.bn + C + c, a.beta[c] - meanf * scale);
    fold_store<SHARED>(a.bn + 2 * C + c, meanf);
    fold_store<SHARED>(a.bn + 3 * C + c, rstd);
    if (a.moving_mean) {
      // AssignMovingAvg: m -= (m - batch) * (1 - momentum); biased variance
      a.moving_mean[c] = a.moving_mean[c] - (a.moving_mean[c] - meanf) * a.one_minus_momentum;
      a.moving_var[c] = a.moving_var[c] - (a.moving_var[c] - varf) * a.one_minus_momentum;
    }
  }
}

template <bool SHARED>
__device__ __forceinline__ void dw_bwd_fold(const DwFold& a, const int blk, double* const red_) {
  double (*red)[FIN_RG][FIN_CG] = reinterpret_cast<double (*)[FIN_RG][FIN_CG]>(red_);
  const float* __restrict__ part = a.part;
  const int n_parts = a.n_parts, C = a.C;
  const bool on = threadIdx.x < FOLD_THREADS;
  const int cg = threadIdx.x % FIN_CG, rg = (threadIdx.x / FIN_CG) % FIN_RG;
  const int c = blk * FIN_CG + cg;
  double s[5] = {0, 0, 0, 0, 0};
  if (on && c < C) {
    for (int t = rg; t < n_parts; t += FIN_U * FIN_RG) {   // as in bn_stats_fold: 5 FIN_U loads in flight per trip
      float v[FIN_U][5];
#pragma unroll
      for (int u = 0; u < FIN_U; ++u) {
        const int tt = t + u * FIN_RG;
        const bool ok = tt < n_parts;
        const float* src = ok ? part + (int64_t)tt * 5 * C + c : g_zero_fin;
#pragma unroll
        for (int q = 0; q < 5; ++q) v[u][q] = src[ok ? q * C : q];
      }
#pragma unroll
      for (int u = 0; u < FIN_U; ++u)
#pragma unroll
        for (int q = 0; q < 5; ++q) s[q] += (double)v[u][q];
    }
  }
  if (on) {
#pragma unroll
    for (int q = 0; q < 5; ++q) red[q][rg][cg] = s[q];
  }
  __syncthreads();
  if (on && rg == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      double acc = 0.0;
      for (int r = 0; r < FIN_RG; ++r) acc += red[q][r][cg];
      s[q] = acc;
    }
    if (a.dbeta) a.dbeta[c] = (float)s[0];
    if (a.dgamma) a.dgamma[c] = (float)s[1];
    if (a.coef) {
      fold_store<SHARED>(a.coef + c, (float)(s[0] * a.inv_count));
      fold_store<SHARED>(a.coef + C + c, (float)(s[1] * a.inv_count));
    }
    if (a.dw) {
      a.dw[c] = (float)s[2];
      a.dw[C + c] = (float)s[3];
      a.dw[2 * C + c] = (float)s[4];
    }
  }
}

// ---- the flag between the fold workgroups and the rest of a grid ------------------------------------------------------------
// The fold workgroups are the FIRST blocks of the grid (dispatched before any waiting block, so the wait cannot starve them).
// The results the waiting workgroups need are stored at agent scope (fold_store: written through this XCD's L2); every wave waits
// for its stores to be acknowledged, then one thread counts the workgroup in.  A waiting workgroup polls the counter with agent-scope loads
// (they bypass its XCD's L2) and only then touches the results - lines no workgroup of its XCD can have read in this launch.
// The wait is BOUNDED (~0.3 s): a grid must drain whatever happens (a fold that never signals would otherwise hang the GPU).
__device__ __forceinline__ void fold_signal(unsigned* flag) {
  __builtin_amdgcn_s_waitcnt(0);      // this wave's agent-scope stores have been acknowledged (vmcnt counts stores on gfx9)
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifndef KWS_FOLD_SLEEP0
#define KWS_FOLD_SLEEP0 0     // s_sleep units (64 cycles) before the first poll
#endif
#ifndef KWS_FOLD_SLEEP
#define KWS_FOLD_SLEEP 2      // ... between polls
#endif
__device__ __forceinline__ void fold_wait(const unsigned* flag, unsigned want) {
  if (threadIdx.x == 0) {
    int spins = 0;
    if (KWS_FOLD_SLEEP0) __builtin_amdgcn_s_sleep(KWS_FOLD_SLEEP0);
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 21)) __builtin_amdgcn_s_sleep(KWS_FOLD_SLEEP);
  }
  __syncthreads();
}

}  // namespace kws_fold
