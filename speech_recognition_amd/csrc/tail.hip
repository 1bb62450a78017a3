// Attention + pooling + classifier tail of conv_1d_time_sliced_with_attention_model
// (reference model.py:819-830), its loss (utils.py:87-108) and their backward, fused into one
// workgroup-per-clip kernel (SURVEY 8a rows a12, a13 and the matching part of a15).
//
//   x      = relu6(bn12(y12))                       [T, C]   (BN scale/shift applied on load)
//   att    = softmax(Dense_T(Dropout(Flatten(x))))  [T]
//   feat   = [max_t(x * att) ; mean_t(x)]           [2C]
//   p      = softmax(Dense_NC(Dropout(feat)))       [NC]
// The per-clip tile (T*C <= 9216 floats) stays in LDS between forward and backward.  Weight
// gradients need a sum over the batch: the kernel writes the per-clip operands (dropped inputs and
// logit gradients) and `small_wgrad_kernel` reduces them in a fixed order.
#include "internal.h"

namespace {

constexpr int MAXT = 16;
constexpr int MAXNC = 64;

struct TailArgs {
  const float* y;       // [B, T, C] pre-BN output of the last pointwise conv
  const float* bn;      // [4C]
  const float* W1;      // [T*C, T]
  const float* b1;      // [T]
  const float* W2;      // [2C, NC]
  const float* labels;  // [B, NC] one-hot (train)
  float* probs;         // [B, NC]
  // train outputs
  float* g;             // [B, T, C] masked gradient wrt bn12 output
  float* part;          // [B][5][C] BN-backward partial sums (slots 2..4 zero)
  float* xd;            // [B, T*C] dropped flatten (operand of dW1)
  float* fd;            // [B, 2C]  dropped features (operand of dW2)
  float* dl1;           // [B, T]
  float* dl2;           // [B, NC]
  float* per_loss;      // [B]
  float* per_correct;   // [B]
  float* att_out;       // [B, T] attention weights (debug / parity view), may be NULL
  int B, T, C, NC;
  uint32_t key1, key2, thresh;
  float inv_keep;
  float label_smoothing;
  float inv_loss_batch;
  int64_t row_offset;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// all-lanes reductions of one wave (xor butterfly: the same fixed order on every lane)
__device__ __forceinline__ float wave_all_sum(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_all_max(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_all_min_i(int v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}
// softmax of n <= 64 values in LDS by the first wave, one value per lane (the single-thread loop this replaces was a chain
// of n dependent expf / divisions on the kernel's critical path)
__device__ __forceinline__ void wave_softmax(const float* in, float* out, int n) {
  const int lane = threadIdx.x;
  const float v = lane < n ? in[lane] : -INFINITY;
  const float m = wave_all_max(v);
  const float e = lane < n ? expf(v - m) : 0.f;
  const float den = wave_all_sum(e);
  if (lane < n) out[lane] = e / den;
}

// sums `nv` (<= NV) per-thread values over the 256-thread block; result broadcast through out[] (LDS).
// vals is indexed with compile-time constants only (runtime-indexed register arrays go to scratch).
template <int NV>
__device__ void block_sum(const float (&vals)[NV], int nv, float* scratch /*[4*MAXNC]*/, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (v < nv) {
      const float s = wave_sum(vals[v]);
      if (lane == 0) scratch[wave * MAXNC + v] = s;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nv) {
    const int v = threadIdx.x;
    out[v] = ((scratch[v] + scratch[MAXNC + v]) + scratch[2 * MAXNC + v]) + scratch[3 * MAXNC + v];
  }
  __syncthreads();
}

// TT > 0: the attention width T is known at compile time (9 for the 16000-sample model), which lets both
// passes over the [T*C, T] attention kernel W1 read it as whole 16-byte vectors, 4 rows (= T float4) per
// thread and step, consecutive threads on consecutive 16*T-byte chunks.  The generic path (TT = 0) walks W1
// with 4-byte strided loads: at T = 9 every wave-load touched 18 cache lines and the 1024 workgroups pulled
// ~1.8 GB through L1 per step (0.35 ms; this kernel's HBM-side traffic is only ~60 MB).
template <bool TRAIN, int TT>
__global__ __launch_bounds__(256, 4) void ts_tail_kernel(TailArgs a) {   // 4 waves per SIMD: every clip of a 1024 batch resident
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = a.T, C = a.C, NC = a.NC, TC = T * C;
  float* xs = lds;                 // [TC]   x = relu6(bn(y)); overwritten in place by dx in the backward half
  float* feat = xs + TC;           // [2C]   (one [TC] tile keeps the workgroup at 30 KB of LDS: all 1024 clips resident)
  float* dfeat = feat + 2 * C;     // [2C]
  float* scratch = dfeat + 2 * C;  // [4*MAXNC]
  float* l1 = scratch + 4 * MAXNC; // [MAXT] logits1 / att
  float* att = l1 + MAXT;          // [MAXT]
  float* dl1 = att + MAXT;         // [MAXT]
  float* datt = dl1 + MAXT;        // [MAXT]
  float* l2 = datt + MAXT;         // [MAXNC]
  float* pp = l2 + MAXNC;          // [MAXNC]
  float* dl2 = pp + MAXNC;         // [MAXNC]
  float* part16 = dl2 + MAXNC;     // [16*16] partial dot products

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const float* yb = a.y + (int64_t)b * TC;
  const uint32_t row = (uint32_t)(a.row_offset + b);

  // ---- x = relu6(bn(y)), dropout 1 -----------------------------------------------------------
  // The kernel is one serial chain per clip with every clip resident at once: its duration is the chain's LATENCY.  Global
  // loads are therefore issued in batches (a "load, use, next element" loop costs one memory round trip per iteration: the
  // 18 of this loop and the 18 of the last one were most of the kernel's 62 us) and the few scalars the single-thread
  // sections need (labels, attention bias) are fetched into LDS here, under the first batch.
  float* lab = part16 + 256;       // [MAXNC] this clip's labels
  float* b1s = lab + MAXNC;        // [MAXT]
  if (TRAIN && tid < NC) lab[tid] = a.labels[(int64_t)b * NC + tid];
  if (tid >= 64 && tid < 64 + T) b1s[tid - 64] = a.b1[tid - 64];
  if (TT > 0) {                    // (T C and C are multiples of 4: host-checked for this instance)
    constexpr int NB1 = 5;         // 16-byte loads in flight per thread
    const int n4 = TC >> 2;
    for (int i0 = 0; i0 < n4; i0 += 256 * NB1) {
      float4 yv[NB1], scv[NB1], shv[NB1];
#pragma unroll
      for (int k = 0; k < NB1; ++k) {
        const int i = i0 + 256 * k + tid;
        const int ii = i < n4 ? i : 0;
        yv[k] = reinterpret_cast<const float4*>(yb)[ii];
        scv[k] = *reinterpret_cast<const float4*>(a.bn + (4 * ii) % C);
        shv[k] = *reinterpret_cast<const float4*>(a.bn + C + (4 * ii) % C);
      }
#pragma unroll
      for (int k = 0; k < NB1; ++k) {
        const int i = i0 + 256 * k + tid;
        if (i >= n4) continue;
        const int e = 4 * i;
        const float4 sc = scv[k], sh = shv[k];
        const float4 x = make_float4(relu6f(fmaf(yv[k].x, sc.x, sh.x)), relu6f(fmaf(yv[k].y, sc.y, sh.y)),
                                     relu6f(fmaf(yv[k].z, sc.z, sh.z)), relu6f(fmaf(yv[k].w, sc.w, sh.w)));
        *reinterpret_cast<float4*>(xs + e) = x;
        if (TRAIN) {
          const uint32_t id = row * (uint32_t)TC + (uint32_t)e;
          const float4 d = make_float4(kws_keep(id, a.key1, a.thresh) ? x.x * a.inv_keep : 0.f,
                                       kws_keep(id + 1, a.key1, a.thresh) ? x.y * a.inv_keep : 0.f,
                                       kws_keep(id + 2, a.key1, a.thresh) ? x.z * a.inv_keep : 0.f,
                                       kws_keep(id + 3, a.key1, a.thresh) ? x.w * a.inv_keep : 0.f);
          *reinterpret_cast<float4*>(a.xd + (int64_t)b * TC + e) = d;
        }
      }
    }
  } else {
  for (int e = tid; e < TC; e += 256) {
    const int c = e % C;
    const float x = relu6f(fmaf(yb[e], a.bn[c], a.bn[C + c]));
    xs[e] = x;
    if (TRAIN)   // the dropped activations are only stored for the dW1 reduction; the logits recompute the mask
      a.xd[(int64_t)b * TC + e] = kws_keep(row * (uint32_t)TC + (uint32_t)e, a.key1, a.thresh) ? x * a.inv_keep : 0.f;
  }
  }
  auto dropped = [&](float x, int e) -> float {
    if (!TRAIN) return x;
    return kws_keep(row * (uint32_t)TC + (uint32_t)e, a.key1, a.thresh) ? x * a.inv_keep : 0.f;
  };
  __syncthreads();
  // ---- logits1 = dropout(x) . W1 + b1 ------------------------------------------------------------------
  {
    if (TT > 0) {
      // thread <- row groups rg, rg+256, ...: 4 rows of W1 = TT float4, 4 x values = one LDS float4
      constexpr int TV = TT > 0 ? TT : 1;
      float pl[TV];
#pragma unroll
      for (int t = 0; t < TV; ++t) pl[t] = 0.f;
      for (int rg = tid; rg < TC / 4; rg += 256) {
        const float4* wp = reinterpret_cast<const float4*>(a.W1 + (int64_t)rg * 4 * TV);
        float wv[4 * TV];
#pragma unroll
        for (int i = 0; i < TV; ++i) {
          const float4 w4 = wp[i];
          wv[4 * i] = w4.x; wv[4 * i + 1] = w4.y; wv[4 * i + 2] = w4.z; wv[4 * i + 3] = w4.w;
        }
        const float4 x4 = *reinterpret_cast<const float4*>(xs + rg * 4);
        const float xr[4] = {dropped(x4.x, rg * 4), dropped(x4.y, rg * 4 + 1), dropped(x4.z, rg * 4 + 2),
                             dropped(x4.w, rg * 4 + 3)};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < TV; ++t) pl[t] = fmaf(xr[r], wv[r * TV + t], pl[t]);
      }
      block_sum(pl, TV, scratch, l1);
      if (tid < T) l1[tid] += b1s[tid];
      __syncthreads();
    } else {
      // generic T: thread (t = tid%16, slice = tid/16)
      const int t = tid & 15, sl = tid >> 4;
      float s = 0.f;
      if (t < T)
        for (int e = sl; e < TC; e += 16) s = fmaf(dropped(xs[e], e), a.W1[(int64_t)e * T + t], s);
      part16[sl * 16 + t] = s;
      __syncthreads();
      if (tid < T) {
        float acc = b1s[tid];
        for (int k = 0; k < 16; ++k) acc += part16[k * 16 + tid];
        l1[tid] = acc;
      }
      __syncthreads();
    }
    if (tid < 64) wave_softmax(l1, att, T);
    __syncthreads();
    if (TRAIN && a.att_out != nullptr && tid < T) a.att_out[(int64_t)b * T + tid] = att[tid];
  }
  // ---- pooling: feat = [max_t x*att ; mean_t x], dropout 2 -----------------------------------
  for (int c = tid; c < C; c += 256) {
    float mx = xs[c] * att[0], sm = xs[c];
    for (int t = 1; t < T; ++t) {
      mx = fmaxf(mx, xs[t * C + c] * att[t]);
      sm += xs[t * C + c];
    }
    feat[c] = mx;
    feat[C + c] = sm / (float)T;
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    float f = feat[i];
    if (TRAIN) {
      f = kws_keep(row * (uint32_t)(2 * C) + (uint32_t)i, a.key2, a.thresh) ? f * a.inv_keep : 0.f;
      a.fd[(int64_t)b * 2 * C + i] = f;
    }
    dfeat[i] = f;  // dropped features (forward use); overwritten by the gradient below
  }
  __syncthreads();
  // ---- logits2 = fd . W2 : thread (k = tid%64, slice = tid/64) -------------------------------
  {
    if ((NC & 3) == 0 && NC <= 16) {
      // thread <- rows i, i+256, ...: a row of W2 is NC/4 whole 16-byte vectors (the per-class walk below issued
      // 256 dependent 4-byte loads per thread on 12 of 64 lanes: 153 k of this kernel's 250 k cycles)
      float pl[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) pl[q] = 0.f;
      // four rows of W2 (<= 16 vectors) per thread and trip, all requested before the first is used
      const int nv = NC >> 2;
      for (int i0 = tid; i0 < 2 * C; i0 += 4 * 256) {
        float4 w4[4][4];
        float f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 256 * u;
          const bool ok = i < 2 * C;
          const float4* wp = reinterpret_cast<const float4*>(a.W2 + (int64_t)(ok ? i : 0) * NC);
#pragma unroll
          for (int v = 0; v < 4; ++v) w4[u][v] = wp[v < nv ? v : 0];
          f[u] = dfeat[ok ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + 256 * u >= 2 * C) continue;       // (the rows that exist, in row order)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (v < nv) {
              pl[4 * v] = fmaf(f[u], w4[u][v].x, pl[4 * v]);
              pl[4 * v + 1] = fmaf(f[u], w4[u][v].y, pl[4 * v + 1]);
              pl[4 * v + 2] = fmaf(f[u], w4[u][v].z, pl[4 * v + 2]);
              pl[4 * v + 3] = fmaf(f[u], w4[u][v].w, pl[4 * v + 3]);
            }
        }
      }
      block_sum(pl, NC, scratch, l2);
    } else {
      const int k = tid & 63, sl = tid >> 6;
      float s = 0.f;
      if (k < NC)
        for (int i = sl; i < 2 * C; i += 4) s = fmaf(dfeat[i], a.W2[(int64_t)i * NC + k], s);
      scratch[sl * MAXNC + k] = s;
      __syncthreads();
      if (tid < NC) l2[tid] = ((scratch[tid] + scratch[MAXNC + tid]) + scratch[2 * MAXNC + tid]) + scratch[3 * MAXNC + tid];
      __syncthreads();
    }
    if (tid < 64) wave_softmax(l2, pp, NC);
    __syncthreads();
    if (tid < NC) a.probs[(int64_t)b * NC + tid] = pp[tid];
  }
  if (!TRAIN) return;

  // ---- loss: softmax-CE on log(clip(p)) with label smoothing (utils.py:100-108) --------------
  if (tid < 64) {   // one class per lane (NC <= 64); sums and the two arg-maxima as wave reductions
    const float eps = 1e-7f;
    const bool in = tid < NC;
    const float pq = in ? pp[tid] : 0.f, yq = in ? lab[tid] : 0.f;
    const float pc = fminf(fmaxf(pq, eps), 1.f - eps);
    const float S = wave_all_sum(in ? pc : 0.f);
    const float logS = logf(S);
    const float ysm = in ? yq * (1.f - a.label_smoothing) + a.label_smoothing / (float)NC : 0.f;
    const float ysum = wave_all_sum(ysm);
    const float loss = -wave_all_sum(in ? ysm * (logf(pc) - logS) : 0.f);
    // first index of the maximum, as the serial "if (v[q] > v[best]) best = q" walk finds it
    const float pmax = wave_all_max(in ? pq : -INFINITY), ymax = wave_all_max(in ? yq : -INFINITY);
    const int am_p = wave_all_min_i(in && pq == pmax ? tid : 64), am_y = wave_all_min_i(in && yq == ymax ? tid : 64);
    if (tid == 0) {
      a.per_loss[b] = loss;
      a.per_correct[b] = (am_p == am_y) ? 1.f : 0.f;
    }
    // dL/dp (clip passes gradient inside [eps, 1-eps]), then softmax backward
    const float inside = (pq >= eps && pq <= 1.f - eps) ? 1.f : 0.f;
    const float dp = in ? (-ysm / pc + ysum / S) * a.inv_loss_batch * inside : 0.f;
    const float dot = wave_all_sum(dp * pq);
    if (in) dl2[tid] = pq * (dp - dot);
  }
  __syncthreads();
  if (tid < NC) a.dl2[(int64_t)b * NC + tid] = dl2[tid];
  // ---- dfeat = (W2 . dl2) * mask2 / keep ------------------------------------------------------
  if ((NC & 3) == 0 && NC <= 16) {
    const int nv = NC >> 2;
    for (int i0 = tid; i0 < 2 * C; i0 += 4 * 256) {
      float4 w4[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        const float4* wp = reinterpret_cast<const float4*>(a.W2 + (int64_t)(i < 2 * C ? i : 0) * NC);
#pragma unroll
        for (int v = 0; v < 4; ++v) w4[u][v] = wp[v < nv ? v : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        if (i >= 2 * C) continue;
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (v < nv) {
            s = fmaf(w4[u][v].x, dl2[4 * v], s);
            s = fmaf(w4[u][v].y, dl2[4 * v + 1], s);
            s = fmaf(w4[u][v].z, dl2[4 * v + 2], s);
            s = fmaf(w4[u][v].w, dl2[4 * v + 3], s);
          }
        const bool keep = kws_keep(row * (uint32_t)(2 * C) + (uint32_t)i, a.key2, a.thresh);
        dfeat[i] = keep ? s * a.inv_keep : 0.f;
      }
    }
  } else {
    for (int i = tid; i < 2 * C; i += 256) {
      float s = 0.f;
      for (int q = 0; q < NC; ++q) s = fmaf(a.W2[(int64_t)i * NC + q], dl2[q], s);
      const bool keep = kws_keep(row * (uint32_t)(2 * C) + (uint32_t)i, a.key2, a.thresh);
      dfeat[i] = keep ? s * a.inv_keep : 0.f;
    }
  }
  __syncthreads();
  // ---- pooling backward: reduce_max splits its gradient equally among ties -------------------
  float dattl[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) dattl[t] = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float mx = feat[c];
    int n = 0;
    for (int t = 0; t < T; ++t) n += (xs[t * C + c] * att[t] == mx) ? 1 : 0;
    const float share = dfeat[c] / (float)n;
    const float davg = dfeat[C + c] / (float)T;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < T) {
        const float xv = xs[t * C + c];
        const float dxa = (xv * att[t] == mx) ? share : 0.f;
        xs[t * C + c] = dxa * att[t] + davg;  // in place: x at this position is not needed again
        dattl[t] += dxa * xv;
      }
    }
  }
  block_sum(dattl, T, scratch, datt);
  if (tid < 64) {
    const float av = tid < T ? att[tid] : 0.f, dv = tid < T ? datt[tid] : 0.f;
    const float dot = wave_all_sum(av * dv);
    if (tid < T) dl1[tid] = av * (dv - dot);
  }
  __syncthreads();
  if (tid < T) a.dl1[(int64_t)b * T + tid] = dl1[tid];
  // ---- dx += (W1 . dl1) * mask1 / keep ; g = dx * relu6'(pre) ; BN-backward partial sums ------
  if (TT > 0) {
    // coalesced pass over W1 (same 4-row groups as the forward): dx (held in xs) += dropout-masked W1 . dl1
    constexpr int TV = TT > 0 ? TT : 1;
    float dv[TV];
#pragma unroll
    for (int t = 0; t < TV; ++t) dv[t] = dl1[t];
    for (int rg = tid; rg < TC / 4; rg += 256) {
      const float4* wp = reinterpret_cast<const float4*>(a.W1 + (int64_t)rg * 4 * TV);
      float wv[4 * TV];
#pragma unroll
      for (int i = 0; i < TV; ++i) {
        const float4 w4 = wp[i];
        wv[4 * i] = w4.x; wv[4 * i + 1] = w4.y; wv[4 * i + 2] = w4.z; wv[4 * i + 3] = w4.w;
      }
      float sr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < TV; ++q) s = fmaf(wv[r * TV + q], dv[q], s);
        const bool keep = kws_keep(row * (uint32_t)TC + (uint32_t)(rg * 4 + r), a.key1, a.thresh);
        sr[r] = keep ? s * a.inv_keep : 0.f;
      }
      float4 d4 = *reinterpret_cast<float4*>(xs + rg * 4);
      d4.x += sr[0]; d4.y += sr[1]; d4.z += sr[2]; d4.w += sr[3];
      *reinterpret_cast<float4*>(xs + rg * 4) = d4;
    }
    __syncthreads();
  }
  float* gb = a.g + (int64_t)b * TC;
  if (TT > 0) {
    // as the generic loop below, the T pre-activations of a channel loaded in one batch (the sums keep their order in t)
    constexpr int TV = TT > 0 ? TT : 1;
    for (int c = tid; c < C; c += 256) {
      float yv[TV];
#pragma unroll
      for (int t = 0; t < TV; ++t) yv[t] = yb[t * C + c];
      const float sc = a.bn[c], sh = a.bn[C + c], mean = a.bn[2 * C + c], rstd = a.bn[3 * C + c];
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int t = 0; t < TV; ++t) {
        const int e = t * C + c;
        const float pre = fmaf(yv[t], sc, sh);
        const float gv = (pre > 0.f && pre <= 6.f) ? xs[e] : 0.f;
        gb[e] = gv;
        sg += gv;
        sgx = fmaf(gv, (yv[t] - mean) * rstd, sgx);
      }
      float* pb = a.part + (int64_t)b * 5 * C;
      pb[c] = sg;
      pb[C + c] = sgx;
      pb[2 * C + c] = 0.f;
      pb[3 * C + c] = 0.f;
      pb[4 * C + c] = 0.f;
    }
    return;
  }
  for (int c = tid; c < C; c += 256) {
    const float sc = a.bn[c], sh = a.bn[C + c], mean = a.bn[2 * C + c], rstd = a.bn[3 * C + c];
    float sg = 0.f, sgx = 0.f;
    for (int t = 0; t < T; ++t) {
      const int e = t * C + c;
      float dx = xs[e];
      float s = 0.f;
      for (int q = 0; q < T; ++q) s = fmaf(a.W1[(int64_t)e * T + q], dl1[q], s);
      const bool keep = kws_keep(row * (uint32_t)TC + (uint32_t)e, a.key1, a.thresh);
      dx += keep ? s * a.inv_keep : 0.f;
      const float yv = yb[e];
      const float pre = fmaf(yv, sc, sh);
      const float gv = (pre > 0.f && pre <= 6.f) ? dx : 0.f;
      gb[e] = gv;
      sg += gv;
      sgx = fmaf(gv, (yv - mean) * rstd, sgx);
    }
    float* pb = a.part + (int64_t)b * 5 * C;
    pb[c] = sg;
    pb[C + c] = sgx;
    pb[2 * C + c] = 0.f;
    pb[3 * C + c] = 0.f;
    pb[4 * C + c] = 0.f;
  }
}

// out[K, N] = sum_b X[b, K]^T D[b, N]  (N small); optional bias grad out_b[N] = sum_b D[b, N].
// The batch is cut into gridDim.y slices (one partial [K*N] slab each, summed afterwards in slice order)
// so that B*K*N/256 workgroups share the work instead of K*N/256 serial loops over the whole batch.
__global__ __launch_bounds__(256) void small_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ D,
                                                          float* __restrict__ out, int B, int K, int N, int rows_per) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)K * N) return;
  const int i = (int)(id / N), k = (int)(id - (int64_t)i * N);
  const int b0 = blockIdx.y * rows_per;
  int b1 = b0 + rows_per;
  if (b1 > B) b1 = B;
  float s = 0.f;
  for (int b = b0; b < b1; ++b) s = fmaf(X[(int64_t)b * K + i], D[(int64_t)b * N + k], s);
  out[(int64_t)blockIdx.y * K * N + id] = s;
}
// N <= NMAX: one thread per input feature i keeps its N outputs in registers and walks its batch slice once - X is read
// coalesced and exactly once (the generic kernel reads every X element N times, N threads apart), D rows are uniform
// across the workgroup.  Same summation order per output as the generic kernel (ascending b inside a slice).
constexpr int KWS_SMALL_WGRAD_ROWS = 64;   // most rows a slice may hold (B / KWS_SMALL_WGRAD_SLICES, checked by the launcher)
template <int NMAX>
__device__ __forceinline__ void small_wgrad_rows_body(const float* __restrict__ X, const float* __restrict__ D,
                                                      float* __restrict__ out, int B, int K, int N, int rows_per, const int bx,
                                                      const int by) {
  const int i = bx * 256 + threadIdx.x;
  const int b0 = by * rows_per;
  int b1 = b0 + rows_per;
  if (b1 > B) b1 = B;
  __shared__ float sD[KWS_SMALL_WGRAD_ROWS * NMAX];   // this slice's D rows, zero padded to NMAX columns
  for (int j = threadIdx.x; j < KWS_SMALL_WGRAD_ROWS * NMAX; j += 256) {
    const int r = j / NMAX, k = j - r * NMAX;
    sD[j] = (b0 + r < b1 && k < N) ? D[(int64_t)(b0 + r) * N + k] : 0.f;
  }
  __syncthreads();
  float acc[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; ++k) acc[k] = 0.f;
  if (i < K) {
    const float* xp = X + (int64_t)b0 * K + i;
    const int nb = b1 - b0;
    for (int r0 = 0; r0 < nb; r0 += 8) {       // eight independent loads in flight, then their FMAs in row order
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = r0 + u < nb ? xp[(int64_t)(r0 + u) * K] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NMAX; ++k) acc[k] = fmaf(x[u], sD[(r0 + u) * NMAX + k], acc[k]);
    }
    float* o = out + (int64_t)by * K * N + (int64_t)i * N;
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k < N) o[k] = acc[k];
  }
}
template <int NMAX>
__global__ __launch_bounds__(256) void small_wgrad_rows_kernel(const float* __restrict__ X, const float* __restrict__ D,
                                                               float* __restrict__ out, int B, int K, int N,
                                                               int rows_per) {
  small_wgrad_rows_body<NMAX>(X, D, out, B, K, N, rows_per, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                       int S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = in[i];
  for (int k = 1; k < S; ++k) s += in[(int64_t)k * n + i];
  out[i] = s;
}
// out[k] = sum_b D[b, k]: KP columns x NT/KP row groups, each group sums its rows b = g, g+G, ... in ascending order over four
// interleaved accumulators, the groups are combined in ascending order (fixed order, one workgroup).  Round 6: 1024 threads and
// KP = the narrowest of 16 / 32 / 64 that holds N - the 256-thread form gave config C3's 32-column bias 4 row groups of 512 rows
// each: 27 us of dependent loads.
template <int KP, int NT>
__device__ __forceinline__ void colsum_body(const float* __restrict__ D, float* __restrict__ out, int B, int N) {
  constexpr int G = NT / KP;
  __shared__ float red[G][KP];
  const int k = threadIdx.x % KP, g = threadIdx.x / KP;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (k < N) {
    int b = g;
    for (; b + 3 * G < B; b += 4 * G) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = D[(int64_t)(b + q * G) * N + k];
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += v[q];
    }
    for (; b < B; b += G) s[0] += D[(int64_t)b * N + k];
  }
  red[g][k] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (g == 0 && k < N) {
    float t = red[0][k];
    for (int q = 1; q < G; ++q) t += red[q][k];
    out[k] = t;
  }
}
template <int KP, int NT>
__global__ __launch_bounds__(NT) void colsum_kernel(const float* __restrict__ D, float* __restrict__ out, int B, int N) {
  colsum_body<KP, NT>(D, out, B, N);
}
// metrics[0] = sum per_loss (double accumulation), metrics[1] = sum per_correct: thread t sums elements
// t, t+256, ... and the 256 partials are combined in ascending order (fixed order, one workgroup).
__device__ __forceinline__ void metrics_body(const float* per_loss, const float* per_correct, int B, float* metrics) {
  __shared__ double rl[256];
  __shared__ float rc[256];
  double sl = 0.0;
  float sc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    sl += (double)per_loss[b];
    sc += per_correct[b];
  }
  rl[threadIdx.x] = sl;
  rc[threadIdx.x] = sc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      rl[threadIdx.x] += rl[threadIdx.x + w];
      rc[threadIdx.x] += rc[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    metrics[0] = (float)rl[0];
    metrics[1] = rc[0];
    metrics[2] = 0.f;
    metrics[3] = 0.f;
  }
}
__global__ __launch_bounds__(256) void metrics_kernel(const float* per_loss, const float* per_correct, int B,
                                                      float* metrics) {
  metrics_body(per_loss, per_correct, B, metrics);
}

// Round 4: the six small launches that follow the tail kernel off the dependency chain as ONE grid - slices of the classifier's
// weight gradient (blocks [0, gx2 S)), slices of the attention dense layer's (the next gx1 S), its bias gradient (one block),
// the batch metrics (one block).  Each part is the body of the kernel that did it alone: same arithmetic, same order.
struct TailPost {
  kws_tail_post_args a;
  int rows_per, S, gx2, gx1;
};
__global__ __launch_bounds__(256) void tail_post_kernel(TailPost p) {
  int b = blockIdx.x;
  if (b < p.gx2 * p.S) {
    small_wgrad_rows_body<16>(p.a.X2, p.a.D2, p.a.ws2, p.a.B, p.a.K2, p.a.N2, p.rows_per, b % p.gx2, b / p.gx2);
    return;
  }
  b -= p.gx2 * p.S;
  if (b < p.gx1 * p.S) {
    small_wgrad_rows_body<16>(p.a.X1, p.a.D1, p.a.ws1, p.a.B, p.a.K1, p.a.N1, p.rows_per, b % p.gx1, b / p.gx1);
    return;
  }
  b -= p.gx1 * p.S;
  if (b == 0) metrics_body(p.a.per_loss, p.a.per_correct, p.a.B, p.a.metrics);
  else colsum_body<16, 256>(p.a.D1, p.a.bias1, p.a.B, p.a.N1);
}

}  // namespace

int kws_ts_tail_launch(const kws_ts_tail_args* p, hipStream_t st) {
  KWS_REQUIRE(p->T > 0 && p->T <= MAXT && p->NC > 0 && p->NC <= MAXNC && p->C > 0, "ts_tail: bad shape T=%d NC=%d",
              p->T, p->NC);
  const int TC = p->T * p->C;
  const size_t lds_floats = (size_t)TC + 4 * p->C + 4 * MAXNC + 4 * MAXT + 3 * MAXNC + 256 + MAXNC + MAXT;
  KWS_REQUIRE(lds_floats * 4 <= 160 * 1024, "ts_tail: T*C=%d does not fit LDS", TC);
  TailArgs a{};
  a.y = p->y; a.bn = p->bn; a.W1 = p->W1; a.b1 = p->b1; a.W2 = p->W2; a.labels = p->labels; a.probs = p->probs;
  a.g = p->g; a.part = p->part; a.xd = p->xd; a.fd = p->fd; a.dl1 = p->dl1; a.dl2 = p->dl2;
  a.per_loss = p->per_loss; a.per_correct = p->per_correct; a.att_out = p->att;
  a.B = p->B; a.T = p->T; a.C = p->C; a.NC = p->NC;
  a.key1 = kws_dropout_key(p->seed, p->step, 1);
  a.key2 = kws_dropout_key(p->seed, p->step, 2);
  a.thresh = kws_dropout_threshold(p->keep_prob);
  a.inv_keep = (float)(1.0 / (double)p->keep_prob);
  a.label_smoothing = p->label_smoothing;
  a.inv_loss_batch = 1.0f / (float)p->loss_batch;
  a.row_offset = p->row_offset;
  KwsProfScope prof(p->train ? "tail_train" : "tail_infer", 4.0 * p->B * TC * p->T * (p->train ? 2 : 1), 4.0 * p->B * TC * (p->train ? 4 : 1), st);
  // T = 9 (16000-sample input) takes the vectorised W1 passes; W1 rows must then be 16-byte aligned in
  // groups of 4 (T*C % 4 == 0 and an aligned base, both true for the flat parameter buffer)
  const bool fast9 = p->T == 9 && p->C % 4 == 0 && ((reinterpret_cast<uintptr_t>(p->W1) | reinterpret_cast<uintptr_t>(p->y) |
                                                     reinterpret_cast<uintptr_t>(p->bn) | reinterpret_cast<uintptr_t>(p->xd)) & 15) == 0;
  const void* fn = p->train ? (fast9 ? reinterpret_cast<const void*>(&ts_tail_kernel<true, 9>)
                                     : reinterpret_cast<const void*>(&ts_tail_kernel<true, 0>))
                            : (fast9 ? reinterpret_cast<const void*>(&ts_tail_kernel<false, 9>)
                                     : reinterpret_cast<const void*>(&ts_tail_kernel<false, 0>));
  if (lds_floats * 4 > 64 * 1024)
    KWS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_floats * 4)));
  if (p->train) {
    if (fast9) hipLaunchKernelGGL((ts_tail_kernel<true, 9>), dim3((unsigned)p->B), dim3(256), lds_floats * 4, st, a);
    else hipLaunchKernelGGL((ts_tail_kernel<true, 0>), dim3((unsigned)p->B), dim3(256), lds_floats * 4, st, a);
  } else {
    if (fast9) hipLaunchKernelGGL((ts_tail_kernel<false, 9>), dim3((unsigned)p->B), dim3(256), lds_floats * 4, st, a);
    else hipLaunchKernelGGL((ts_tail_kernel<false, 0>), dim3((unsigned)p->B), dim3(256), lds_floats * 4, st, a);
  }
  KWS_LAUNCH_CHECK("ts_tail_kernel");
  return KWS_OK;
}

int kws_small_wgrad_launch(const float* X, const float* D, float* out, float* out_bias, int B, int K, int N,
                           float* scratch, hipStream_t st) {
  const int64_t n = (int64_t)K * N;
  KwsProfScope prof("small_wgrad", 2.0 * B * K * N, 4.0 * ((double)B * K + (double)B * N + (double)K * N), st);
  int S = scratch ? KWS_SMALL_WGRAD_SLICES : 1;
  if (S > B) S = B;
  const int rows_per = ceil_div(B, S);
  S = ceil_div(B, rows_per);
  float* dst = S > 1 ? scratch : out;
  if (N <= 16 && rows_per <= KWS_SMALL_WGRAD_ROWS)
    hipLaunchKernelGGL(small_wgrad_rows_kernel<16>, dim3((unsigned)ceil_div(K, 256), (unsigned)S), dim3(256), 0, st, X, D, dst,
                       B, K, N, rows_per);
  else
    hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)ceil_div64(n, 256), (unsigned)S), dim3(256), 0, st, X, D, dst, B,
                       K, N, rows_per);
  KWS_LAUNCH_CHECK("small_wgrad_kernel");
  if (S > 1) {
    if (n % 4 == 0) {   // four slab groups per column in parallel (fixed order), 16-byte loads
      KWS_TRY(kws_reduce_slabs_f32(scratch, out, n, S, st));
    } else {
      hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, scratch, out, n, S);
      KWS_LAUNCH_CHECK("slab_sum_kernel");
    }
  }
  if (out_bias) {
    KWS_REQUIRE(N <= 64, "small_wgrad: N=%d > 64", N);
    // (N <= 16 keeps the 256-thread geometry: tail_post_kernel runs the same body in its 256-thread grid, bit for bit)
    if (N <= 16) hipLaunchKernelGGL((colsum_kernel<16, 256>), dim3(1), dim3(256), 0, st, D, out_bias, B, N);
    else if (N <= 32) hipLaunchKernelGGL((colsum_kernel<32, 1024>), dim3(1), dim3(1024), 0, st, D, out_bias, B, N);
    else hipLaunchKernelGGL((colsum_kernel<64, 1024>), dim3(1), dim3(1024), 0, st, D, out_bias, B, N);
    KWS_LAUNCH_CHECK("colsum_kernel");
  }
  return KWS_OK;
}

int kws_tail_post_launch(const kws_tail_post_args* a, int* S_out, hipStream_t st) {
  KWS_REQUIRE(a && S_out && a->X2 && a->D2 && a->ws2 && a->X1 && a->D1 && a->ws1 && a->per_loss && a->per_correct && a->metrics && a->B > 0,
              "tail_post: bad arguments");
  int S = KWS_SMALL_WGRAD_SLICES;
  if (S > a->B) S = a->B;
  const int rows_per = ceil_div(a->B, S);
  S = ceil_div(a->B, rows_per);
  // the slices of kws_small_wgrad_launch's fast kernel, summed later by a slab batch (16-byte columns): anything else takes the six launches
  if (S <= 1 || a->N1 > 16 || a->N2 > 16 || rows_per > KWS_SMALL_WGRAD_ROWS || ((int64_t)a->K1 * a->N1) % 4 != 0 ||
      ((int64_t)a->K2 * a->N2) % 4 != 0)
    return 1;
  TailPost p;
  p.a = *a; p.rows_per = rows_per; p.S = S; p.gx2 = ceil_div(a->K2, 256); p.gx1 = ceil_div(a->K1, 256);
  const int blocks = (p.gx2 + p.gx1) * S + 1 + (a->bias1 ? 1 : 0);
  KwsProfScope prof("small_wgrad", 2.0 * a->B * ((double)a->K1 * a->N1 + (double)a->K2 * a->N2),
                    4.0 * ((double)a->B * (a->K1 + a->K2 + a->N1 + a->N2) + (double)S * ((double)a->K1 * a->N1 + (double)a->K2 * a->N2)), st);
  hipLaunchKernelGGL(tail_post_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  KWS_LAUNCH_CHECK("tail_post_kernel");
  *S_out = S;
  return KWS_OK;
}

int kws_metrics_launch(const float* per_loss, const float* per_correct, int B, float* metrics, hipStream_t st) {
  hipLaunchKernelGGL(metrics_kernel, dim3(1), dim3(256), 0, st, per_loss, per_correct, B, metrics);
  KWS_LAUNCH_CHECK("metrics_kernel");
  return KWS_OK;
}
