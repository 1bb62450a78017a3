// Kernels specific to conv_1d_log_mfcc_model (reference model.py:1400-1479):
// the residual-block join (BN + ReLU6 + MaxPool1D(pool=stride) + Add, model.py:1429-1441) forward and
// backward, and the softmax-over-time attention tail (model.py:1464-1473) with its loss
// (keras categorical_crossentropy, model.py:1477).  All HBM/latency-bound; same conventions as
// dwconv.hip: a thread owns a float4 of channels and walks a short run of time steps, cross-workgroup
// sums are written as partial slabs and folded in a fixed order by the bn.hip finalisers.
#include "net_internal.h"

namespace {

constexpr int TT = 8;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 bn4(float4 v, float4 sc, float4 sh) {
  return make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
}
__device__ __forceinline__ float4 relu6_4(float4 v) {
  return make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w));
}
__device__ __forceinline__ float mk(float pre) { return (pre > 0.f && pre <= 6.f) ? 1.f : 0.f; }

// o[b,t,c] = max_{j<P} relu6(bn(y[b,P t+j,c])) + res
template <int P, bool RES_BN>
__global__ __launch_bounds__(256) void block_out_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bn,
                                                            const float* __restrict__ res,
                                                            const float* __restrict__ res_bn, float* __restrict__ o,
                                                            int64_t n4, int L, int Lo, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int C4 = C >> 2;
  const int c = (int)(i % C4) * 4;
  const int64_t bt = i / C4;  // b*Lo + t
  const int64_t b = bt / Lo;
  const int t = (int)(bt - b * Lo);
  const int64_t u0 = b * L + (int64_t)t * P;   // MaxPool1D(P, P, 'same'): window t = inputs P t .. P t + P - 1 that exist
  const float4 sc = ld4(bn + c), sh = ld4(bn + C + c);
  float4 v = relu6_4(bn4(ld4(y + u0 * C + c), sc, sh));
  if (P == 2 && 2 * t + 1 < L) {               // an odd L leaves the last window with one element (Lo = ceil(L / 2))
    const float4 w = relu6_4(bn4(ld4(y + (u0 + 1) * C + c), sc, sh));
    v = make_float4(fmaxf(v.x, w.x), fmaxf(v.y, w.y), fmaxf(v.z, w.z), fmaxf(v.w, w.w));
  }
  float4 r = ld4(res + bt * C + c);
  if (RES_BN) r = bn4(r, ld4(res_bn + c), ld4(res_bn + C + c));
  *reinterpret_cast<float4*>(o + bt * C + c) = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
}

// Round 6: the join of block i AND the first depthwise convolution of block i + 1 (k = 3, stride 1, 'same': every log-mfcc block starts
// with one) in ONE pass.  A thread computes o = maxpool_P(relu6(bn(y2))) + res for the TT steps of its unit plus one halo step on each
// side, writes its own TT once, and writes z[t] = w0 o[t-1] + w1 o[t] + w2 o[t+1] from registers - the two launches wrote o and read it
// again (one tensor pass of eleven per block, and a launch).  Every value is computed by the expressions of block_out_fwd_kernel and
// dwconv_fwd_kernel<1, false> in their order: o and z are bit-identical to the two launches' (gemm mode 1 keeps those).
// HAS_RES = false: no residual (o = relu6(bn(y)), bn_relu6_apply_kernel's expression): the first convolution's activation and the first
// block's depthwise convolution in one pass.
template <int P, bool RES_BN, bool HAS_RES = true>
__global__ __launch_bounds__(256) void block_out_dw_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bn,
                                                               const float* __restrict__ res, const float* __restrict__ res_bn,
                                                               const float* __restrict__ w, float* __restrict__ o,
                                                               float* __restrict__ z, int B, int L, int Lo, int C, int nchunks) {
  constexpr int NK = TT + 2;
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * nchunks * C4;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int c = (int)(id % C4) * 4;
    const int64_t rest = id / C4;
    const int chunk = (int)(rest % nchunks);
    const int64_t b = rest / nchunks;
    const float4 sc = ld4(bn + c), sh = ld4(bn + C + c);
    float4 rsc = sc, rsh = sh;
    if (RES_BN) {
      rsc = ld4(res_bn + c);
      rsh = ld4(res_bn + C + c);
    }
    const float4 w0 = ld4(w + c), w1 = ld4(w + C + c), w2 = ld4(w + 2 * C + c);
    const int t0 = chunk * TT;
    const float* yb = y + b * (int64_t)L * C + c;
    const float* rb = res + b * (int64_t)Lo * C + c;
    float4 a0[NK], a1[P == 2 ? NK : 1], rv[HAS_RES ? NK : 1];
    // all loads of the unit first, from clamped rows (a step outside [0, Lo) becomes a zero below: the convolution's 'same' padding)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int t = t0 - 1 + k;
      const int tc = t < 0 ? 0 : (t < Lo ? t : Lo - 1);
      a0[k] = ld4(yb + (int64_t)tc * P * C);
      if (P == 2) {
        const int u1 = 2 * tc + 1 < L ? 2 * tc + 1 : L - 1;
        a1[k] = ld4(yb + (int64_t)u1 * C);
      }
      if (HAS_RES) rv[k] = ld4(rb + (int64_t)tc * C);
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int t = t0 - 1 + k;
      const int tc = t < 0 ? 0 : (t < Lo ? t : Lo - 1);
      float4 v = relu6_4(bn4(a0[k], sc, sh));
      if (P == 2 && 2 * tc + 1 < L) {               // an odd L leaves the last window with one element
        const float4 u = relu6_4(bn4(a1[k], sc, sh));
        v = make_float4(fmaxf(v.x, u.x), fmaxf(v.y, u.y), fmaxf(v.z, u.z), fmaxf(v.w, u.w));
      }
      float4 ov = v;
      if (HAS_RES) {
        float4 r = rv[k];
        if (RES_BN) r = bn4(r, rsc, rsh);
        ov = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
      }
      a0[k] = (t >= 0 && t < Lo) ? ov : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* ob = o + b * (int64_t)Lo * C + c;
    float* zb = z + b * (int64_t)Lo * C + c;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int t = t0 + i;
      if (t >= Lo) continue;
      *reinterpret_cast<float4*>(ob + (int64_t)t * C) = a0[i + 1];
      float4 q = make_float4(w0.x * a0[i].x, w0.y * a0[i].y, w0.z * a0[i].z, w0.w * a0[i].w);
      q = bn4(a0[i + 1], w1, q);                    // fmaf(w1, o[t], q), the depthwise kernel's order
      q = bn4(a0[i + 2], w2, q);
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f qv = {q.x, q.y, q.z, q.w};          // streamed once: a non-temporal 16-byte store, as dwconv_fwd_kernel's
      __builtin_nontemporal_store(qv, reinterpret_cast<v4f*>(zb + (int64_t)t * C));
    }
  }
}

// Backward of the join's main branch.  One thread: float4 of channels x TT output steps.
// part[block][5][C]: sums of (g, g*xhat, 0, 0, 0).
template <int P, bool RELU>
__global__ __launch_bounds__(256) void block_out_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ y,
                                                            const float* __restrict__ bn, float* __restrict__ g,
                                                            float* __restrict__ part, int B, int L, int Lo, int C,
                                                            int nchunks, int R, int Cb) {
  // blockIdx.y selects a slice of Cb <= 1024 channels (more than 1024 channels: C / Cb slices)
  __shared__ float red[2][256 * 4];
  const int C4 = Cb >> 2;
  const int tid = threadIdx.x;
  const int r = tid / C4, c4 = tid - r * C4;
  const int c = blockIdx.y * Cb + c4 * 4;
  const int64_t unit = (int64_t)blockIdx.x * R + r;
  const int64_t b = unit / nchunks;
  const int chunk = (int)(unit - b * nchunks);
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
  if (b < B) {
    const float4 sc = ld4(bn + c), sh = ld4(bn + C + c), mean = ld4(bn + 2 * C + c), rstd = ld4(bn + 3 * C + c);
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int t = chunk * TT + i;
      if (t >= Lo) break;
      const float4 d = ld4(dO + (b * Lo + t) * (int64_t)C + c);
      const int64_t u0 = b * L + (int64_t)t * P;
      const bool two = P == 2 && 2 * t + 1 < L;     // an odd L: the last window has one element
      const float4 y0 = ld4(y + u0 * C + c);
      const float4 p0 = bn4(y0, sc, sh);
      float4 g0, g1 = make_float4(0.f, 0.f, 0.f, 0.f), y1 = g1;
      if (P == 1) {
        g0 = RELU ? make_float4(d.x * mk(p0.x), d.y * mk(p0.y), d.z * mk(p0.z), d.w * mk(p0.w)) : d;
      } else if (!two) {
        g0 = make_float4(d.x * mk(p0.x), d.y * mk(p0.y), d.z * mk(p0.z), d.w * mk(p0.w));
      } else {
        y1 = ld4(y + (u0 + 1) * C + c);
        const float4 p1 = bn4(y1, sc, sh);
        // the FIRST maximum of the pool window receives the gradient (MaxPoolGrad)
        const bool w1x = relu6f(p1.x) > relu6f(p0.x), w1y = relu6f(p1.y) > relu6f(p0.y);
        const bool w1z = relu6f(p1.z) > relu6f(p0.z), w1w = relu6f(p1.w) > relu6f(p0.w);
        g0 = make_float4(w1x ? 0.f : d.x * mk(p0.x), w1y ? 0.f : d.y * mk(p0.y), w1z ? 0.f : d.z * mk(p0.z),
                         w1w ? 0.f : d.w * mk(p0.w));
        g1 = make_float4(w1x ? d.x * mk(p1.x) : 0.f, w1y ? d.y * mk(p1.y) : 0.f, w1z ? d.z * mk(p1.z) : 0.f,
                         w1w ? d.w * mk(p1.w) : 0.f);
      }
      *reinterpret_cast<float4*>(g + u0 * C + c) = g0;
      sg.x += g0.x; sg.y += g0.y; sg.z += g0.z; sg.w += g0.w;
      sgx.x = fmaf(g0.x, (y0.x - mean.x) * rstd.x, sgx.x);
      sgx.y = fmaf(g0.y, (y0.y - mean.y) * rstd.y, sgx.y);
      sgx.z = fmaf(g0.z, (y0.z - mean.z) * rstd.z, sgx.z);
      sgx.w = fmaf(g0.w, (y0.w - mean.w) * rstd.w, sgx.w);
      if (two) {
        *reinterpret_cast<float4*>(g + (u0 + 1) * C + c) = g1;
        sg.x += g1.x; sg.y += g1.y; sg.z += g1.z; sg.w += g1.w;
        sgx.x = fmaf(g1.x, (y1.x - mean.x) * rstd.x, sgx.x);
        sgx.y = fmaf(g1.y, (y1.y - mean.y) * rstd.y, sgx.y);
        sgx.z = fmaf(g1.z, (y1.z - mean.z) * rstd.z, sgx.z);
        sgx.w = fmaf(g1.w, (y1.w - mean.w) * rstd.w, sgx.w);
      }
    }
  }
  *reinterpret_cast<float4*>(&red[0][tid * 4]) = sg;
  *reinterpret_cast<float4*>(&red[1][tid * 4]) = sgx;
  __syncthreads();
  for (int o = tid; o < 5 * Cb; o += blockDim.x) {
    const int q = o / Cb, ch = o - q * Cb;
    float s = 0.f;
    if (q < 2)
      for (int rr = 0; rr < R; ++rr) s += red[q][rr * Cb + ch];
    part[((int64_t)blockIdx.x * 5 + q) * C + blockIdx.y * Cb + ch] = s;
  }
}

// Two-pass form of the join backward + BatchNorm backward (round 4; what kws_dwconv_bwd_bn_f32 is for the depthwise blocks):
//   PASS 1  the reductions only - sums of g and g * xhat over (batch, time), g = the masked / max-pool-routed gradient, which is
//           NOT stored; a grid-stride walk by at most 256 workgroups, so the partial rows go straight to kws_dw_bwd_finalize
//           (the one-pass kernel above leaves one row per 256-thread workgroup: 1,536 rows for the first block of config C3 and a
//           slice_reduce launch in front of every finalise);
//   PASS 2  g recomputed from dO and y and dy = gamma rstd (g - c1 - xhat c2) written directly (kws_bn_bwd_apply's expression,
//           the same operation order: bit-identical to "store g, then kws_bn_bwd_apply").
// 5 tensor passes per join instead of 6, and no launch between the finalise and the next GEMM but this one.  All loads of a
// unit (TT steps x {dO, y0, y1}) are issued before the first is used - from clamped addresses - as in dwconv.hip.
#ifndef KWS_JOIN1_THREADS
#define KWS_JOIN1_THREADS 512          // threads of a PASS 1 workgroup (one workgroup per CU: 512 = 2 waves per SIMD)
#endif
#ifndef KWS_JOIN1_TT
#define KWS_JOIN1_TT 4                 // output steps per thread and unit in PASS 1 (all loads of a unit are issued before the first is used; measured 16: 21.1, 8: 18.5, 4: 17.1 us on the 64-channel joins of C3 - profiles/r06_c3_joins_vs_copy_rate.txt)
#endif
#ifndef KWS_JOIN2_TT
#define KWS_JOIN2_TT 4                 // the same for PASS 2 (which also writes dy): 8 -> 4 = 21.3 -> 20.1 us on the same joins
#endif
template <int P, bool RELU, int PASS, int NT, int TTJ>
// (dO and out carry no __restrict__: the shortcut BatchNorm's pass 2 runs in place, out == dO with P == 1 - every thread loads its
// own rows of a unit before it stores them, which is only defined behaviour while the compiler may not assume the two apart)
__global__ __launch_bounds__(NT) void block_join_bwd_kernel(const float* dO, const float* __restrict__ y,
                                                             const float* __restrict__ bn, const float* __restrict__ gamma,
                                                             const float* __restrict__ coef, float* out,
                                                             float* __restrict__ part, int L, int Lo, int C, int nchunks, int R,
                                                             int Cb, int64_t units) {
  __shared__ float red[2][NT * 4];
  const int C4 = Cb >> 2;
  const int tid = threadIdx.x;
  const int r = tid / C4, c4 = tid - r * C4;
  const int c = blockIdx.y * Cb + c4 * 4;
  const float4 sc = ld4(bn + c), sh = ld4(bn + C + c), mean = ld4(bn + 2 * C + c), rstd = ld4(bn + 3 * C + c);
  float4 ga = sc, c1 = sc, c2 = sc;
  if (PASS == 2) {
    ga = ld4(gamma + c);
    c1 = ld4(coef + c);
    c2 = ld4(coef + C + c);
  }
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
  for (int64_t unit = (int64_t)blockIdx.x * R + r; unit < units; unit += (int64_t)gridDim.x * R) {
    const int64_t b = unit / nchunks;
    const int chunk = (int)(unit - b * nchunks);
    float4 d[TTJ], y0[TTJ], y1[P == 2 ? TTJ : 1];
#pragma unroll
    for (int i = 0; i < TTJ; ++i) {
      const int t = chunk * TTJ + i;
      const int tc = t < Lo ? t : Lo - 1;
      d[i] = ld4(dO + (b * Lo + tc) * (int64_t)C + c);
      y0[i] = ld4(y + (b * L + (int64_t)tc * P) * C + c);
      if (P == 2) {
        const int u1 = 2 * tc + 1 < L ? 2 * tc + 1 : L - 1;
        y1[i] = ld4(y + (b * L + u1) * (int64_t)C + c);
      }
    }
#pragma unroll
    for (int i = 0; i < TTJ; ++i) {
      const int t = chunk * TTJ + i;
      if (t >= Lo) continue;
      const int64_t u0 = b * L + (int64_t)t * P;
      const bool two = P == 2 && 2 * t + 1 < L;     // an odd L: the last window has one element
      const float4 dd = d[i], a0 = y0[i];
      const float4 p0 = bn4(a0, sc, sh);
      float4 g0, g1 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = g1;
      if (P == 1) {
        g0 = RELU ? make_float4(dd.x * mk(p0.x), dd.y * mk(p0.y), dd.z * mk(p0.z), dd.w * mk(p0.w)) : dd;
      } else if (!two) {
        g0 = make_float4(dd.x * mk(p0.x), dd.y * mk(p0.y), dd.z * mk(p0.z), dd.w * mk(p0.w));
      } else {
        a1 = y1[i];
        const float4 p1 = bn4(a1, sc, sh);
        const bool w1x = relu6f(p1.x) > relu6f(p0.x), w1y = relu6f(p1.y) > relu6f(p0.y);
        const bool w1z = relu6f(p1.z) > relu6f(p0.z), w1w = relu6f(p1.w) > relu6f(p0.w);
        g0 = make_float4(w1x ? 0.f : dd.x * mk(p0.x), w1y ? 0.f : dd.y * mk(p0.y), w1z ? 0.f : dd.z * mk(p0.z),
                         w1w ? 0.f : dd.w * mk(p0.w));
        g1 = make_float4(w1x ? dd.x * mk(p1.x) : 0.f, w1y ? dd.y * mk(p1.y) : 0.f, w1z ? dd.z * mk(p1.z) : 0.f,
                         w1w ? dd.w * mk(p1.w) : 0.f);
      }
      if (PASS == 1) {
        sg.x += g0.x; sg.y += g0.y; sg.z += g0.z; sg.w += g0.w;
        sgx.x = fmaf(g0.x, (a0.x - mean.x) * rstd.x, sgx.x);
        sgx.y = fmaf(g0.y, (a0.y - mean.y) * rstd.y, sgx.y);
        sgx.z = fmaf(g0.z, (a0.z - mean.z) * rstd.z, sgx.z);
        sgx.w = fmaf(g0.w, (a0.w - mean.w) * rstd.w, sgx.w);
        if (two) {
          sg.x += g1.x; sg.y += g1.y; sg.z += g1.z; sg.w += g1.w;
          sgx.x = fmaf(g1.x, (a1.x - mean.x) * rstd.x, sgx.x);
          sgx.y = fmaf(g1.y, (a1.y - mean.y) * rstd.y, sgx.y);
          sgx.z = fmaf(g1.z, (a1.z - mean.z) * rstd.z, sgx.z);
          sgx.w = fmaf(g1.w, (a1.w - mean.w) * rstd.w, sgx.w);
        }
      } else {
        float4 o;
        o.x = ga.x * rstd.x * (g0.x - c1.x - (a0.x - mean.x) * rstd.x * c2.x);
        o.y = ga.y * rstd.y * (g0.y - c1.y - (a0.y - mean.y) * rstd.y * c2.y);
        o.z = ga.z * rstd.z * (g0.z - c1.z - (a0.z - mean.z) * rstd.z * c2.z);
        o.w = ga.w * rstd.w * (g0.w - c1.w - (a0.w - mean.w) * rstd.w * c2.w);
        *reinterpret_cast<float4*>(out + u0 * C + c) = o;
        if (two) {
          o.x = ga.x * rstd.x * (g1.x - c1.x - (a1.x - mean.x) * rstd.x * c2.x);
          o.y = ga.y * rstd.y * (g1.y - c1.y - (a1.y - mean.y) * rstd.y * c2.y);
          o.z = ga.z * rstd.z * (g1.z - c1.z - (a1.z - mean.z) * rstd.z * c2.z);
          o.w = ga.w * rstd.w * (g1.w - c1.w - (a1.w - mean.w) * rstd.w * c2.w);
          *reinterpret_cast<float4*>(out + (u0 + 1) * C + c) = o;
        }
      }
    }
  }
  if (PASS == 1) {
    *reinterpret_cast<float4*>(&red[0][tid * 4]) = sg;
    *reinterpret_cast<float4*>(&red[1][tid * 4]) = sgx;
    __syncthreads();
    for (int o = tid; o < 5 * Cb; o += blockDim.x) {
      const int q = o / Cb, ch = o - q * Cb;
      float s = 0.f;
      if (q < 2)
        for (int rr = 0; rr < R; ++rr) s += red[q][rr * Cb + ch];
      part[((int64_t)blockIdx.x * 5 + q) * C + blockIdx.y * Cb + ch] = s;
    }
  }
}

// ---- 3-wide SAME max-pool join (conv_1d_residual, model.py:874-875): window of output t = inputs
// t*S - PL .. t*S - PL + 2 (positions outside [0, L) are -inf), the FIRST maximum wins --------------------------
__device__ __forceinline__ float4 act_or_ninf(const float* y, int64_t row0, int u, int L, int C, int c, float4 sc, float4 sh) {
  if (u < 0 || u >= L) return make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  return relu6_4(bn4(ld4(y + (row0 + u) * C + c), sc, sh));
}

template <bool RES_BN>
__global__ __launch_bounds__(256) void block_out3_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bn,
                                                             const float* __restrict__ res,
                                                             const float* __restrict__ res_bn, float* __restrict__ o,
                                                             int64_t n4, int L, int Lo, int C, int S, int PL) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int C4 = C >> 2;
  const int c = (int)(i % C4) * 4;
  const int64_t bt = i / C4;  // b*Lo + t
  const int64_t b = bt / Lo;
  const int t = (int)(bt - b * Lo);
  const float4 sc = ld4(bn + c), sh = ld4(bn + C + c);
  const int u0 = t * S - PL;
  const float4 a0 = act_or_ninf(y, b * L, u0, L, C, c, sc, sh), a1 = act_or_ninf(y, b * L, u0 + 1, L, C, c, sc, sh),
               a2 = act_or_ninf(y, b * L, u0 + 2, L, C, c, sc, sh);
  const float4 v = make_float4(fmaxf(fmaxf(a0.x, a1.x), a2.x), fmaxf(fmaxf(a0.y, a1.y), a2.y),
                               fmaxf(fmaxf(a0.z, a1.z), a2.z), fmaxf(fmaxf(a0.w, a1.w), a2.w));
  float4 r = ld4(res + bt * C + c);
  if (RES_BN) r = bn4(r, ld4(res_bn + c), ld4(res_bn + C + c));
  *reinterpret_cast<float4*>(o + bt * C + c) = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
}

// offset (0..2) of the first maximum of a window
__device__ __forceinline__ int first_max3(float a0, float a1, float a2) {
  int j = 0;
  float m = a0;
  if (a1 > m) { m = a1; j = 1; }
  if (a2 > m) j = 2;
  return j;
}

// Backward of the 3-wide join's main branch: one thread = float4 of channels x TT INPUT positions; every input
// position collects dO of the (at most 3) windows it won, masked by its own ReLU6 gate.
// part[block][5][C]: sums of (g, g*xhat, 0, 0, 0).
__global__ __launch_bounds__(256) void block_out3_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ y,
                                                             const float* __restrict__ bn, float* __restrict__ g,
                                                             float* __restrict__ part, int B, int L, int Lo, int C,
                                                             int S, int PL, int nchunks, int R, int Cb) {
  __shared__ float red[2][256 * 4];
  const int C4 = Cb >> 2;
  const int tid = threadIdx.x;
  const int r = tid / C4, c4 = tid - r * C4;
  const int c = blockIdx.y * Cb + c4 * 4;
  const int64_t unit = (int64_t)blockIdx.x * R + r;
  const int64_t b = unit / nchunks;
  const int chunk = (int)(unit - b * nchunks);
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
  if (b < B) {
    const float4 sc = ld4(bn + c), sh = ld4(bn + C + c), mean = ld4(bn + 2 * C + c), rstd = ld4(bn + 3 * C + c);
    for (int i = 0; i < TT; ++i) {
      const int u = chunk * TT + i;
      if (u >= L) break;
      const float4 yu = ld4(y + (b * L + u) * (int64_t)C + c);
      const float4 pu = bn4(yu, sc, sh);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 3; ++j) {          // u is element j of window t = (u + PL - j) / S
        const int num = u + PL - j;
        if (num < 0 || num % S != 0) continue;
        const int t = num / S;
        if (t >= Lo) continue;
        const int u0 = t * S - PL;
        const float4 a0 = act_or_ninf(y, b * L, u0, L, C, c, sc, sh), a1 = act_or_ninf(y, b * L, u0 + 1, L, C, c, sc, sh),
                     a2 = act_or_ninf(y, b * L, u0 + 2, L, C, c, sc, sh);
        const float4 d = ld4(dO + (b * Lo + t) * (int64_t)C + c);
        if (first_max3(a0.x, a1.x, a2.x) == j) acc.x += d.x;
        if (first_max3(a0.y, a1.y, a2.y) == j) acc.y += d.y;
        if (first_max3(a0.z, a1.z, a2.z) == j) acc.z += d.z;
        if (first_max3(a0.w, a1.w, a2.w) == j) acc.w += d.w;
      }
      const float4 g0 = make_float4(acc.x * mk(pu.x), acc.y * mk(pu.y), acc.z * mk(pu.z), acc.w * mk(pu.w));
      *reinterpret_cast<float4*>(g + (b * L + u) * (int64_t)C + c) = g0;
      sg.x += g0.x; sg.y += g0.y; sg.z += g0.z; sg.w += g0.w;
      sgx.x = fmaf(g0.x, (yu.x - mean.x) * rstd.x, sgx.x);
      sgx.y = fmaf(g0.y, (yu.y - mean.y) * rstd.y, sgx.y);
      sgx.z = fmaf(g0.z, (yu.z - mean.z) * rstd.z, sgx.z);
      sgx.w = fmaf(g0.w, (yu.w - mean.w) * rstd.w, sgx.w);
    }
  }
  *reinterpret_cast<float4*>(&red[0][tid * 4]) = sg;
  *reinterpret_cast<float4*>(&red[1][tid * 4]) = sgx;
  __syncthreads();
  for (int o = tid; o < 5 * Cb; o += blockDim.x) {
    const int q = o / Cb, ch = o - q * Cb;
    float s = 0.f;
    if (q < 2)
      for (int rr = 0; rr < R; ++rr) s += red[q][rr * Cb + ch];
    part[((int64_t)blockIdx.x * 5 + q) * C + blockIdx.y * Cb + ch] = s;
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

__global__ __launch_bounds__(256) void add_strided_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                          int64_t n4, int L_out, int L_in, int C, int stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int C4 = C >> 2;
  const int c = (int)(i % C4) * 4;
  const int64_t bt = i / C4;
  const int64_t b = bt / L_in;
  const int t = (int)(bt - b * L_in);
  float* o = out + ((b * L_out + (int64_t)t * stride) * C + c);
  const float4 x = *reinterpret_cast<const float4*>(o), y = ld4(in + bt * C + c);
  *reinterpret_cast<float4*>(o) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// ------------------------------------------------------------------------------------------------------
// attention tail
// ------------------------------------------------------------------------------------------------------
constexpr int LM_MAXT = 16;
constexpr int LM_MAXNC = 64;

struct LmArgs {
  kws_lm_tail_args a;
  uint32_t key, thresh;
  float inv_keep, inv_loss_batch;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// u[b,t] = sum_c Wa[c] * sum_j wa[j,c] x[b, t+j-1, c]      (dw k3 SAME, then pointwise C -> 1)
__global__ __launch_bounds__(256) void lm_att_logits_kernel(LmArgs p) {
  __shared__ float red[4][LM_MAXT];
  const kws_lm_tail_args& a = p.a;
  const int T = a.T, C = a.C, b = blockIdx.x, tid = threadIdx.x;
  const float* xb = a.x + (int64_t)b * T * C;
  float acc[LM_MAXT];
#pragma unroll
  for (int t = 0; t < LM_MAXT; ++t) acc[t] = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float w0 = a.wa[c], w1 = a.wa[C + c], w2 = a.wa[2 * C + c], wp = a.Wa[c];
    float xv[LM_MAXT + 1];                            // the column's T values in flight at once, from clamped addresses (a load behind
#pragma unroll                                        // "t + 1 < T ? .. : 0" is a branch and a wait per step)
    for (int t = 0; t < LM_MAXT; ++t) xv[t] = xb[(t < T ? t : 0) * C + c];
    xv[LM_MAXT] = 0.f;
    float xm = 0.f, x0 = xv[0];
#pragma unroll
    for (int t = 0; t < LM_MAXT; ++t) {
      if (t < T) {
        const float xp = (t + 1 < T) ? xv[t + 1] : 0.f;
        acc[t] = fmaf(wp, fmaf(w2, xp, fmaf(w1, x0, w0 * xm)), acc[t]);
        xm = x0;
        x0 = xp;
      }
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < LM_MAXT; ++t) {
    if (t < T) {
      const float s = wave_sum(acc[t]);
      if (lane == 0) red[wave][t] = s;
    }
  }
  __syncthreads();
  if (tid < T) a.u[(int64_t)b * T + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// BatchNorm statistics of the 1-channel attention logits over all B*T values (fixed-order, one workgroup).
// Round 6: 16 waves and four independent loads in flight per thread - the 256-thread loop of one dependent load per iteration took
// 26 us for the 24,576 values of config C3 (96 round trips), 31 us in the backward twin below: 1 % of that step between them.
constexpr int LM_BN_NT = 1024;
__global__ __launch_bounds__(LM_BN_NT) void lm_att_bn_kernel(LmArgs p, int training) {
  __shared__ double red[2][LM_BN_NT];
  const kws_lm_tail_args& a = p.a;
  const int n = a.B * a.T, tid = threadIdx.x;
  float gamma = a.bn_gamma[0], beta = a.bn_beta[0];
  if (!training) {
    if (tid == 0) {
      const float rstd = 1.0f / sqrtf(a.mv[0] + KWS_BN_EPS);
      a.bn[0] = gamma * rstd; a.bn[1] = beta - a.mm[0] * gamma * rstd; a.bn[2] = a.mm[0]; a.bn[3] = rstd;
    }
    return;
  }
  double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
  int i = tid;
  for (; i + 3 * LM_BN_NT < n; i += 4 * LM_BN_NT) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = a.u[i + k * LM_BN_NT];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] += (double)v[k];
      ss[k] += (double)v[k] * (double)v[k];
    }
  }
  for (; i < n; i += LM_BN_NT) {
    const double v = a.u[i];
    s[0] += v;
    ss[0] += v * v;
  }
  red[0][tid] = (s[0] + s[1]) + (s[2] + s[3]);
  red[1][tid] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
  __syncthreads();
  for (int o = LM_BN_NT / 2; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] += red[0][tid + o];
      red[1][tid] += red[1][tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double mean = red[0][0] / n;
    double var = red[1][0] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)KWS_BN_EPS));
    const float meanf = (float)mean, varf = (float)var;
    a.bn[0] = gamma * rstd; a.bn[1] = beta - meanf * gamma * rstd; a.bn[2] = meanf; a.bn[3] = rstd;
    const float omm = (float)(1.0 - (double)KWS_BN_MOMENTUM);
    a.mm[0] = a.mm[0] - (a.mm[0] - meanf) * omm;
    a.mv[0] = a.mv[0] - (a.mv[0] - varf) * omm;
  }
}

// per clip: att = softmax_t(relu6(bn(u))), feat = mean_t(x * att), dropout, dense + softmax, CCE loss and the
// backward down to (dX partial, masked gradient of the attention logits)
// (256, 8): eight waves per SIMD = 64 registers - with 69 the 2,048 workgroups of config C3's batch needed a second, quarter-full round
template <bool TRAIN>
__global__ __launch_bounds__(256, 8) void lm_tail_kernel(LmArgs p) {
  __shared__ float s_att[LM_MAXT], s_pre[LM_MAXT], s_datt[LM_MAXT], s_p[LM_MAXNC], s_dl[LM_MAXNC];
  __shared__ float s_feat[1024], s_red[4][LM_MAXNC];
  const kws_lm_tail_args& a = p.a;
  const int T = a.T, C = a.C, NC = a.NC, b = blockIdx.x, tid = threadIdx.x;
  const float* xb = a.x + (int64_t)b * T * C;
  const uint32_t row = (uint32_t)(a.row_offset + b);
  // Round 6: the clip's logits and labels are fetched by T + NC threads at once; the serial sections below (thread 0: two softmaxes and
  // the loss, kept serial so that every sum keeps its order) read them from LDS.  They used to load them one by one from global memory
  // - 12 + 32 dependent round trips per clip on a chip running 2,048 such workgroups at once: 87 us for the batch, now see profiles/.
  __shared__ float s_u[LM_MAXT], s_y[LM_MAXNC];
  if (tid < T) s_u[tid] = a.u[(int64_t)b * T + tid];
  if (TRAIN && tid >= 64 && tid < 64 + NC) s_y[tid - 64] = a.labels[(int64_t)b * NC + tid - 64];
  __syncthreads();
  if (tid == 0) {
    float m = -INFINITY;
    const float bn0 = a.bn[0], bn1 = a.bn[1];
    for (int t = 0; t < T; ++t) {
      const float pre = fmaf(s_u[t], bn0, bn1);
      s_pre[t] = pre;
      s_att[t] = relu6f(pre);
      m = fmaxf(m, s_att[t]);
    }
    float den = 0.f;
    for (int t = 0; t < T; ++t) {
      s_att[t] = expf(s_att[t] - m);
      den += s_att[t];
    }
    for (int t = 0; t < T; ++t) s_att[t] /= den;
  }
  __syncthreads();
  if (TRAIN && a.att != nullptr && tid < T) a.att[(int64_t)b * T + tid] = s_att[tid];
  for (int c = tid; c < C; c += 256) {
    float f = 0.f;
    float xv[LM_MAXT];                                // all T loads of the column in flight at once (T is a run-time bound: the plain
#pragma unroll                                        // loop waited for one load per iteration)
    for (int t = 0; t < LM_MAXT; ++t) xv[t] = xb[(t < T ? t : 0) * C + c];
#pragma unroll
    for (int t = 0; t < LM_MAXT; ++t)
      if (t < T) f = fmaf(xv[t], s_att[t], f);
    f = f / (float)T;
    if (TRAIN) {
      f = kws_keep(row * (uint32_t)C + (uint32_t)c, p.key, p.thresh) ? f * p.inv_keep : 0.f;
      a.fd[(int64_t)b * C + c] = f;
    }
    s_feat[c] = f;
  }
  __syncthreads();
  {
    const int k = tid & 63, sl = tid >> 6;
    float s = 0.f;
    if (k < NC) {
      // eight weight loads in flight per thread, consumed in the loop's own order (round 6: one dependent load per iteration before)
      for (int c0 = sl; c0 < C; c0 += 32) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + 4 * j;
          w[j] = a.Wd[(int64_t)(c < C ? c : sl) * NC + k];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + 4 * j;
          if (c < C) s = fmaf(s_feat[c], w[j], s);
        }
      }
    }
    s_red[sl][k] = s;
    __syncthreads();
    if (tid < NC) s_p[tid] = (((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid]) + a.bd[tid];
    __syncthreads();
    if (tid == 0) {
      float m = s_p[0];
      for (int q = 1; q < NC; ++q) m = fmaxf(m, s_p[q]);
      float den = 0.f;
      for (int q = 0; q < NC; ++q) {
        s_p[q] = expf(s_p[q] - m);
        den += s_p[q];
      }
      for (int q = 0; q < NC; ++q) s_p[q] /= den;
    }
    __syncthreads();
    if (tid < NC) a.probs[(int64_t)b * NC + tid] = s_p[tid];
  }
  if (!TRAIN) return;
  if (tid == 0) {
    // keras categorical_crossentropy: p /= sum(p); clip(eps, 1-eps); -sum(y log p)
    const float eps = 1e-7f;
    const float* yl = s_y;
    float S = 0.f;
    for (int q = 0; q < NC; ++q) S += s_p[q];
    float loss = 0.f, dotp = 0.f;
    int am_p = 0, am_y = 0;
    for (int q = 0; q < NC; ++q) {
      const float pn = s_p[q] / S;
      const float pc = fminf(fmaxf(pn, eps), 1.f - eps);
      loss -= yl[q] * logf(pc);
      const float inside = (pn >= eps && pn <= 1.f - eps) ? 1.f : 0.f;
      const float dpn = (-yl[q] / pc) * inside * p.inv_loss_batch;
      s_dl[q] = dpn;   // dL/dpn for now
      dotp += dpn * s_p[q];
      if (s_p[q] > s_p[am_p]) am_p = q;
      if (yl[q] > yl[am_y]) am_y = q;
    }
    a.per_loss[b] = loss;
    a.per_correct[b] = (am_p == am_y) ? 1.f : 0.f;
    float dot2 = 0.f;
    for (int q = 0; q < NC; ++q) {
      const float dp = s_dl[q] / S - dotp / (S * S);   // through the renormalisation
      s_dl[q] = dp;
      dot2 += dp * s_p[q];
    }
    for (int q = 0; q < NC; ++q) s_dl[q] = s_p[q] * (s_dl[q] - dot2);   // softmax backward
  }
  __syncthreads();
  if (tid < NC) a.dl[(int64_t)b * NC + tid] = s_dl[tid];
  // dfeat -> dX (through Multiply + GAP) and datt
  float dattl[LM_MAXT];
#pragma unroll
  for (int t = 0; t < LM_MAXT; ++t) dattl[t] = 0.f;
  float* dxb = a.dX + (int64_t)b * T * C;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int q0 = 0; q0 < NC; q0 += 8) {            // the row of Wd: eight loads in flight, the products in index order
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = a.Wd[(int64_t)c * NC + (q0 + j < NC ? q0 + j : NC - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (q0 + j < NC) s = fmaf(w[j], s_dl[q0 + j], s);
    }
    const bool keep = kws_keep(row * (uint32_t)C + (uint32_t)c, p.key, p.thresh);
    const float dprod = (keep ? s * p.inv_keep : 0.f) / (float)T;
    float xv[LM_MAXT];
#pragma unroll
    for (int t = 0; t < LM_MAXT; ++t) xv[t] = xb[(t < T ? t : 0) * C + c];
#pragma unroll
    for (int t = 0; t < LM_MAXT; ++t) {
      if (t < T) {
        dxb[t * C + c] = dprod * s_att[t];
        dattl[t] = fmaf(dprod, xv[t], dattl[t]);
      }
    }
  }
  {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < LM_MAXT; ++t) {
      if (t < T) {
        const float s = wave_sum(dattl[t]);
        if (lane == 0) s_red[wave][t] = s;
      }
    }
    __syncthreads();
    if (tid < T) s_datt[tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
    __syncthreads();
    if (tid == 0) {
      float dot = 0.f;
      for (int t = 0; t < T; ++t) dot += s_att[t] * s_datt[t];
      for (int t = 0; t < T; ++t) {
        const float du = s_att[t] * (s_datt[t] - dot);
        a.gu[(int64_t)b * T + t] = du * mk(s_pre[t]);
      }
    }
  }
}

// BN backward of the attention logits (1 channel): sums over all B*T, then coef / dgamma / dbeta
__global__ __launch_bounds__(LM_BN_NT) void lm_att_bn_bwd_kernel(LmArgs p) {
  __shared__ double red[2][LM_BN_NT];
  const kws_lm_tail_args& a = p.a;
  const int n = a.B * a.T, tid = threadIdx.x;
  const float mean = a.bn[2], rstd = a.bn[3];
  double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
  int i = tid;
  for (; i + 3 * LM_BN_NT < n; i += 4 * LM_BN_NT) {
    float g[4], u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g[k] = a.gu[i + k * LM_BN_NT];
      u[k] = a.u[i + k * LM_BN_NT];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] += (double)g[k];
      sx[k] += (double)g[k] * (double)((u[k] - mean) * rstd);
    }
  }
  for (; i < n; i += LM_BN_NT) {
    const double g = a.gu[i];
    s[0] += g;
    sx[0] += g * (double)((a.u[i] - mean) * rstd);
  }
  red[0][tid] = (s[0] + s[1]) + (s[2] + s[3]);
  red[1][tid] = (sx[0] + sx[1]) + (sx[2] + sx[3]);
  __syncthreads();
  for (int o = LM_BN_NT / 2; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] += red[0][tid + o];
      red[1][tid] += red[1][tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.d_beta[0] = (float)red[0][0];
    a.d_gamma[0] = (float)red[1][0];
    a.coef[0] = (float)(red[0][0] / n);
    a.coef[1] = (float)(red[1][0] / n);
  }
}

// per clip: dyu -> dza -> (dWa, dwa partials) and dX += dw^T dza
__global__ __launch_bounds__(256) void lm_att_bwd_kernel(LmArgs p) {
  __shared__ float s_dyu[LM_MAXT];
  const kws_lm_tail_args& a = p.a;
  const int T = a.T, C = a.C, b = blockIdx.x, tid = threadIdx.x;
  if (tid < T) {
    const float uu = a.u[(int64_t)b * T + tid];
    const float xh = (uu - a.bn[2]) * a.bn[3];
    s_dyu[tid] = a.bn_gamma[0] * a.bn[3] * (a.gu[(int64_t)b * T + tid] - a.coef[0] - xh * a.coef[1]);
  }
  __syncthreads();
  const float* xb = a.x + (int64_t)b * T * C;
  float* dxb = a.dX + (int64_t)b * T * C;
  float* pb = a.part + (int64_t)b * 5 * C;
  for (int c = tid; c < C; c += 256) {
    const float w0 = a.wa[c], w1 = a.wa[C + c], w2 = a.wa[2 * C + c], wp = a.Wa[c];
    float sWa = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < T; ++t) {
      const float xm = t > 0 ? xb[(t - 1) * C + c] : 0.f, x0 = xb[t * C + c];
      const float xp = t + 1 < T ? xb[(t + 1) * C + c] : 0.f;
      const float za = fmaf(w2, xp, fmaf(w1, x0, w0 * xm));
      const float dy = s_dyu[t];
      sWa = fmaf(za, dy, sWa);
      const float dza = dy * wp;
      s0 = fmaf(dza, xm, s0);
      s1 = fmaf(dza, x0, s1);
      s2 = fmaf(dza, xp, s2);
      // dX[t'] += sum_j wa[j] dza[t' - j + 1]
      const float dzm = t > 0 ? s_dyu[t - 1] * wp : 0.f, dzp = t + 1 < T ? s_dyu[t + 1] * wp : 0.f;
      dxb[t * C + c] += fmaf(w0, dzp, fmaf(w1, dza, w2 * dzm));
    }
    pb[c] = sWa; pb[C + c] = 0.f; pb[2 * C + c] = s0; pb[3 * C + c] = s1; pb[4 * C + c] = s2;
  }
}

struct Geom {
  int nchunks, R, block, ny, Cb;
  int64_t grid;
};
bool geom_ok(int C) { return C > 0 && C % 4 == 0 && (C / 4) % ceil_div(C / 4, 256) == 0; }
Geom geom(int B, int Lo, int C) {
  Geom g;
  g.ny = ceil_div(C / 4, 256);          // channel slices of at most 1024 channels (steffeNet: 1536 = 2 x 768)
  g.Cb = C / g.ny;
  const int C4 = g.Cb / 4;
  g.nchunks = ceil_div(Lo, TT);
  g.R = 256 / C4 < 1 ? 1 : 256 / C4;
  g.block = g.R * C4;
  g.grid = ceil_div64((int64_t)B * g.nchunks, g.R);
  return g;
}

LmArgs make_args(const kws_lm_tail_args* a) {
  LmArgs p;
  p.a = *a;
  p.key = kws_dropout_key(a->seed, a->step, 1);
  p.thresh = kws_dropout_threshold(a->keep_prob);
  p.inv_keep = (float)(1.0 / (double)a->keep_prob);
  p.inv_loss_batch = 1.0f / (float)(a->loss_batch > 0 ? a->loss_batch : 1);
  return p;
}

// ------------------------------------------------------------------------------------------------------
// global max ++ average pooling tail (steffeNet)
// ------------------------------------------------------------------------------------------------------
constexpr int GP_MAXC = 2048;
struct GpArgs {
  kws_gp_tail_args a;
  uint32_t key, thresh;
  float inv_keep, inv_loss_batch;
};

template <bool TRAIN>
__global__ __launch_bounds__(256) void gp_tail_kernel(GpArgs p) {
  __shared__ float s_feat[2 * GP_MAXC], s_red[4][LM_MAXNC], s_p[LM_MAXNC], s_dl[LM_MAXNC];
  const kws_gp_tail_args& a = p.a;
  const int T = a.T, C = a.C, NC = a.NC, b = blockIdx.x, tid = threadIdx.x;
  const bool pmax = a.pool_max != 0;
  const int F = pmax ? 2 * C : C;         // features: [max | avg] or [avg]
  const int avg0 = pmax ? C : 0;          // where the averages start
  const float* xb = a.x + (int64_t)b * T * C;
  const uint32_t row = (uint32_t)(a.row_offset + b);
  for (int c = tid; c < C; c += 256) {
    float mx = xb[c], sm = xb[c];
    for (int t = 1; t < T; ++t) {
      const float v = xb[t * C + c];
      mx = fmaxf(mx, v);
      sm += v;
    }
    float f0 = mx, f1 = sm / (float)T;
    if (TRAIN) {
      if (pmax) f0 = kws_keep(row * (uint32_t)F + (uint32_t)c, p.key, p.thresh) ? f0 * p.inv_keep : 0.f;
      f1 = kws_keep(row * (uint32_t)F + (uint32_t)(avg0 + c), p.key, p.thresh) ? f1 * p.inv_keep : 0.f;
      if (pmax) a.fd[(int64_t)b * F + c] = f0;
      a.fd[(int64_t)b * F + avg0 + c] = f1;
    }
    if (pmax) s_feat[c] = f0;
    s_feat[avg0 + c] = f1;
  }
  __syncthreads();
  {
    const int k = tid & 63, sl = tid >> 6;
    float s = 0.f;
    if (k < NC)
      for (int i = sl; i < F; i += 4) s = fmaf(s_feat[i], a.Wd[(int64_t)i * NC + k], s);
    s_red[sl][k] = s;
    __syncthreads();
    if (tid < NC)
      s_p[tid] = (((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid]) + (a.bd ? a.bd[tid] : 0.f);
    __syncthreads();
    if (tid == 0) {
      float m = s_p[0];
      for (int q = 1; q < NC; ++q) m = fmaxf(m, s_p[q]);
      float den = 0.f;
      for (int q = 0; q < NC; ++q) {
        s_p[q] = expf(s_p[q] - m);
        den += s_p[q];
      }
      for (int q = 0; q < NC; ++q) s_p[q] /= den;
    }
    __syncthreads();
    if (tid < NC) a.probs[(int64_t)b * NC + tid] = s_p[tid];
  }
  if (!TRAIN) return;
  if (tid == 0) {
    const float eps = 1e-7f;
    const float* yl = a.labels + (int64_t)b * NC;
    int am_p = 0, am_y = 0;
    float loss = 0.f;
    if (a.loss_kind == 0) {
      // softmax-CE on log(clip(p)) with label smoothing (utils.py:100-108), as in the raw-waveform net's tail
      float S = 0.f, ysum = 0.f;
      for (int q = 0; q < NC; ++q) S += fminf(fmaxf(s_p[q], eps), 1.f - eps);
      const float logS = logf(S);
      for (int q = 0; q < NC; ++q) {
        const float ysm = yl[q] * (1.f - a.label_smoothing) + a.label_smoothing / (float)NC;
        ysum += ysm;
        const float pc = fminf(fmaxf(s_p[q], eps), 1.f - eps);
        loss -= ysm * (logf(pc) - logS);
        if (s_p[q] > s_p[am_p]) am_p = q;
        if (yl[q] > yl[am_y]) am_y = q;
      }
      float dot = 0.f;
      for (int q = 0; q < NC; ++q) {
        const float ysm = yl[q] * (1.f - a.label_smoothing) + a.label_smoothing / (float)NC;
        const float pc = fminf(fmaxf(s_p[q], eps), 1.f - eps);
        const float inside = (s_p[q] >= eps && s_p[q] <= 1.f - eps) ? 1.f : 0.f;
        const float dp = (-ysm / pc + ysum / S) * p.inv_loss_batch * inside;
        s_dl[q] = dp;
        dot += dp * s_p[q];
      }
      for (int q = 0; q < NC; ++q) s_dl[q] = s_p[q] * (s_dl[q] - dot);
    } else {
      // keras categorical_crossentropy: p /= sum(p); clip(eps, 1-eps); -sum(y log p)   (as lm_tail_kernel)
      float S = 0.f;
      for (int q = 0; q < NC; ++q) S += s_p[q];
      float dotp = 0.f;
      for (int q = 0; q < NC; ++q) {
        const float pn = s_p[q] / S;
        const float pc = fminf(fmaxf(pn, eps), 1.f - eps);
        loss -= yl[q] * logf(pc);
        const float inside = (pn >= eps && pn <= 1.f - eps) ? 1.f : 0.f;
        const float dpn = (-yl[q] / pc) * inside * p.inv_loss_batch;
        s_dl[q] = dpn;
        dotp += dpn * s_p[q];
        if (s_p[q] > s_p[am_p]) am_p = q;
        if (yl[q] > yl[am_y]) am_y = q;
      }
      float dot2 = 0.f;
      for (int q = 0; q < NC; ++q) {
        const float dp = s_dl[q] / S - dotp / (S * S);
        s_dl[q] = dp;
        dot2 += dp * s_p[q];
      }
      for (int q = 0; q < NC; ++q) s_dl[q] = s_p[q] * (s_dl[q] - dot2);
    }
    a.per_loss[b] = loss;
    a.per_correct[b] = (am_p == am_y) ? 1.f : 0.f;
  }
  __syncthreads();
  if (tid < NC) a.dl[(int64_t)b * NC + tid] = s_dl[tid];
  // dfeat = (Wd . dl) * mask / keep; reduce_max shares its gradient equally among ties (_MinOrMaxGrad)
  float* dxb = a.dX + (int64_t)b * T * C;
  for (int c = tid; c < C; c += 256) {
    float d0 = 0.f, d1 = 0.f;
    for (int q = 0; q < NC; ++q) {
      if (pmax) d0 = fmaf(a.Wd[(int64_t)c * NC + q], s_dl[q], d0);
      d1 = fmaf(a.Wd[(int64_t)(avg0 + c) * NC + q], s_dl[q], d1);
    }
    if (pmax) d0 = kws_keep(row * (uint32_t)F + (uint32_t)c, p.key, p.thresh) ? d0 * p.inv_keep : 0.f;
    d1 = kws_keep(row * (uint32_t)F + (uint32_t)(avg0 + c), p.key, p.thresh) ? d1 * p.inv_keep : 0.f;
    const float avg = d1 / (float)T;
    if (pmax) {
      float mx = xb[c];
      for (int t = 1; t < T; ++t) mx = fmaxf(mx, xb[t * C + c]);
      int ties = 0;
      for (int t = 0; t < T; ++t) ties += xb[t * C + c] == mx ? 1 : 0;
      const float share = d0 / (float)ties;
      for (int t = 0; t < T; ++t) dxb[t * C + c] = (xb[t * C + c] == mx ? share : 0.f) + avg;
    } else {
      for (int t = 0; t < T; ++t) dxb[t * C + c] = avg;
    }
  }
}

}  // namespace

int kws_gp_tail_launch(const kws_gp_tail_args* a, int training, hipStream_t st) {
  KWS_REQUIRE(a && a->x && a->Wd && a->probs && a->B > 0 && a->T > 0 && a->C > 0 && a->C <= GP_MAXC && a->NC > 0 &&
                  a->NC <= LM_MAXNC,
              "gp_tail: bad arguments (T=%d C=%d NC=%d)", a ? a->T : 0, a ? a->C : 0, a ? a->NC : 0);
  KWS_REQUIRE(!training || (a->labels && a->dX && a->fd && a->dl && a->per_loss && a->per_correct),
              "gp_tail: training needs labels, dX, fd, dl, per_loss, per_correct");
  GpArgs p;
  p.a = *a;
  p.key = kws_dropout_key(a->seed, a->step, 1);
  p.thresh = kws_dropout_threshold(a->keep_prob);
  p.inv_keep = (float)(1.0 / (double)a->keep_prob);
  p.inv_loss_batch = 1.0f / (float)(a->loss_batch > 0 ? a->loss_batch : 1);
  KwsProfScope prof("gp_tail", 0.0, 4.0 * (double)a->B * a->T * a->C * (training ? 3.0 : 1.0), st);
  if (training) hipLaunchKernelGGL((gp_tail_kernel<true>), dim3((unsigned)a->B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gp_tail_kernel<false>), dim3((unsigned)a->B), dim3(256), 0, st, p);
  KWS_LAUNCH_CHECK("gp_tail_kernel");
  return KWS_OK;
}

int kws_block_out_fwd(const float* y, const float* bn, const float* res, const float* res_bn, float* o, int B, int L,
                      int C, int pool, hipStream_t st) {
  KWS_REQUIRE(y && bn && res && o && B > 0 && L > 0 && C % 4 == 0 && (pool == 1 || pool == 2),
              "block_out_fwd: bad arguments (L=%d C=%d pool=%d)", L, C, pool);
  const int Lo = (L + pool - 1) / pool;             // 'same' pooling: ceil
  const int64_t n4 = (int64_t)B * Lo * C / 4;
  KwsProfScope prof("block_out_fwd", 4.0 * B * L * C, 4.0 * ((double)B * L * C + 2.0 * B * Lo * C), st);
  dim3 g((unsigned)ceil_div64(n4, 256)), b(256);
  if (pool == 1) {
    if (res_bn) hipLaunchKernelGGL((block_out_fwd_kernel<1, true>), g, b, 0, st, y, bn, res, res_bn, o, n4, L, Lo, C);
    else hipLaunchKernelGGL((block_out_fwd_kernel<1, false>), g, b, 0, st, y, bn, res, res_bn, o, n4, L, Lo, C);
  } else {
    if (res_bn) hipLaunchKernelGGL((block_out_fwd_kernel<2, true>), g, b, 0, st, y, bn, res, res_bn, o, n4, L, Lo, C);
    else hipLaunchKernelGGL((block_out_fwd_kernel<2, false>), g, b, 0, st, y, bn, res, res_bn, o, n4, L, Lo, C);
  }
  KWS_LAUNCH_CHECK("block_out_fwd_kernel");
  return KWS_OK;
}

// internal (net_logmfcc.hip): kws_block_out_fwd and the next block's first depthwise convolution (k = 3, stride 1, pad (1, 1)) in one pass:
// o [B, Lo, C] and z [B, Lo, C]; w = that convolution's kernel [3, C]
int kws_block_out_dw_fwd(const float* y, const float* bn, const float* res, const float* res_bn, const float* w, float* o, float* z, int B,
                         int L, int C, int pool, hipStream_t st) {
  // res == NULL (pool 1 only): no residual - o = relu6(bn(y)), the activation kws_bn_relu6_apply materialises
  KWS_REQUIRE(y && bn && w && o && z && B > 0 && L > 0 && C % 4 == 0 && (pool == 1 || pool == 2) && (res || (pool == 1 && !res_bn)),
              "block_out_dw_fwd: bad arguments (L=%d C=%d pool=%d)", L, C, pool);
  const int Lo = (L + pool - 1) / pool;
  const int nchunks = ceil_div(Lo, TT);
  const int64_t threads = (int64_t)B * nchunks * (C / 4);
  int64_t grid = ceil_div64(threads, 256);
  if (grid > 4096) grid = 4096;                     // grid-stride, as the depthwise forward kernel
  KwsProfScope prof("block_out_fwd", 10.0 * B * Lo * C, 4.0 * ((double)B * L * C + (res ? 3.0 : 2.0) * B * Lo * C), st);
  dim3 g((unsigned)grid), b(256);
  if (!res) {
    hipLaunchKernelGGL((block_out_dw_fwd_kernel<1, false, false>), g, b, 0, st, y, bn, y, res_bn, w, o, z, B, L, Lo, C, nchunks);
  } else if (pool == 1) {
    if (res_bn) hipLaunchKernelGGL((block_out_dw_fwd_kernel<1, true>), g, b, 0, st, y, bn, res, res_bn, w, o, z, B, L, Lo, C, nchunks);
    else hipLaunchKernelGGL((block_out_dw_fwd_kernel<1, false>), g, b, 0, st, y, bn, res, res_bn, w, o, z, B, L, Lo, C, nchunks);
  } else {
    if (res_bn) hipLaunchKernelGGL((block_out_dw_fwd_kernel<2, true>), g, b, 0, st, y, bn, res, res_bn, w, o, z, B, L, Lo, C, nchunks);
    else hipLaunchKernelGGL((block_out_dw_fwd_kernel<2, false>), g, b, 0, st, y, bn, res, res_bn, w, o, z, B, L, Lo, C, nchunks);
  }
  KWS_LAUNCH_CHECK("block_out_dw_fwd_kernel");
  return KWS_OK;
}

int64_t kws_block_out_bwd_part_floats(int B, int L, int C, int pool) {
  if (B <= 0 || L <= 0 || !geom_ok(C) || pool < 1) return 0;
  return geom(B, (L + pool - 1) / pool, C).grid * 5 * C;
}

int kws_block_out_bwd(const float* dO, const float* y, const float* bn, float* g, float* part, int B, int L, int C,
                      int pool, int relu, hipStream_t st) {
  KWS_REQUIRE(dO && y && bn && g && part && B > 0 && L > 0 && geom_ok(C) && (pool == 1 || pool == 2) &&
                  (relu || pool == 1),
              "block_out_bwd: bad arguments (L=%d C=%d pool=%d relu=%d)", L, C, pool, relu);
  const int Lo = (L + pool - 1) / pool;
  const Geom ge = geom(B, Lo, C);
  KwsProfScope prof("block_join_bwd", 6.0 * B * L * C, 4.0 * (2.0 * B * L * C + (double)B * Lo * C), st);
  dim3 gr((unsigned)ge.grid, (unsigned)ge.ny), b((unsigned)ge.block);
  if (pool == 2) hipLaunchKernelGGL((block_out_bwd_kernel<2, true>), gr, b, 0, st, dO, y, bn, g, part, B, L, Lo, C, ge.nchunks, ge.R, ge.Cb);
  else if (relu) hipLaunchKernelGGL((block_out_bwd_kernel<1, true>), gr, b, 0, st, dO, y, bn, g, part, B, L, Lo, C, ge.nchunks, ge.R, ge.Cb);
  else hipLaunchKernelGGL((block_out_bwd_kernel<1, false>), gr, b, 0, st, dO, y, bn, g, part, B, L, Lo, C, ge.nchunks, ge.R, ge.Cb);
  KWS_LAUNCH_CHECK("block_out_bwd_kernel");
  return KWS_OK;
}

// 3-wide SAME max-pool join: L inputs -> Lo = ceil(L / stride) outputs, window of output t starts at t*stride - pad_l
// two-pass join backward (block_join_bwd_kernel): geometry, partial rows (<= 256 per channel slice) and the launcher
struct JoinGeom {
  int nchunks, R, block, ny, Cb, grid;
  int64_t units;
};
static JoinGeom join_geom(int B, int Lo, int C, int threads = 512, int tt = TT) {
  JoinGeom g;
  g.ny = ceil_div(C / 4, 256);
  g.Cb = C / g.ny;
  const int C4 = g.Cb / 4;
  g.nchunks = ceil_div(Lo, tt);
  g.R = threads / C4 < 1 ? 1 : threads / C4;
  g.block = g.R * C4;
  g.units = (int64_t)B * g.nchunks;
  const int64_t wgs = ceil_div64(g.units, g.R);
  g.grid = (int)(wgs < 256 ? wgs : 256);             // one 512-thread workgroup per CU; the rows go straight to the finaliser
  return g;
}
int kws_block_join_bwd_parts(int B, int L, int C, int pool) {
  if (B <= 0 || L <= 0 || !geom_ok(C) || pool < 1) return 0;
  return join_geom(B, (L + pool - 1) / pool, C, KWS_JOIN1_THREADS, KWS_JOIN1_TT).grid;
}
int kws_block_join_bwd(const float* dO, const float* y, const float* bn, const float* gamma, const float* coef, float* out,
                       float* part, int pass, int B, int L, int C, int pool, int relu, hipStream_t st) {
  KWS_REQUIRE(dO && y && bn && B > 0 && L > 0 && geom_ok(C) && (pool == 1 || pool == 2) && (relu || pool == 1) &&
                  ((pass == 1 && part) || (pass == 2 && gamma && coef && out)),
              "block_join_bwd: bad arguments (L=%d C=%d pool=%d relu=%d pass=%d)", L, C, pool, relu, pass);
  const int Lo = (L + pool - 1) / pool;
  const JoinGeom ge = join_geom(B, Lo, C, pass == 1 ? KWS_JOIN1_THREADS : 512, pass == 1 ? KWS_JOIN1_TT : KWS_JOIN2_TT);
  KwsProfScope prof("block_join_bwd", 6.0 * B * L * C, 4.0 * ((pass == 1 ? 1.0 : 2.0) * B * L * C + (double)B * Lo * C), st);
  dim3 gr((unsigned)ge.grid, (unsigned)ge.ny), b((unsigned)ge.block);
#define KWS_JOIN_LAUNCH(P_, RELU_)                                                                                              \
  do {                                                                                                                          \
    if (pass == 1)                                                                                                              \
      hipLaunchKernelGGL((block_join_bwd_kernel<P_, RELU_, 1, KWS_JOIN1_THREADS, KWS_JOIN1_TT>), gr, b, 0, st, dO, y, bn, gamma, coef, out, part, L, Lo, C, \
                         ge.nchunks, ge.R, ge.Cb, ge.units);                                                                    \
    else                                                                                                                        \
      hipLaunchKernelGGL((block_join_bwd_kernel<P_, RELU_, 2, 512, KWS_JOIN2_TT>), gr, b, 0, st, dO, y, bn, gamma, coef, out, part, L, Lo, C,  \
                         ge.nchunks, ge.R, ge.Cb, ge.units);                                                                    \
  } while (0)
  if (pool == 2) KWS_JOIN_LAUNCH(2, true);
  else if (relu) KWS_JOIN_LAUNCH(1, true);
  else KWS_JOIN_LAUNCH(1, false);
#undef KWS_JOIN_LAUNCH
  KWS_LAUNCH_CHECK("block_join_bwd_kernel");
  return KWS_OK;
}

int kws_block_out3_fwd(const float* y, const float* bn, const float* res, const float* res_bn, float* o, int B, int L,
                       int Lo, int C, int stride, int pad_l, hipStream_t st) {
  KWS_REQUIRE(y && bn && res && o && B > 0 && L > 0 && Lo > 0 && C % 4 == 0 && (stride == 1 || stride == 2) &&
                  pad_l >= 0 && pad_l <= 1 && (Lo - 1) * stride - pad_l < L,
              "block_out3_fwd: bad arguments (L=%d Lo=%d C=%d stride=%d pad_l=%d)", L, Lo, C, stride, pad_l);
  const int64_t n4 = (int64_t)B * Lo * C / 4;
  KwsProfScope prof("block_out_fwd", 6.0 * B * L * C, 4.0 * ((double)B * L * C + 2.0 * B * Lo * C), st);
  dim3 g((unsigned)ceil_div64(n4, 256)), b(256);
  if (res_bn) hipLaunchKernelGGL((block_out3_fwd_kernel<true>), g, b, 0, st, y, bn, res, res_bn, o, n4, L, Lo, C, stride, pad_l);
  else hipLaunchKernelGGL((block_out3_fwd_kernel<false>), g, b, 0, st, y, bn, res, res_bn, o, n4, L, Lo, C, stride, pad_l);
  KWS_LAUNCH_CHECK("block_out3_fwd_kernel");
  return KWS_OK;
}

int64_t kws_block_out3_bwd_part_floats(int B, int L, int C) {
  if (B <= 0 || L <= 0 || !geom_ok(C)) return 0;
  return geom(B, L, C).grid * 5 * C;
}

int kws_block_out3_bwd(const float* dO, const float* y, const float* bn, float* g, float* part, int B, int L, int Lo,
                       int C, int stride, int pad_l, hipStream_t st) {
  KWS_REQUIRE(dO && y && bn && g && part && B > 0 && L > 0 && Lo > 0 && geom_ok(C) && (stride == 1 || stride == 2) &&
                  pad_l >= 0 && pad_l <= 1,
              "block_out3_bwd: bad arguments (L=%d Lo=%d C=%d stride=%d pad_l=%d)", L, Lo, C, stride, pad_l);
  const Geom ge = geom(B, L, C);
  KwsProfScope prof("block_join_bwd", 12.0 * B * L * C, 4.0 * (2.0 * B * L * C + (double)B * Lo * C), st);
  dim3 gr((unsigned)ge.grid, (unsigned)ge.ny), b((unsigned)ge.block);
  hipLaunchKernelGGL(block_out3_bwd_kernel, gr, b, 0, st, dO, y, bn, g, part, B, L, Lo, C, stride, pad_l, ge.nchunks, ge.R,
                     ge.Cb);
  KWS_LAUNCH_CHECK("block_out3_bwd_kernel");
  return KWS_OK;
}

int kws_add_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t st) {
  KWS_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "add: bad arguments");
  KwsProfScope prof("add", (double)n, 12.0 * n, st);
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)ceil_div64(n / 4, 256)), dim3(256), 0, st, a, b, out, n / 4);
  KWS_LAUNCH_CHECK("add_kernel");
  return KWS_OK;
}

int kws_add_strided_f32(float* out, const float* in, int B, int L_out, int L_in, int C, int stride, hipStream_t st) {
  KWS_REQUIRE(out && in && B > 0 && C % 4 == 0 && (L_in - 1) * stride < L_out, "add_strided: bad arguments");
  const int64_t n4 = (int64_t)B * L_in * C / 4;
  KwsProfScope prof("add", (double)n4 * 4, 48.0 * n4, st);
  hipLaunchKernelGGL(add_strided_kernel, dim3((unsigned)ceil_div64(n4, 256)), dim3(256), 0, st, out, in, n4, L_out, L_in,
                     C, stride);
  KWS_LAUNCH_CHECK("add_strided_kernel");
  return KWS_OK;
}

int kws_lm_tail_fwd(const kws_lm_tail_args* a, int training, hipStream_t st) {
  KWS_REQUIRE(a->T > 0 && a->T <= LM_MAXT && a->NC > 0 && a->NC <= LM_MAXNC && a->C > 0 && a->C <= 1024,
              "lm_tail: bad shape T=%d C=%d NC=%d", a->T, a->C, a->NC);
  const LmArgs p = make_args(a);
  KwsProfScope prof(training ? "tail_train" : "tail_infer", 8.0 * a->B * a->T * a->C, 8.0 * a->B * a->T * a->C, st);
  hipLaunchKernelGGL(lm_att_logits_kernel, dim3((unsigned)a->B), dim3(256), 0, st, p);
  KWS_LAUNCH_CHECK("lm_att_logits_kernel");
  hipLaunchKernelGGL(lm_att_bn_kernel, dim3(1), dim3(LM_BN_NT), 0, st, p, training);
  KWS_LAUNCH_CHECK("lm_att_bn_kernel");
  if (training) hipLaunchKernelGGL((lm_tail_kernel<true>), dim3((unsigned)a->B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((lm_tail_kernel<false>), dim3((unsigned)a->B), dim3(256), 0, st, p);
  KWS_LAUNCH_CHECK("lm_tail_kernel");
  return KWS_OK;
}

int kws_lm_tail_bwd(const kws_lm_tail_args* a, hipStream_t st) {
  const LmArgs p = make_args(a);
  KwsProfScope prof("tail_train", 12.0 * a->B * a->T * a->C, 12.0 * a->B * a->T * a->C, st);
  hipLaunchKernelGGL(lm_att_bn_bwd_kernel, dim3(1), dim3(LM_BN_NT), 0, st, p);
  KWS_LAUNCH_CHECK("lm_att_bn_bwd_kernel");
  hipLaunchKernelGGL(lm_att_bwd_kernel, dim3((unsigned)a->B), dim3(256), 0, st, p);
  KWS_LAUNCH_CHECK("lm_att_bwd_kernel");
  return KWS_OK;
}
