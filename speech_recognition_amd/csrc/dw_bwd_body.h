// The depthwise-backward pass (reference model.py:34-51) as a device function of a block id:
// dwconv.hip's kernels run it on their own grids.  (Round 5 also ran it on the first blocks of a grid that held weight-gradient
// work items - a measured loss, retired in round 6: scripts/probes/wgrad_beside_dwbwd/ - which is why it takes (bid, nblk, by)
// instead of reading blockIdx.)
#pragma once
#include "internal.h"

namespace kws_dw {

// Streamed operands (read or written exactly once per launch): KWS_DW_NT bit 1 marks the stores, bit 2 the loads
// non-temporal.  Measured over the eleven layers at batch 1024 (scripts/bench_dwconv.py): stores non-temporal 357 -> 333 us
// forward (6.06 TB/s) and 584 -> 532 us backward pass 2 (5.96 TB/s), pass 1 (no tensor store) unchanged; loads non-temporal
// 7 - 10 % SLOWER (the k = 3 halo re-reads want the cache).  In the bench step 4.82 -> 4.78 ms.  Default: stores only.
#ifndef KWS_DW_NT
#define KWS_DW_NT 1
#endif
typedef float dw_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
  if (KWS_DW_NT & 2) {
    const dw_v4f v = __builtin_nontemporal_load(reinterpret_cast<const dw_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4_stream(float* p, float4 o) {
  if (KWS_DW_NT & 1) {
    dw_v4f v = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(v, reinterpret_cast<dw_v4f*>(p));
    return;
  }
  *reinterpret_cast<float4*>(p) = o;
}

#ifndef KWS_DW_TT
#define KWS_DW_TT 8
#endif
constexpr int TT = KWS_DW_TT;  // time steps per thread

__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// DW_BWD_THREADS threads per workgroup of the modes that write partial rows (0 and 1): 1024 = one workgroup per CU carries
// the 16 waves that stream best (as 256-thread workgroups that took 1024 of them, i.e. 1024 partial rows and a slice_reduce
// launch in front of every finalise kernel; with <= 256 rows kws_dw_bwd_finalize sums them directly - one 5 us launch less
// per block and step: 0.21 -> 0.185 ms of small kernels).  Pass 2 (MODE 2, no partial rows) keeps 256-thread workgroups:
// measured with 1024 it streams 4 % slower (533 -> 556 us over the eleven layers).
#ifndef DW_BWD_THREADS
#define DW_BWD_THREADS 512
#endif
#ifndef DW_BWD_HT
#define DW_BWD_HT 8      // positions per batch of loads (divides TT)
#endif
#ifndef DW_BWD2_THREADS
#define DW_BWD2_THREADS 512
#endif
#ifndef DW_BWD_PARTS
#define DW_BWD_PARTS 256
#endif

struct BwdArgs {
  const float* dz; const float* y; const float* bn; const float* w; const float* coef;
  float* g; float* part;
  int B, Lin, Lout, C, pad_l, nchunks, R, Cb;
  unsigned* amax;
  // MODE 0's added tensor (`coef`): add_stride <= 1: the shape of g, added everywhere; s > 1: [B, add_len, C], row q added at position s q
  // (the input gradient of a stride-s shortcut convolution joining the depthwise input gradient: round 6, instead of an add_strided pass)
  int add_stride = 0, add_len = 0;
};
// LDS floats a workgroup of `threads` threads needs for one pass (pass 2 reduces nothing)
constexpr int bwd_smem_floats(int mode, int threads) { return mode == 2 ? 4 : 5 * threads * 4; }

// Backward: one thread per (clip, run of TT input positions, float4 of channels).
// part[block][5][C] = per-block sums of (g, g*xhat, dz*a@tap0, dz*a@tap1, dz*a@tap2).
// MODE 0: store g and the partial sums (kws_dwconv_bwd_f32).
// MODE 1 / 2: the two passes of kws_dwconv_bwd_bn_f32, which never materialises g: pass 1 only reduces, pass 2
//   recomputes g from the same operands (bit-identical) and stores the BatchNorm input gradient
//   dy = scale * (g - c1 - xhat * c2) directly - 5 tensor passes instead of the 6 of "store g, then
//   kws_bn_bwd_apply" for a stride-1 layer, 4 instead of 5.5 for a stride-2 layer.
// The block is (bid, nblk, by) instead of (blockIdx.x, gridDim.x, blockIdx.y) and `red` is the caller's LDS
// (bwd_smem_floats): a kernel that holds other work too hands this body the blocks it sets aside for it.  A caller may
// walk several block ids one after the other (a barrier between them): block `bid` of `nblk` does exactly what the
// workgroup with that index of a `nblk`-workgroup launch does, down to the order of every sum.
// GUARD: the launching workgroup may be wider than the R * C4 threads this shape uses; the extra threads only keep the barrier.
// ADDS (MODE 0): the added tensor is the coarser one (add_stride > 1) - a compile-time form, so that the plain add keeps its code
template <int S, bool HAS_BN, int MODE, bool GUARD, bool ADDS = false>
__device__ __forceinline__ void bwd_body(const BwdArgs& p, float* const red, const int bid, const int nblk, const int by) {
  const float* __restrict__ dz = p.dz;
  const float* __restrict__ y = p.y;
  const float* __restrict__ bn = p.bn;
  const float* __restrict__ w = p.w;
  const float* __restrict__ coef = p.coef;
  float* __restrict__ g = p.g;
  float* __restrict__ part = p.part;
  const int B = p.B, Lin = p.Lin, Lout = p.Lout, C = p.C, pad_l = p.pad_l, nchunks = p.nchunks, R = p.R, Cb = p.Cb;
  constexpr int RSTRIDE = (MODE == 2 ? 4 : DW_BWD_THREADS * 4);   // floats per quantity in `red`
  float gmax = 0.f;                                 // MODE 2: |dy| maximum of this thread (amax may be NULL)
  // `by` selects a slice of Cb <= 1024 channels (Cb = C unless C > 1024)
  const int C4 = Cb >> 2;
  const int tid = threadIdx.x;
  const bool live = !GUARD || tid < R * C4;
  const int r = live ? tid / C4 : 0, c4 = live ? tid - r * C4 : 0;
  const int c = by * Cb + c4 * 4;
  float4 sg = f4_zero(), sgx = f4_zero(), sw0 = f4_zero(), sw1 = f4_zero(), sw2 = f4_zero();
  float4 sc = f4_zero(), sh = f4_zero(), mean = f4_zero(), rstd = f4_zero();
  if (HAS_BN) {
    sc = *reinterpret_cast<const float4*>(bn + c);
    sh = *reinterpret_cast<const float4*>(bn + C + c);
    mean = *reinterpret_cast<const float4*>(bn + 2 * C + c);
    rstd = *reinterpret_cast<const float4*>(bn + 3 * C + c);
  }
  const float4 w0 = *reinterpret_cast<const float4*>(w + c);
  const float4 w1 = *reinterpret_cast<const float4*>(w + C + c);
  const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * C + c);
  float4 c1 = f4_zero(), c2 = f4_zero();
  if (MODE == 2) {
    c1 = *reinterpret_cast<const float4*>(coef + c);
    c2 = *reinterpret_cast<const float4*>(coef + C + c);
  }
  // grid-stride over the (clip, time-chunk) units: the grid is capped so that a launch leaves at most
  // KWS_DW_BWD_MAX_PARTS partial rows (summed per thread in unit order), which kws_dw_bwd_finalize folds
  // without a pre-reduction pass
  for (int64_t ub = bid; ub * R < (int64_t)B * nchunks; ub += nblk) {
    const int64_t unit = ub * R + r;
    const int64_t b = unit / nchunks;
    const int chunk = (int)(unit - b * nchunks);
    if (b >= B || !live) continue;
    const float* yb = y + b * (int64_t)Lin * C + c;
    float* gb = g + b * (int64_t)Lin * C + c;
    const float* dzb = dz + b * (int64_t)Lout * C + c;
    const int u0 = chunk * TT;
    // The unit's TT positions in batches of HT: ALL loads of a batch are issued first - unconditionally, from clamped
    // addresses, zeroed afterwards where the tap falls outside the clip - and only then consumed.  (Written as "load when
    // inside, use, next position" the compiler put a vmcnt(0) wait behind every position's pair of loads: two 16-byte loads
    // in flight per wave, eight dependent round trips per unit - 5.1 TB/s for pass 1 where the forward kernel streams 6.1.)
    constexpr int HT = DW_BWD_HT;
    constexpr int ND = S == 1 ? HT + 2 : HT / 2 + 2;  // dz rows a batch touches
    const bool podd = (pad_l & 1) != 0;               // S == 2: parity of (u + pad_l) at the even positions of a batch
#pragma unroll
    for (int h = 0; h < TT / HT; ++h) {
      const int ub = u0 + h * HT;                      // even (u0 is a multiple of TT)
      if (ub >= Lin) break;
      float4 yv[HT], D[ND];
#pragma unroll
      for (int i = 0; i < HT; ++i) yv[i] = ld4_stream(yb + (int64_t)(ub + i < Lin ? ub + i : ub) * C);
      // S == 1: tap j of position u reads dz[u + pad_l - j]: rows ub + pad_l - 2 ... ub + pad_l + HT - 1
      // S == 2: tap j reads dz[(u + pad_l - j) / 2] when that is whole: rows ((ub + pad_l) >> 1) - 1 ... + ND - 1
      const int t_lo = S == 1 ? ub + pad_l - 2 : ((ub + pad_l) >> 1) - 1;
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        const int t = t_lo + j;
        D[j] = ld4_stream(dzb + (int64_t)(t < 0 ? 0 : (t < Lout ? t : Lout - 1)) * C);
      }
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        const int t = t_lo + j;
        if (t < 0 || t >= Lout) D[j] = f4_zero();
      }
#pragma unroll
      for (int i = 0; i < HT; ++i) {
        const int u = ub + i;
        if (u >= Lin) continue;
        float4 d0, d1, d2;
        if (S == 1) {
          d0 = D[i + 2]; d1 = D[i + 1]; d2 = D[i];
        } else if ((i & 1) == 0) {                     // u + pad_l even <=> !podd: taps 0 and 2; odd: tap 1
          d0 = podd ? f4_zero() : D[i / 2 + 1];
          d2 = podd ? f4_zero() : D[i / 2];
          d1 = podd ? D[i / 2 + 1] : f4_zero();
        } else {
          d0 = podd ? D[(i + 1) / 2 + 1] : f4_zero();
          d2 = podd ? D[(i + 1) / 2] : f4_zero();
          d1 = podd ? f4_zero() : D[(i - 1) / 2 + 1];
        }
        const float4 yy = yv[i];
        float4 a = yy, mk = make_float4(1.f, 1.f, 1.f, 1.f), xh = f4_zero();
        if (HAS_BN) {
          const float4 pre = make_float4(fmaf(yy.x, sc.x, sh.x), fmaf(yy.y, sc.y, sh.y), fmaf(yy.z, sc.z, sh.z),
                                         fmaf(yy.w, sc.w, sh.w));
          a = make_float4(relu6f(pre.x), relu6f(pre.y), relu6f(pre.z), relu6f(pre.w));
          mk = make_float4((pre.x > 0.f && pre.x <= 6.f) ? 1.f : 0.f, (pre.y > 0.f && pre.y <= 6.f) ? 1.f : 0.f,
                           (pre.z > 0.f && pre.z <= 6.f) ? 1.f : 0.f, (pre.w > 0.f && pre.w <= 6.f) ? 1.f : 0.f);
          xh = make_float4((yy.x - mean.x) * rstd.x, (yy.y - mean.y) * rstd.y, (yy.z - mean.z) * rstd.z,
                           (yy.w - mean.w) * rstd.w);
        }
        float4 da = f4_mul(w0, d0);
        da = f4_fma(w1, d1, da);
        da = f4_fma(w2, d2, da);
        const float4 gv = f4_mul(da, mk);
        if (MODE == 2) {
          // same expression as bn_bwd_apply_kernel (bn.hip), with scale = gamma * rstd from the BN table
          float4 o;
          o.x = sc.x * (gv.x - c1.x - (yy.x - mean.x) * rstd.x * c2.x);
          o.y = sc.y * (gv.y - c1.y - (yy.y - mean.y) * rstd.y * c2.y);
          o.z = sc.z * (gv.z - c1.z - (yy.z - mean.z) * rstd.z * c2.z);
          o.w = sc.w * (gv.w - c1.w - (yy.w - mean.w) * rstd.w * c2.w);
          st4_stream(gb + (int64_t)u * C, o);
          gmax = kws_abs4max(gmax, o);
          continue;
        }
        if (MODE == 0) {
          float4 o = gv;
          if (coef != nullptr) {   // MODE 0 reuses `coef` as an optional tensor added to the input gradient (residual join)
            if (!ADDS) {
              const float4 ad = *reinterpret_cast<const float4*>(coef + (b * (int64_t)Lin + u) * C + c);
              o = make_float4(o.x + ad.x, o.y + ad.y, o.z + ad.z, o.w + ad.w);
            } else {               // every add_stride-th position receives a row of the coarser tensor (address selected, not branched)
              const int q = u / p.add_stride;
              const bool on = q * p.add_stride == u && q < p.add_len;
              const float4 ad = *reinterpret_cast<const float4*>(coef + (b * (int64_t)p.add_len + (on ? q : 0)) * C + c);
              o = make_float4(on ? o.x + ad.x : o.x, on ? o.y + ad.y : o.y, on ? o.z + ad.z : o.z, on ? o.w + ad.w : o.w);
            }
          }
          st4_stream(gb + (int64_t)u * C, o);
        }
        sg.x += gv.x; sg.y += gv.y; sg.z += gv.z; sg.w += gv.w;
        sgx = f4_fma(gv, xh, sgx);
        sw0 = f4_fma(d0, a, sw0);
        sw1 = f4_fma(d1, a, sw1);
        sw2 = f4_fma(d2, a, sw2);
      }
    }
  }
  if (MODE == 2) {
    if (p.amax && live) kws_absmax_commit(p.amax, gmax);
    return;
  }
  if constexpr (MODE != 2) {
  if (live) {
    *reinterpret_cast<float4*>(&red[0 * RSTRIDE + tid * 4]) = sg;
    *reinterpret_cast<float4*>(&red[1 * RSTRIDE + tid * 4]) = sgx;
    *reinterpret_cast<float4*>(&red[2 * RSTRIDE + tid * 4]) = sw0;
    *reinterpret_cast<float4*>(&red[3 * RSTRIDE + tid * 4]) = sw1;
    *reinterpret_cast<float4*>(&red[4 * RSTRIDE + tid * 4]) = sw2;
  }
  __syncthreads();
  // fixed-order reduction over the R rows of this block: thread (q, channel)
  for (int o = tid; o < 5 * Cb; o += blockDim.x) {
    const int q = o / Cb, ch = o - q * Cb;
    float s = 0.f;
    for (int rr = 0; rr < R; ++rr) s += red[q * RSTRIDE + (rr * C4) * 4 + ch];
    part[((int64_t)bid * 5 + q) * C + by * Cb + ch] = s;
  }
  }
}

// measured at batch 1024 (ms per step, dwconv_bwd + finalisation): 256 rows 1.37, 512 1.02, 1024 0.86, 2048 0.91,
// uncapped (6400) 1.03: four resident workgroups per CU stream best and leave few rows to fold
// (the table above is for 256-thread workgroups; DW_BWD_THREADS = 1024 reaches the rate of its 1024-row entry with 256)
constexpr int KWS_DW_BWD_MAX_PARTS = DW_BWD_PARTS;
struct BwdGeom {
  int nchunks, R, block, ny, Cb;
  int64_t grid;
};
inline bool bwd_geom_ok(int C) { return C > 0 && C % 4 == 0 && (C / 4) % ceil_div(C / 4, 256) == 0; }
inline BwdGeom bwd_geom(int B, int Lin, int C, bool parts = true) {
  BwdGeom g;
  const int threads = parts ? DW_BWD_THREADS : DW_BWD2_THREADS;
  const int max_parts = parts ? KWS_DW_BWD_MAX_PARTS : 1024;
  g.ny = ceil_div(C / 4, 256);   // channel slices of at most 1024 channels
  g.Cb = C / g.ny;
  const int C4 = g.Cb / 4;
  g.nchunks = ceil_div(Lin, TT);
  g.R = threads / C4;
  if (g.R < 1) g.R = 1;
  g.block = g.R * C4;
  g.grid = ceil_div64((int64_t)B * g.nchunks, g.R);
  if (g.grid > max_parts) g.grid = max_parts;
  return g;
}

}  // namespace kws_dw
