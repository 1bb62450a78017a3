// Depthwise 3-tap convolution along time, channels-last, with the producer's BatchNorm+ReLU6
// applied on load.  HBM-bound: every thread owns one float4 of
// channels (16-B coalesced loads across the C dimension) and walks a short run of time steps so the
// three taps are reused from registers.
#include "dw_bwd_body.h"

namespace {
using namespace kws_dw;

#ifndef DW_FWD_GRID
#define DW_FWD_GRID 4096
#endif

template <int S, bool HAS_BN>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bn,
                                                         const float* __restrict__ w, float* __restrict__ z,
                                                         int B, int Lin, int Lout, int C, int pad_l, int nchunks,
                                                         unsigned* amax) {
  const int C4 = C >> 2;
  float zmax = 0.f;                                 // |z| maximum of this thread (fp16 x 2 GEMM arm; amax may be NULL)
  const int64_t total = (int64_t)B * nchunks * C4;
  // grid-stride over (clip, time chunk, channel quad) units with a capped grid: a few resident workgroups per
  // CU that keep streaming beat tens of thousands of short-lived ones (same finding as dwconv_bwd)
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(id % C4);
    const int64_t rest = id / C4;
    const int chunk = (int)(rest % nchunks);
    const int64_t b = rest / nchunks;
    const int c = c4 * 4;
    float4 sc = f4_zero(), sh = f4_zero();
    if (HAS_BN) {
      sc = *reinterpret_cast<const float4*>(bn + c);
      sh = *reinterpret_cast<const float4*>(bn + C + c);
    }
    const float4 w0 = *reinterpret_cast<const float4*>(w + c);
    const float4 w1 = *reinterpret_cast<const float4*>(w + C + c);
    const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * C + c);
    const float* yb = y + b * (int64_t)Lin * C + c;
    float* zb = z + b * (int64_t)Lout * C + c;
    // SAME padding pads the ACTIVATION with zeros.  All NL input rows of the unit are loaded first - unconditionally, a row
    // outside the clip reads row 0 and is zeroed after the activation - and then consumed (a load per step behind an
    // "if (t >= Lout) break" left one or two loads in flight per wave; see dwconv_bwd_kernel)
    constexpr int NL = S == 1 ? TT + 2 : 2 * TT + 1;
    const int t0 = chunk * TT;
    const int u_lo = S * t0 - pad_l;
    float4 av[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int u = u_lo + k;
      av[k] = ld4_stream(yb + (int64_t)(u >= 0 && u < Lin ? u : 0) * C);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int u = u_lo + k;
      float4 v = av[k];
      if (HAS_BN) {
        v.x = relu6f(fmaf(v.x, sc.x, sh.x));
        v.y = relu6f(fmaf(v.y, sc.y, sh.y));
        v.z = relu6f(fmaf(v.z, sc.z, sh.z));
        v.w = relu6f(fmaf(v.w, sc.w, sh.w));
      }
      av[k] = u >= 0 && u < Lin ? v : f4_zero();
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int t = t0 + i;
      if (t >= Lout) continue;
      float4 o = f4_mul(w0, av[S * i]);
      o = f4_fma(w1, av[S * i + 1], o);
      o = f4_fma(w2, av[S * i + 2], o);
      st4_stream(zb + (int64_t)t * C, o);
      zmax = kws_abs4max(zmax, o);
    }
  }
  if (amax) kws_absmax_commit(amax, zmax);
}

// The backward passes: kws_dw::bwd_body (dw_bwd_body.h) as a kernel of its own.
// -DKWS_DW_BWD2_MINW=<waves per SIMD> / -DKWS_DW_BWD1_MINW: register caps of pass 2 / of the row-writing modes (experiment, round 6: at the
// default the compiler takes 130 - 184 registers, i.e. ONE 512-thread workgroup per CU; 4 = 128 registers = two)
#ifndef KWS_DW_BWD2_MINW
#define KWS_DW_BWD2_MINW 1
#endif
#ifndef KWS_DW_BWD1_MINW
#define KWS_DW_BWD1_MINW 1
#endif
template <int S, bool HAS_BN, int MODE, bool ADDS = false>
__global__ __launch_bounds__(MODE == 2 ? DW_BWD2_THREADS : DW_BWD_THREADS, MODE == 2 ? KWS_DW_BWD2_MINW : KWS_DW_BWD1_MINW) void dwconv_bwd_kernel(BwdArgs p) {
  __shared__ float red[bwd_smem_floats(MODE, DW_BWD_THREADS)];
  bwd_body<S, HAS_BN, MODE, false, ADDS>(p, red, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

}  // namespace


namespace {
template <int MODE>
int launch_dw_bwd(const float* dz, const float* y, const float* bn, const float* w, const float* coef, float* g,
                  float* part, int B, int L_in, int L_out, int C, int stride, int pad_l, hipStream_t st,
                  unsigned* amax = nullptr, int add_stride = 0, int add_len = 0) {
  const BwdGeom ge = bwd_geom(B, L_in, C, MODE != 2);
  KWS_REQUIRE(ge.grid < 0x7FFFFFFF, "dwconv_bwd: grid too large");
  dim3 gr((unsigned)ge.grid, (unsigned)ge.ny), b((unsigned)ge.block);
  BwdArgs a;
  a.dz = dz; a.y = y; a.bn = bn; a.w = w; a.coef = coef; a.g = g; a.part = part;
  a.B = B; a.Lin = L_in; a.Lout = L_out; a.C = C; a.pad_l = pad_l; a.nchunks = ge.nchunks; a.R = ge.R; a.Cb = ge.Cb; a.amax = amax;
  a.add_stride = add_stride; a.add_len = add_len;
  if (MODE == 0 && add_stride > 1) {               // the strided-add form (no BatchNorm on the input: the residual programs' first depthwise layers)
    if (stride == 1) hipLaunchKernelGGL((dwconv_bwd_kernel<1, false, MODE, true>), gr, b, 0, st, a);
    else hipLaunchKernelGGL((dwconv_bwd_kernel<2, false, MODE, true>), gr, b, 0, st, a);
    KWS_LAUNCH_CHECK("dwconv_bwd_kernel");
    return KWS_OK;
  }
  if (stride == 1) {
    if (bn) hipLaunchKernelGGL((dwconv_bwd_kernel<1, true, MODE>), gr, b, 0, st, a);
    else hipLaunchKernelGGL((dwconv_bwd_kernel<1, false, MODE>), gr, b, 0, st, a);
  } else {
    if (bn) hipLaunchKernelGGL((dwconv_bwd_kernel<2, true, MODE>), gr, b, 0, st, a);
    else hipLaunchKernelGGL((dwconv_bwd_kernel<2, false, MODE>), gr, b, 0, st, a);
  }
  KWS_LAUNCH_CHECK("dwconv_bwd_kernel");
  return KWS_OK;
}
}  // namespace

extern "C" {

int kws_dwconv_fwd_f32(const float* y, const float* bn, const float* w, float* z, int B, int L_in, int L_out,
                       int C, int stride, int pad_l, void* stream) {
  return kws_dwconv_fwd_amax_f32(y, bn, w, z, B, L_in, L_out, C, stride, pad_l, nullptr, (hipStream_t)stream);
}

// internal: as kws_dwconv_fwd_f32, and the |z| maximum into amax (KWS_ABSMAX_WORDS words, may be NULL)
int kws_dwconv_fwd_amax_f32(const float* y, const float* bn, const float* w, float* z, int B, int L_in, int L_out,
                            int C, int stride, int pad_l, unsigned* amax, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  KWS_REQUIRE(y && w && z, "dwconv_fwd: NULL pointer");
  KWS_REQUIRE(B > 0 && L_in > 0 && L_out > 0 && C > 0 && C % 4 == 0, "dwconv_fwd: bad shape B=%d L=%d->%d C=%d",
              B, L_in, L_out, C);
  KWS_REQUIRE(stride == 1 || stride == 2, "dwconv_fwd: stride %d unsupported", stride);
  KWS_REQUIRE(pad_l >= 0 && stride * (L_out - 1) + 2 - pad_l < L_in + 2, "dwconv_fwd: geometry reads past padding");
  const int nchunks = ceil_div(L_out, TT);
  const int64_t threads = (int64_t)B * nchunks * (C / 4);
  int64_t grid = ceil_div64(threads, 256);
  if (grid > DW_FWD_GRID) grid = DW_FWD_GRID;   // grid-stride; 1024 .. uncapped measured within noise of each other
  dim3 g((unsigned)grid), b(256);
  hipStream_t st = (hipStream_t)stream;
  KwsProfScope prof("dwconv_fwd", 6.0 * B * L_out * C, 4.0 * ((double)B * L_in * C + (double)B * L_out * C), st);
  if (stride == 1) {
    if (bn) hipLaunchKernelGGL((dwconv_fwd_kernel<1, true>), g, b, 0, st, y, bn, w, z, B, L_in, L_out, C, pad_l, nchunks, amax);
    else hipLaunchKernelGGL((dwconv_fwd_kernel<1, false>), g, b, 0, st, y, bn, w, z, B, L_in, L_out, C, pad_l, nchunks, amax);
  } else {
    if (bn) hipLaunchKernelGGL((dwconv_fwd_kernel<2, true>), g, b, 0, st, y, bn, w, z, B, L_in, L_out, C, pad_l, nchunks, amax);
    else hipLaunchKernelGGL((dwconv_fwd_kernel<2, false>), g, b, 0, st, y, bn, w, z, B, L_in, L_out, C, pad_l, nchunks, amax);
  }
  KWS_LAUNCH_CHECK("dwconv_fwd_kernel");
  return KWS_OK;
}

int64_t kws_dwconv_bwd_part_floats(int B, int L_in, int C) {
  if (B <= 0 || L_in <= 0 || !bwd_geom_ok(C)) return 0;
  const BwdGeom g = bwd_geom(B, L_in, C);
  return g.grid * 5 * C;
}

int kws_dwconv_bwd_f32(const float* dz, const float* y, const float* bn, const float* w, float* g, float* part,
                       int B, int L_in, int L_out, int C, int stride, int pad_l, void* stream) {
  KWS_REQUIRE(dz && y && w && g && part, "dwconv_bwd: NULL pointer");
  KWS_REQUIRE(B > 0 && L_in > 0 && L_out > 0 && bwd_geom_ok(C),
              "dwconv_bwd: bad shape B=%d L=%d->%d C=%d", B, L_in, L_out, C);
  KWS_REQUIRE(stride == 1 || stride == 2, "dwconv_bwd: stride %d unsupported", stride);
  hipStream_t st = (hipStream_t)stream;
  KwsProfScope prof("dwconv_bwd", 12.0 * B * L_in * C, 4.0 * (2.0 * B * L_in * C + (double)B * L_out * C), st);
  return launch_dw_bwd<0>(dz, y, bn, w, nullptr, g, part, B, L_in, L_out, C, stride, pad_l, st);
}

// internal (residual-family programs): as kws_dwconv_bwd_f32 without a BatchNorm on the input, g = dgrad + add
int kws_dwconv_bwd_acc_f32(const float* dz, const float* y, const float* w, const float* add, float* g, float* part, int B,
                           int L_in, int L_out, int C, int stride, int pad_l, hipStream_t st) {
  KWS_REQUIRE(dz && y && w && add && g && part && add != g, "dwconv_bwd_acc: bad pointers");
  KWS_REQUIRE(B > 0 && L_in > 0 && L_out > 0 && bwd_geom_ok(C) && (stride == 1 || stride == 2),
              "dwconv_bwd_acc: bad shape B=%d L=%d->%d C=%d stride=%d", B, L_in, L_out, C, stride);
  KwsProfScope prof("dwconv_bwd", 13.0 * B * L_in * C, 4.0 * (3.0 * B * L_in * C + (double)B * L_out * C), st);
  return launch_dw_bwd<0>(dz, y, nullptr, w, add, g, part, B, L_in, L_out, C, stride, pad_l, st);
}

// internal (residual-family programs, round 6): g = dgrad, and row q of add [B, add_len, C] added at position add_stride q - the input
// gradient of the block's stride-s shortcut convolution joins the depthwise input gradient without an add_strided pass (the same single
// addition per element: bit-identical to the two launches)
int kws_dwconv_bwd_acc_strided_f32(const float* dz, const float* y, const float* w, const float* add, int add_stride, int add_len,
                                   float* g, float* part, int B, int L_in, int L_out, int C, int stride, int pad_l, hipStream_t st) {
  KWS_REQUIRE(dz && y && w && add && g && part && add != g, "dwconv_bwd_acc_strided: bad pointers");
  KWS_REQUIRE(B > 0 && L_in > 0 && L_out > 0 && bwd_geom_ok(C) && (stride == 1 || stride == 2) && add_stride >= 2 && add_len > 0 &&
                  (int64_t)(add_len - 1) * add_stride < L_in,
              "dwconv_bwd_acc_strided: bad shape B=%d L=%d->%d C=%d stride=%d add %d x %d", B, L_in, L_out, C, stride, add_len, add_stride);
  KwsProfScope prof("dwconv_bwd", 13.0 * B * L_in * C, 4.0 * (2.0 * B * L_in * C + (double)B * L_out * C + (double)B * add_len * C), st);
  return launch_dw_bwd<0>(dz, y, nullptr, w, add, g, part, B, L_in, L_out, C, stride, pad_l, st, nullptr, add_stride, add_len);
}

int kws_dwconv_bwd_bn_f32(const float* dz, const float* y, const float* bn, const float* w, const float* coef,
                          float* dy, float* part, int pass, int B, int L_in, int L_out, int C, int stride, int pad_l,
                          void* stream) {
  return kws_dwconv_bwd_bn_amax_f32(dz, y, bn, w, coef, dy, part, pass, B, L_in, L_out, C, stride, pad_l, nullptr,
                                    (hipStream_t)stream);
}

// internal: as kws_dwconv_bwd_bn_f32; pass 2 also leaves the |dy| maximum in amax (may be NULL)
int kws_dwconv_bwd_bn_amax_f32(const float* dz, const float* y, const float* bn, const float* w, const float* coef,
                               float* dy, float* part, int pass, int B, int L_in, int L_out, int C, int stride, int pad_l,
                               unsigned* amax, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  KWS_REQUIRE(dz && y && bn && w, "dwconv_bwd_bn: NULL pointer");
  KWS_REQUIRE(pass == 1 ? part != nullptr : (pass == 2 && coef != nullptr && dy != nullptr),
              "dwconv_bwd_bn: pass %d needs %s", pass, pass == 1 ? "part" : "coef and dy");
  KWS_REQUIRE(B > 0 && L_in > 0 && L_out > 0 && bwd_geom_ok(C),
              "dwconv_bwd_bn: bad shape B=%d L=%d->%d C=%d", B, L_in, L_out, C);
  KWS_REQUIRE(stride == 1 || stride == 2, "dwconv_bwd_bn: stride %d unsupported", stride);
  hipStream_t st = (hipStream_t)stream;
  KwsProfScope prof("dwconv_bwd", 12.0 * B * L_in * C,
                    4.0 * ((pass == 2 ? 2.0 : 1.0) * B * L_in * C + (double)B * L_out * C), st);
  if (pass == 1) return launch_dw_bwd<1>(dz, y, bn, w, nullptr, nullptr, part, B, L_in, L_out, C, stride, pad_l, st);
  return launch_dw_bwd<2>(dz, y, bn, w, coef, dy, nullptr, B, L_in, L_out, C, stride, pad_l, st, amax);
}

}  // extern "C"
