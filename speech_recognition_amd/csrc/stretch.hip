// Speed-TTA time stretch on the device.
//
// Replaces the reference's offline pass create_tta_set.py:9-22 (librosa.effects.time_stretch(data, 0.9),
// keep the last 16000 samples, int16 round trip through a wav file) whose output make_submission.py:86-100,
// 133-136 feeds to the three "slow" predict passes.  librosa's algorithm (0.5.x defaults: STFT 2048/512,
// periodic Hann, reflect-padded centre frames -> phase vocoder -> ISTFT with window-sum-square
// normalisation, trimmed by 1024 on both sides) is restated in oracle/stretch.py.
//
// One 256-thread workgroup per clip streams through the output frames; nothing but the PCM in and the
// kept samples out touches HBM (64 kB + 64 kB per clip):
//   * a real 2048-point transform is a 1024-point complex radix-4 Stockham FFT (5 passes, thread j owns
//     elements j + 256 r of every pass, its 12 twiddles live in registers) plus a split step; the first
//     forward pass reads its operands straight from the windowed PCM and the last inverse pass leaves
//     samples 2j + 512 r + {0,1} in the registers of thread j, so the analysis window, the synthesis window,
//     the overlap-add accumulator and the window-sum-square normaliser are all thread-private registers;
//   * the split step makes thread j the owner of bins {j, 1024-j, j+256, 768-j} (thread 0 also 0, 1024
//     and 512) for the whole clip: the two STFT columns the vocoder interpolates between and the phase
//     accumulator stay in registers as well.  The accumulator is carried as a unit complex number,
//     u <- u * unit(c1) * conj(unit(c0)), which is exp(1j * phase_acc) of librosa's loop without any
//     atan2 / sincos and without the float32 accumulator growing to 6e4 rad (oracle/stretch.py header);
//   * LDS holds only the two FFT ping-pong buffers (16 kB).
#include <math.h>

#include <vector>

#include "common.h"
#include "internal.h"

struct kws_stretch_plan {
  int n_samples = 0, n_frames = 0, n_steps = 0;
  double rate = 0;
  float2* tw = nullptr;   // [1024] e^{-2 pi j m / 1024}
  float2* tws = nullptr;  // [1025] e^{-2 pi j k / 2048}
  float* win = nullptr;   // [2048] periodic Hann
  int* sidx = nullptr;    // [n_steps] int(step)
  float* salpha = nullptr;  // [n_steps] step mod 1
};

namespace {

constexpr int NFFT = 2048, HOP = 512, MC = 1024, NT = 256;
#ifndef KWS_STRETCH_WPE
#define KWS_STRETCH_WPE 4  // waves per SIMD the register budget is sized for
#endif

struct StretchArgs {
  const void* x;
  float* out;
  const float2* tw;
  const float2* tws;
  const float* win;
  const int* sidx;
  const float* salpha;
  int B, L, n_frames, T, keep, quantize;
  float in_scale;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// natural-order 4-point DFT; INV selects e^{+...}
template <bool INV>
__device__ __forceinline__ void fft4(float2 (&u)[4]) {
  const float2 s02 = cadd(u[0], u[2]), d02 = csub(u[0], u[2]);
  const float2 s13 = cadd(u[1], u[3]), d13 = csub(u[1], u[3]);
  const float2 jd = make_float2(-d13.y, d13.x);  // j * d13
  u[0] = cadd(s02, s13);
  u[2] = csub(s02, s13);
  if (INV) {
    u[1] = cadd(d02, jd);
    u[3] = csub(d02, jd);
  } else {
    u[1] = csub(d02, jd);
    u[3] = cadd(d02, jd);
  }
}

// Passes of the 1024-point Stockham FFT.  `tw` = e^{-2 pi j m / 1024}, m < 768, in LDS (pass s of thread j uses
// entries r * (j mod 4^s) * 4^(4-s): holding the 12 of them in registers costs more occupancy than the reads).
// FIRST_REGS: pass 0 takes its operands from u (else from src); LAST_REGS: pass 4 leaves its results in u
// (else writes dst).  Returns with the result in `u` or in the buffer that was written last; buffers
// alternate a -> b -> a ...  One barrier after every LDS write.
template <bool INV, bool FIRST_REGS, bool LAST_REGS>
__device__ __forceinline__ float2* fft1024(float2 (&u)[4], float2* a, float2* b, const float2* tw, int j) {
  float2* src = a;
  float2* dst = b;
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int Ns = 1 << (2 * s);
    if (!(s == 0 && FIRST_REGS)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) u[r] = src[j + NT * r];
    }
    if (s > 0) {
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        const float2 tq = tw[r * (j & (Ns - 1)) * (NT / Ns)];
        u[r] = INV ? cmulc(u[r], tq) : cmul(u[r], tq);
      }
    }
    fft4<INV>(u);
    if (s == 4 && LAST_REGS) return src;
    const int k = j & (Ns - 1);
    const int j0 = ((j - k) << 2) + k;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[j0 + r * Ns] = u[r];
    __syncthreads();
    float2* t = src;
    src = dst;
    dst = t;
  }
  return src;  // holds the result
}

template <typename TIn>
__device__ __forceinline__ float load_pcm(const TIn* x, int o, float scale);
template <>
__device__ __forceinline__ float load_pcm<float>(const float* x, int o, float scale) {
  return x[o] * scale;
}
template <>
__device__ __forceinline__ float load_pcm<int16_t>(const int16_t* x, int o, float scale) {
  return (float)x[o] / scale;  // create_tta_set.py:18  np.float32(data) / 32767
}

// One STFT bin in the form the vocoder consumes: magnitude and unit phasor (angle(0) = 0 -> phasor 1).
struct Polar {
  float m;
  float2 n;
};
__device__ __forceinline__ Polar to_polar(float2 c) {
  const float s = c.x * c.x + c.y * c.y;
  const bool nz = s > 1e-37f;
  const float inv = nz ? __builtin_amdgcn_rsqf(s) : 0.f;
  Polar p;
  p.m = s * inv;
  p.n = nz ? make_float2(c.x * inv, c.y * inv) : make_float2(1.f, 0.f);
  return p;
}
// one vocoder step of a bin: returns the synthesis bin, advances the accumulator u by angle(b) - angle(a)
__device__ __forceinline__ float2 vocode(const Polar& a, const Polar& b, float al, float2& u) {
  const float mag = (1.f - al) * a.m + al * b.m;
  const float2 y = make_float2(mag * u.x, mag * u.y);
  const float2 un = cmulc(cmul(u, b.n), a.n);
  const float h = 1.5f - 0.5f * (un.x * un.x + un.y * un.y);  // Newton step back to |u| = 1
  u = make_float2(un.x * h, un.y * h);
  return y;
}

// Bin ownership of thread j for the whole clip: slot 0 -> bin j (j = 0: DC), slot 1 -> 1024 - j (j = 0: Nyquist),
// slot 2 -> j + 256, slot 3 -> 768 - j.  Bin 512 (its own mirror) belongs to thread 0 and lives in LDS.
constexpr int NSLOT = 4;
struct Column {
  Polar p[NSLOT];
};

template <typename TIn>
__global__ __launch_bounds__(NT, KWS_STRETCH_WPE) void stretch_kernel(StretchArgs a) {
  __shared__ float2 bufA[MC];
  __shared__ float2 bufB[MC];
  __shared__ float2 twl[768];
  __shared__ Polar mid[2];  // bin 512 of the two live columns (thread 0 only)
  __shared__ float2 mid_u;
  const int j = threadIdx.x;
  const int clip = blockIdx.x;
  const TIn* x = reinterpret_cast<const TIn*>(a.x) + (size_t)clip * a.L;
  float* out = a.out + (size_t)clip * a.keep;

  // thread-private tables
  float w[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w[r][0] = a.win[2 * j + HOP * r];
    w[r][1] = a.win[2 * j + HOP * r + 1];
  }
  for (int m = j; m < 768; m += NT) twl[m] = a.tw[m];
  __syncthreads();
  const float2 ws0 = a.tws[j], ws1 = a.tws[j + NT];

  Column Ca, Cb;
  float2 U[NSLOT];
  float ola[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) ola[r][0] = ola[r][1] = 0.f;

  // ---- analysis: STFT column f into C (zeros beyond the last frame: the two padded columns) ----
  auto analyse = [&](int f, Column& C, Polar& C512) {
    if (f >= a.n_frames) {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) C.p[i] = to_polar(make_float2(0.f, 0.f));
      if (j == 0) C512 = to_polar(make_float2(0.f, 0.f));
      return;
    }
    float2 u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int o = HOP * f + 2 * j + HOP * r + e - NFFT / 2;  // np.pad(y, 1024, mode='reflect')
        o = o < 0 ? -o : o;
        o = o >= a.L ? 2 * (a.L - 1) - o : o;
        v[e] = load_pcm<TIn>(x, o, a.in_scale) * w[r][e];
      }
      u[r] = make_float2(v[0], v[1]);
    }
    const float2* Z = fft1024<false, true, false>(u, bufA, bufB, twl, j);
    // split: X[k] = E + T, X[1024-k] = conj(E - T), E = (Z[k] + conj Z[M-k]) / 2, T = -j W^k (Z[k] - conj Z[M-k]) / 2
    auto split = [&](int k, float2 wk, Polar& Xk, Polar& Xm) {
      const float2 za = Z[k], zb = cconj(Z[(MC - k) & (MC - 1)]);
      const float2 E = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y + zb.y));
      const float2 D = make_float2(0.5f * (za.x - zb.x), 0.5f * (za.y - zb.y));
      const float2 wd = cmul(wk, D);
      const float2 T = make_float2(wd.y, -wd.x);  // -j * wd
      Xk = to_polar(cadd(E, T));
      Xm = to_polar(cconj(csub(E, T)));
    };
    split(j, ws0, C.p[0], C.p[1]);
    if (j == 0) {
      const float2 z0 = Z[0];
      C.p[0] = to_polar(make_float2(z0.x + z0.y, 0.f));  // DC and Nyquist are real
      C.p[1] = to_polar(make_float2(z0.x - z0.y, 0.f));
      const float2 z5 = Z[512];                           // W^512 = -j:  X[512] = conj(Z[512])
      C512 = to_polar(cconj(z5));
    }
    split(j + NT, ws1, C.p[2], C.p[3]);
    __syncthreads();  // Z fully consumed before the buffers are reused
  };

  analyse(0, Ca, mid[0]);
  analyse(1, Cb, mid[1]);
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) U[i] = Ca.p[i].n;
  if (j == 0) mid_u = mid[0].n;
  int cur = 0;  // Ca = column cur, Cb = column cur + 1; mid[cur & 1], mid[(cur + 1) & 1]

  const int Ltrim = HOP * (a.T - 1);
  const int skip = Ltrim > a.keep ? Ltrim - a.keep : 0;

  auto emit = [&](int t) {
    // positions 512 t + 2 j + e are final; normaliser = window sum-square over the frames covering them
    const int m_lo = t - (a.T - 1) > 0 ? t - (a.T - 1) : 0;
    const int m_hi = t < 3 ? t : 3;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float ss = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m >= m_lo && m <= m_hi) ss += w[m][e] * w[m][e];
      float v = ola[0][e];
      if (ss > 1.17549435e-38f) v = v / ss;
      const int r = HOP * t + 2 * j + e - NFFT / 2;
      const int o = r - skip;
      if (r < Ltrim && o >= 0 && o < a.keep) {
        if (a.quantize) {
          // np.int16(data * 32767) (C cast: truncation, wraps beyond int16) then DecodeWav's / 32768
          const int q = (int)(v * 32767.f);
          v = (float)(int16_t)q * (1.f / 32768.f);
        }
        out[o] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ola[r][0] = ola[r + 1][0];
      ola[r][1] = ola[r + 1][1];
    }
    ola[3][0] = ola[3][1] = 0.f;
  };

  for (int t = 0; t < a.T; ++t) {
    const int s = a.sidx[t];
    const float al = a.salpha[t];
    while (cur < s) {
      Ca = Cb;
      analyse(cur + 2, Cb, mid[cur & 1]);  // column cur + 2 replaces column cur
      ++cur;
    }
    // ---- phase vocoder step on the owned bins, then the inverse pack:
    //      Z'[k] = E + jO, Z'[M-k] = conj(E - jO), E = (X_k + conj X_m)/2, O = conj(W^k) (X_k - conj X_m)/2
    auto pack = [&](int k, float2 wk, float2 Xk, float2 Xm) {
      const float2 xm = cconj(Xm);
      const float2 E = make_float2(0.5f * (Xk.x + xm.x), 0.5f * (Xk.y + xm.y));
      const float2 T = make_float2(0.5f * (Xk.x - xm.x), 0.5f * (Xk.y - xm.y));
      const float2 O = cmulc(T, wk);
      const float2 jO = make_float2(-O.y, O.x);
      bufA[k] = cadd(E, jO);
      bufA[(MC - k) & (MC - 1)] = cconj(csub(E, jO));
    };
    {
      const float2 y0 = vocode(Ca.p[0], Cb.p[0], al, U[0]);
      const float2 y1 = vocode(Ca.p[1], Cb.p[1], al, U[1]);
      if (j != 0) pack(j, ws0, y0, y1);
      const float2 y2 = vocode(Ca.p[2], Cb.p[2], al, U[2]);
      const float2 y3 = vocode(Ca.p[3], Cb.p[3], al, U[3]);
      pack(j + NT, ws1, y2, y3);
      if (j == 0) {
        bufA[0] = make_float2(0.5f * (y0.x + y1.x), 0.5f * (y0.x - y1.x));  // irfft ignores Im of DC / Nyquist
        float2 u5 = mid_u;
        const float2 y5 = vocode(mid[cur & 1], mid[(cur + 1) & 1], al, u5);
        mid_u = u5;
        bufA[512] = cconj(y5);
      }
    }
    __syncthreads();
    float2 u[4];
    fft1024<true, false, true>(u, bufA, bufB, twl, j);
    __syncthreads();  // every thread has read its last-pass operands before bufA is packed again
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ola[r][0] += u[r].x * (1.f / MC) * w[r][0];
      ola[r][1] += u[r].y * (1.f / MC) * w[r][1];
    }
    emit(t);
  }
  for (int t = a.T; t < a.T + 3; ++t) emit(t);
  // shorter than `keep`: DecodeWav pads the file with zeros at the end (input_data.py:335-336)
  for (int o = Ltrim + j; o < a.keep; o += NT) out[o] = 0.f;
}

}  // namespace

extern "C" {

int kws_stretch_plan_create(int n_samples, double rate, kws_stretch_plan_t** plan) {
  KWS_REQUIRE(plan, "stretch_plan_create: NULL pointer");
  KWS_REQUIRE(rate > 0.0, "stretch_plan_create: rate must be a positive number");  // librosa ParameterError
  KWS_REQUIRE(n_samples > NFFT / 2, "stretch_plan_create: n_samples=%d (reflect padding needs > %d)", n_samples,
              NFFT / 2);
  kws_stretch_plan* p = new kws_stretch_plan();
  p->n_samples = n_samples;
  p->rate = rate;
  p->n_frames = 1 + n_samples / HOP;
  // np.arange(0, n_frames, rate): ceil(n_frames / rate) values i * rate
  std::vector<int> sidx;
  std::vector<float> salpha;
  const int n_steps = (int)ceil((double)p->n_frames / rate);
  for (int i = 0; i < n_steps; ++i) {
    const double step = i * rate;
    sidx.push_back((int)step);
    salpha.push_back((float)fmod(step, 1.0));
  }
  p->n_steps = n_steps;
  std::vector<float2> tw(MC), tws(MC + 1);
  for (int m = 0; m < MC; ++m) {
    const double ang = -2.0 * M_PI * m / MC;
    tw[m] = make_float2((float)cos(ang), (float)sin(ang));
  }
  for (int k = 0; k <= MC; ++k) {
    const double ang = -2.0 * M_PI * k / NFFT;
    tws[k] = make_float2((float)cos(ang), (float)sin(ang));
  }
  std::vector<float> win(NFFT);
  for (int i = 0; i < NFFT; ++i) win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / NFFT));
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    KWS_HIP(hipMalloc(dst, bytes));
    KWS_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return KWS_OK;
  };
  int rc = up((void**)&p->tw, tw.data(), tw.size() * sizeof(float2));
  if (rc == KWS_OK) rc = up((void**)&p->tws, tws.data(), tws.size() * sizeof(float2));
  if (rc == KWS_OK) rc = up((void**)&p->win, win.data(), win.size() * sizeof(float));
  if (rc == KWS_OK) rc = up((void**)&p->sidx, sidx.data(), sidx.size() * sizeof(int));
  if (rc == KWS_OK) rc = up((void**)&p->salpha, salpha.data(), salpha.size() * sizeof(float));
  if (rc != KWS_OK) {
    kws_stretch_plan_destroy(p);
    return rc;
  }
  *plan = p;
  return KWS_OK;
}

int kws_stretch_plan_destroy(kws_stretch_plan_t* p) {
  if (!p) return KWS_OK;
  (void)hipFree(p->tw);
  (void)hipFree(p->tws);
  (void)hipFree(p->win);
  (void)hipFree(p->sidx);
  (void)hipFree(p->salpha);
  delete p;
  return KWS_OK;
}

int kws_stretch_out_samples(const kws_stretch_plan_t* p) { return p ? HOP * (p->n_steps - 1) : KWS_E_INVALID; }

static int stretch_launch(const kws_stretch_plan_t* p, const void* x, bool i16, float in_scale, float* out, int B,
                          int keep, int quantize, void* stream) {
  KWS_REQUIRE(p && x && out, "time_stretch: NULL pointer");
  KWS_REQUIRE(B >= 0 && keep > 0, "time_stretch: B=%d keep=%d", B, keep);
  if (B == 0) return KWS_OK;
  StretchArgs a;
  a.x = x; a.out = out; a.tw = p->tw; a.tws = p->tws; a.win = p->win; a.sidx = p->sidx; a.salpha = p->salpha;
  a.B = B; a.L = p->n_samples; a.n_frames = p->n_frames; a.T = p->n_steps; a.keep = keep; a.quantize = quantize;
  a.in_scale = in_scale;
  hipStream_t st = (hipStream_t)stream;
  if (i16)
    hipLaunchKernelGGL(stretch_kernel<int16_t>, dim3(B), dim3(NT), 0, st, a);
  else
    hipLaunchKernelGGL(stretch_kernel<float>, dim3(B), dim3(NT), 0, st, a);
  KWS_LAUNCH_CHECK("stretch_kernel");
  return KWS_OK;
}

int kws_time_stretch_f32(const kws_stretch_plan_t* plan, const float* x, float in_scale, float* out, int B, int keep,
                         int quantize, void* stream) {
  return stretch_launch(plan, x, false, in_scale, out, B, keep, quantize, stream);
}

int kws_time_stretch_i16(const kws_stretch_plan_t* plan, const int16_t* x, float* out, int B, int keep, int quantize,
                         void* stream) {
  return stretch_launch(plan, x, true, 32767.f, out, B, keep, quantize, stream);
}

}  // extern "C"
