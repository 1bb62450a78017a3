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

// Passes of the 1024-point Stockham FFT.  `twr[s][r]` = twiddles of pass s+1 (Ns = 4^(s+1)).
// FIRST_REGS: pass 0 takes its operands from u (else from src); LAST_REGS: pass 4 leaves its results in u
// (else writes dst).  Returns with the result in `u` or in the buffer that was written last; buffers
// alternate a -> b -> a ...  One barrier after every LDS write.
template <bool INV, bool FIRST_REGS, bool LAST_REGS>
__device__ __forceinline__ float2* fft1024(float2 (&u)[4], float2* a, float2* b, const float2 (&twr)[4][3], int j) {
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
      for (int r = 1; r < 4; ++r) u[r] = INV ? cmulc(u[r], twr[s - 1][r - 1]) : cmul(u[r], twr[s - 1][r - 1]);
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

__device__ __forceinline__ float2 unit_or_one(float2 c, float m) {
  return m > 0.f ? make_float2(c.x / m, c.y / m) : make_float2(1.f, 0.f);
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

// Bin ownership of thread j: item 0 -> (j, 1024-j) [j = 0: (0, 1024), both real], item 1 -> (j+256, 768-j),
// item 2 (thread 0 only) -> 512 (its own mirror).  Slot 2*i holds bin p, slot 2*i+1 the mirror.
constexpr int NSLOT = 6;

template <typename TIn>
__global__ __launch_bounds__(NT) void stretch_kernel(StretchArgs a) {
  __shared__ float2 bufA[MC];
  __shared__ float2 bufB[MC];
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
  float2 twr[4][3];
#pragma unroll
  for (int s = 1; s < 5; ++s) {
    const int Ns = 1 << (2 * s);
    const int k = j & (Ns - 1);
#pragma unroll
    for (int r = 1; r < 4; ++r) twr[s - 1][r - 1] = a.tw[r * k * (NT / Ns)];
  }
  const float2 ws0 = a.tws[j], ws1 = a.tws[j + NT], ws2 = a.tws[512];

  float2 Ca[NSLOT], Cb[NSLOT], U[NSLOT];
  float ola[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) ola[r][0] = ola[r][1] = 0.f;

  // ---- analysis: STFT column f into C (zeros beyond the last frame: the two padded columns) ----
  auto analyse = [&](int f, float2 (&C)[NSLOT]) {
    if (f >= a.n_frames) {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) C[i] = make_float2(0.f, 0.f);
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
    const float2* Z = fft1024<false, true, false>(u, bufA, bufB, twr, j);
    // split: X[k] = E + T, X[1024-k] = conj(E - T), E = (Z[k] + conj Z[M-k]) / 2, T = -j W^k (Z[k] - conj Z[M-k]) / 2
    auto split = [&](int k, float2 wk, float2& Xk, float2& Xm) {
      const float2 za = Z[k], zb = cconj(Z[(MC - k) & (MC - 1)]);
      const float2 E = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y + zb.y));
      const float2 D = make_float2(0.5f * (za.x - zb.x), 0.5f * (za.y - zb.y));
      const float2 wd = cmul(wk, D);
      const float2 T = make_float2(wd.y, -wd.x);  // -j * wd
      Xk = cadd(E, T);
      Xm = cconj(csub(E, T));
    };
    if (j == 0) {
      const float2 z0 = Z[0];
      C[0] = make_float2(z0.x + z0.y, 0.f);
      C[1] = make_float2(z0.x - z0.y, 0.f);
      float2 dummy;
      split(512, ws2, C[4], dummy);
    } else {
      split(j, ws0, C[0], C[1]);
      C[4] = C[5] = make_float2(0.f, 0.f);
    }
    split(j + NT, ws1, C[2], C[3]);
    C[5] = make_float2(0.f, 0.f);
    __syncthreads();  // Z fully consumed before the buffers are reused
  };

  analyse(0, Ca);
  analyse(1, Cb);
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) U[i] = unit_or_one(Ca[i], sqrtf(Ca[i].x * Ca[i].x + Ca[i].y * Ca[i].y));
  int cur = 0;  // Ca = column cur, Cb = column cur + 1

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
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) Ca[i] = Cb[i];
      analyse(cur + 2, Cb);
      ++cur;
    }
    // ---- phase vocoder step on the owned bins ----
    float2 Y[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const float m0 = sqrtf(Ca[i].x * Ca[i].x + Ca[i].y * Ca[i].y);
      const float m1 = sqrtf(Cb[i].x * Cb[i].x + Cb[i].y * Cb[i].y);
      const float mag = (1.f - al) * m0 + al * m1;
      Y[i] = make_float2(mag * U[i].x, mag * U[i].y);
      float2 un = cmulc(cmul(U[i], unit_or_one(Cb[i], m1)), unit_or_one(Ca[i], m0));
      const float n = sqrtf(un.x * un.x + un.y * un.y);
      U[i] = unit_or_one(un, n);
    }
    // ---- inverse pack: Z'[k] = E + jO, Z'[M-k] = conj(E - jO), E = (X_k + conj X_m)/2, O = conj(W^k) (X_k - conj X_m)/2
    auto pack = [&](int k, float2 wk, float2 Xk, float2 Xm, bool mirror) {
      const float2 xm = cconj(Xm);
      const float2 E = make_float2(0.5f * (Xk.x + xm.x), 0.5f * (Xk.y + xm.y));
      const float2 T = make_float2(0.5f * (Xk.x - xm.x), 0.5f * (Xk.y - xm.y));
      const float2 O = cmulc(T, wk);
      const float2 jO = make_float2(-O.y, O.x);
      bufA[k] = cadd(E, jO);
      if (mirror) bufA[MC - k] = cconj(csub(E, jO));
    };
    if (j == 0) {
      bufA[0] = make_float2(0.5f * (Y[0].x + Y[1].x), 0.5f * (Y[0].x - Y[1].x));  // irfft ignores Im of DC / Nyquist
      pack(512, ws2, Y[4], Y[4], false);
    } else {
      pack(j, ws0, Y[0], Y[1], true);
    }
    pack(j + NT, ws1, Y[2], Y[3], true);
    __syncthreads();
    float2 u[4];
    fft1024<true, false, true>(u, bufA, bufB, twr, j);
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
