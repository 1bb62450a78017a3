// Batched STFT -> |X| -> mel -> log -> DCT feature kernel.
//
// One table-driven kernel serves both reference feature paths:
//   path B  input_data.py:361-381  stft(480,160,fft 512) -> abs -> mel matmul -> log(+1e-6) -> DCT-II[:K]
//   path A  audio.py:15-23         AudioSpectrogram(squared) -> Mfcc (sqrt, filterbank, log floor, DCT)
// Design (gfx950): a workgroup of NW waves owns a run of NW*FPW frames of ONE clip.  The PCM run is
// read from HBM once with coalesced loads into LDS (each sample is reused by its 3 overlapping
// frames from LDS, never re-read from HBM); every wave then transforms its frames one at a time:
// the 512-point real FFT is a 256-point complex radix-4 Stockham Cooley-Tukey (4 stages, one
// butterfly per lane per stage, ping-pong in LDS) plus the real-input split step; magnitudes, the
// sparse (CSR) triangular mel bands, log and the small dense DCT all stay in LDS - only the
// [F, n_out] features go back to HBM.
#include "internal.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int NFFT = 512;
constexpr int NC = 256;  // complex points

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

struct StftArgs {
  kws_stft_plan pl;
  const float* x;
  float* out;
  int L, F, out_kind;
  int run_samples;  // samples staged per workgroup
};

template <int NW, int FPW>
__global__ __launch_bounds__(NW * 64) void stft_kernel(StftArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const kws_stft_plan& pl = a.pl;
  const int n_mel = pl.n_mel, n_out = pl.n_out;
  // LDS carve (floats)
  float* s_x = lds;                                  // [run_samples]
  float* s_win = s_x + ((a.run_samples + 3) & ~3);   // [512]
  float2* s_w256 = reinterpret_cast<float2*>(s_win + NFFT);       // [256]
  float2* s_w512 = s_w256 + NC;                                   // [258]
  float* s_dct = reinterpret_cast<float*>(s_w512 + 258);          // [n_mel*n_out]
  float* s_bw = s_dct + ((n_mel * n_out + 3) & ~3);               // [n_w]
  float* s_wave = s_bw + ((pl.n_w + 3) & ~3);                     // per-wave regions
  constexpr int WAVE_FLOATS = 2 * 2 * NC + 260 + 128;             // ping, pong, mag, logmel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float2* bufA = reinterpret_cast<float2*>(s_wave + wave * WAVE_FLOATS);
  float2* bufB = bufA + NC;
  float* s_mag = reinterpret_cast<float*>(bufB + NC);             // [260]
  float* s_lm = s_mag + 260;                                      // [128]

  const int b = blockIdx.y;
  const int f_base = blockIdx.x * (NW * FPW);
  const float* xb = a.x + (int64_t)b * a.L;
  const int s0 = f_base * pl.frame_step;
  // ---- stage the PCM run + tables ------------------------------------------------------------
  for (int i = tid; i < a.run_samples; i += NW * 64) {
    const int s = s0 + i;
    s_x[i] = (s < a.L) ? xb[s] : 0.f;
  }
  for (int i = tid; i < NFFT; i += NW * 64) s_win[i] = pl.window[i];
  for (int i = tid; i < NC; i += NW * 64) s_w256[i] = pl.w256[i];
  for (int i = tid; i < 257; i += NW * 64) s_w512[i] = pl.w512[i];
  for (int i = tid; i < n_mel * n_out; i += NW * 64) s_dct[i] = pl.dct[i];
  for (int i = tid; i < pl.n_w; i += NW * 64) s_bw[i] = pl.band_w[i];
  __syncthreads();

  for (int fi = 0; fi < FPW; ++fi) {
    const int f = f_base + wave * FPW + fi;
    if (f >= a.F) break;  // wave-uniform
    const float* fx = s_x + (f - f_base) * pl.frame_step;
    // ---- windowed frame packed as 256 complex points z[n] = w x[2n] + i w x[2n+1] -----------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = lane + 64 * r;
      float2 z = make_float2(0.f, 0.f);
      if (2 * n + 1 < pl.frame_len) {
        const float2 xv = *reinterpret_cast<const float2*>(fx + 2 * n);  // frame_step, f_base even -> 8-B aligned
        const float2 wv = *reinterpret_cast<const float2*>(s_win + 2 * n);
        z = make_float2(xv.x * wv.x, xv.y * wv.y);
      } else if (2 * n < pl.frame_len) {
        z.x = fx[2 * n] * s_win[2 * n];
      }
      bufA[n] = z;
    }
    // ---- radix-4 Stockham, p = 1, 4, 16, 64 ------------------------------------------------
    float2* src = bufA;
    float2* dst = bufB;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int p = 1 << (2 * st);
      const int k = lane & (p - 1);
      const int j = ((lane - k) << 2) + k;
      const int step = 64 >> (2 * st);
      const float2 u0 = src[lane];
      float2 u1 = src[lane + 64], u2 = src[lane + 128], u3 = src[lane + 192];
      if (st > 0) {
        u1 = cmul(u1, s_w256[k * step]);
        u2 = cmul(u2, s_w256[2 * k * step]);
        u3 = cmul(u3, s_w256[3 * k * step]);
      }
      const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3);
      const float2 d = csub(u1, u3);
      const float2 v3 = make_float2(d.y, -d.x);  // (u1-u3) * -i
      dst[j] = cadd(v0, v2);
      dst[j + p] = cadd(v1, v3);
      dst[j + 2 * p] = csub(v0, v2);
      dst[j + 3 * p] = csub(v1, v3);
      float2* t = src;
      src = dst;
      dst = t;
      // all 64 lanes of this wave must see the stage's stores: same-wave LDS ops are ordered, but
      // the compiler must not reorder across: use a wave-level barrier/fence
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const float2* Z = src;  // after 4 swaps: src == bufA
    // ---- real-input split: X[k] = E + W512^k O, magnitude -----------------------------------
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int k = lane + 64 * r;
      if (k <= NC) {
        const float2 zk = Z[k & (NC - 1)];
        const float2 zn0 = Z[(NC - k) & (NC - 1)];
        const float2 zn = make_float2(zn0.x, -zn0.y);
        const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
        const float2 dd = csub(zk, zn);
        const float2 O = make_float2(0.5f * dd.y, -0.5f * dd.x);  // -0.5 i (zk - zn)
        const float2 X = cadd(E, cmul(s_w512[k], O));
        s_mag[k] = sqrtf(X.x * X.x + X.y * X.y);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (a.out_kind == 1) {
      float* o = a.out + ((int64_t)b * a.F + f) * 257;
      for (int k = lane; k < 257; k += 64) o[k] = s_mag[k];
      continue;
    }
    // ---- sparse mel bands + log -------------------------------------------------------------
    for (int m = lane; m < n_mel; m += 64) {
      const int st0 = pl.band_start[m], cnt = pl.band_cnt[m], ofs = pl.band_ofs[m];
      float s = 0.f;
      for (int i = 0; i < cnt; ++i) s = fmaf(s_mag[st0 + i], s_bw[ofs + i], s);
      s += pl.log_offset;
      if (pl.log_floor > 0.f) s = fmaxf(s, pl.log_floor);
      s_lm[m] = logf(s);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (a.out_kind == 2) {
      float* o = a.out + ((int64_t)b * a.F + f) * n_mel;
      for (int m = lane; m < n_mel; m += 64) o[m] = s_lm[m];
      continue;
    }
    // ---- DCT (dense, from LDS) ---------------------------------------------------------------
    float* o = a.out + ((int64_t)b * a.F + f) * n_out;
    for (int q = lane; q < n_out; q += 64) {
      float s = 0.f;
      for (int m = 0; m < n_mel; ++m) s = fmaf(s_lm[m], s_dct[m * n_out + q], s);
      o[q] = s;
    }
    // next frame reuses bufA/s_mag/s_lm: order this frame's reads before the next frame's writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int NW, int FPW>
int launch_stft(const StftArgs& a, int B, hipStream_t st) {
  const kws_stft_plan& pl = a.pl;
  constexpr int WAVE_FLOATS = 2 * 2 * NC + 260 + 128;
  const size_t floats = ((a.run_samples + 3) & ~3) + NFFT + 2 * NC + 2 * 258 + ((pl.n_mel * pl.n_out + 3) & ~3) +
                        ((pl.n_w + 3) & ~3) + (size_t)NW * WAVE_FLOATS;
  const size_t bytes = floats * 4;
  KWS_REQUIRE(bytes <= 160 * 1024, "stft: LDS need %zu B exceeds 160 KiB", bytes);
  KWS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_kernel<NW, FPW>),   // per device, cheap: every launch
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  dim3 g((unsigned)ceil_div(a.F, NW * FPW), (unsigned)B), blk(NW * 64);
  hipLaunchKernelGGL((stft_kernel<NW, FPW>), g, blk, bytes, st, a);
  KWS_LAUNCH_CHECK("stft_kernel");
  return KWS_OK;
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  KWS_HIP(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(T) + 16));
  KWS_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return KWS_OK;
}

}  // namespace

extern "C" {

int kws_stft_plan_create(int frame_len, int frame_step, int fft_len, int n_mel, int n_out, const float* window,
                         const float* mel, const float* dct, float log_offset, float log_floor,
                         kws_stft_plan_t** plan) {
  KWS_REQUIRE(plan && window && mel && dct, "stft_plan_create: NULL pointer");
  KWS_REQUIRE(fft_len == NFFT, "stft_plan_create: fft_len %d unsupported (512 only)", fft_len);
  KWS_REQUIRE(frame_len > 0 && frame_len <= NFFT && frame_step > 0 && frame_step % 2 == 0,
              "stft_plan_create: frame_len=%d frame_step=%d (need <=512, even step)", frame_len, frame_step);
  KWS_REQUIRE(n_mel > 0 && n_mel <= 128 && n_out > 0 && n_out <= n_mel + 128, "stft_plan_create: n_mel=%d n_out=%d",
              n_mel, n_out);
  kws_stft_plan* p = new kws_stft_plan();
  p->frame_len = frame_len; p->frame_step = frame_step; p->fft_len = fft_len; p->n_bins = fft_len / 2 + 1;
  p->n_mel = n_mel; p->n_out = n_out; p->log_offset = log_offset; p->log_floor = log_floor;
  std::vector<float> win(NFFT, 0.f);
  for (int i = 0; i < frame_len; ++i) win[i] = window[i];
  std::vector<float2> w256(NC), w512(257);
  for (int j = 0; j < NC; ++j) {
    const double ang = -2.0 * M_PI * j / 256.0;
    w256[j] = make_float2((float)cos(ang), (float)sin(ang));
  }
  for (int k = 0; k <= 256; ++k) {
    const double ang = -2.0 * M_PI * k / 512.0;
    w512[k] = make_float2((float)cos(ang), (float)sin(ang));
  }
  // CSR bands: contiguous non-zero bin range of every mel column
  std::vector<int> bs(n_mel), bc(n_mel), bo(n_mel);
  std::vector<float> bw;
  for (int m = 0; m < n_mel; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < p->n_bins; ++k)
      if (mel[(size_t)k * n_mel + m] != 0.f) {
        if (lo < 0) lo = k;
        hi = k;
      }
    bs[m] = lo < 0 ? 0 : lo;
    bc[m] = lo < 0 ? 0 : hi - lo + 1;
    bo[m] = (int)bw.size();
    for (int k = 0; k < bc[m]; ++k) bw.push_back(mel[(size_t)(bs[m] + k) * n_mel + m]);
  }
  if (bw.empty()) bw.push_back(0.f);
  p->n_w = (int)bw.size();
  std::vector<float> d(dct, dct + (size_t)n_mel * n_out);
  // stft4 tables: the DCT matrix padded to 64 columns
  std::vector<float> d64((size_t)n_mel * 64, 0.f);
  for (int m = 0; m < n_mel; ++m)
    for (int q = 0; q < n_out && q < 64; ++q) d64[(size_t)m * 64 + q] = dct[(size_t)m * n_out + q];
  // v4 kernel tables (stft4.hip; index algebra replayed in scripts/emulate_stft4.py)
  static const int KPERM[16] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 8};
  std::vector<float> b4(64 * 16);
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 8; ++j)
      for (int ct = 0; ct < 2; ++ct) {
        const int s4 = lane >> 4, c = lane & 15, jp = j >> 1, part = j & 1;
        const int n1 = 4 * jp + s4;
        // W16^(n1 k1) needs only the angle mod 2 pi: reduce the integer first so cos / sin see an exact multiple of pi/8
        const int e = (n1 * KPERM[c]) & 15;
        const double th = 2.0 * M_PI * e / 16.0;
        double v;
        if (ct == 0) v = part == 0 ? cos(th) : sin(th);
        else v = part == 0 ? -sin(th) : cos(th);
        b4[lane * 16 + j * 2 + ct] = (float)(0.5 * v);
      }
  std::vector<float2> tw4(256), w512p(128);
  for (int c = 0; c < 16; ++c) {
    for (int n2 = 0; n2 < 16; ++n2) {
      const double ang = -2.0 * M_PI * (n2 * KPERM[c]) / 256.0;
      tw4[c * 16 + n2] = make_float2((float)cos(ang), (float)sin(ang));
    }
    for (int k2 = 0; k2 < 8; ++k2) {
      const double ang = -2.0 * M_PI * (KPERM[c] + 16 * k2) / 512.0;
      w512p[c * 8 + k2] = make_float2((float)cos(ang), (float)sin(ang));
    }
  }
  // mel tap windows of the v4 kernel: band m reads mel_maxw taps from bin ws[m] (weights zero outside the band)
  std::vector<int> mws(n_mel, 0);
  std::vector<float> wpad(1, 0.f);
  {
    const int MAGF4 = 260;                                   // floats of a magnitude row in the kernel's LDS
    int mcmax = 0;
    for (int i = 0; i < 8; ++i) p->mel_mc[i] = 0;
    // five lane groups of bands (80 mel bins): the windows start on EVEN bins and the kernel reads its magnitudes as 8-byte
    // pairs - half the LDS instructions and fewer bank conflicts (16 band windows that start 3 - 8 bins apart took 272 LDS
    // cycles per quad where 88 are conflict-free): 50.1 -> 48.3 us per 1024 clips.  Three lane groups (40 bins) keep 4-byte
    // reads: the alignment costs them a fifth block in group 0 and measured 93 -> 96 us.
    const int al_mask = (n_mel + 15) / 16 >= 5 ? 1 : 0;
    for (int m = 0; m < n_mel; ++m) {
      const int g = m >> 4, need = (bc[m] + (bs[m] & al_mask) + 3) / 4;
      if (need > p->mel_mc[g]) p->mel_mc[g] = need;
    }
    for (int i = 0; i < 8; ++i) {
      if (i < (n_mel + 15) / 16 && p->mel_mc[i] == 0) p->mel_mc[i] = 1;
      if (p->mel_mc[i] > mcmax) mcmax = p->mel_mc[i];
    }
    p->mel_maxw = 4 * mcmax;
    if (p->mel_maxw > 64) {
      p->mel_maxw = 0;                                       // stft4 declines (kws_stft4_lds_bytes / launch check it)
    } else {
      wpad.assign((size_t)n_mel * p->mel_maxw, 0.f);
      for (int m = 0; m < n_mel; ++m) {
        const int taps = p->mel_maxw;                        // one window width for every band
        int ws0 = bs[m] & ~al_mask;
        if (ws0 + taps > MAGF4) ws0 = MAGF4 - taps;          // keep the window inside the row: the band sits later in it (even)
        mws[m] = ws0;
        for (int k = 0; k < bc[m]; ++k) wpad[(size_t)m * p->mel_maxw + (bs[m] - ws0) + k] = bw[bo[m] + k];
      }
    }
  }
  int rc = upload(&p->window, win);
  if (rc == KWS_OK) rc = upload(&p->w256, w256);
  if (rc == KWS_OK) rc = upload(&p->w512, w512);
  if (rc == KWS_OK) rc = upload(&p->band_start, bs);
  if (rc == KWS_OK) rc = upload(&p->band_cnt, bc);
  if (rc == KWS_OK) rc = upload(&p->band_ofs, bo);
  if (rc == KWS_OK) rc = upload(&p->band_w, bw);
  if (rc == KWS_OK) rc = upload(&p->dct, d);
  if (rc == KWS_OK) rc = upload(&p->dct64, d64);
  if (rc == KWS_OK) rc = upload(&p->b4, b4);
  if (rc == KWS_OK) rc = upload(&p->tw4, tw4);
  if (rc == KWS_OK) rc = upload(&p->w512p, w512p);
  if (rc == KWS_OK) rc = upload(&p->mel_ws, mws);
  if (rc == KWS_OK) rc = upload(&p->mel_wpad, wpad);
  if (rc == KWS_OK) rc = kws_stft4_prepare(p);
  if (rc != KWS_OK) {
    kws_stft_plan_destroy(p);
    return rc;
  }
  *plan = p;
  return KWS_OK;
}

int kws_stft_plan_destroy(kws_stft_plan_t* p) {
  if (!p) return KWS_OK;
  void* bufs[15] = {p->window, p->w256, p->w512, p->band_start, p->band_cnt, p->band_ofs, p->band_w, p->dct,
                    p->dct64, p->b4, p->tw4, p->w512p, p->mel_ws, p->mel_wpad, p->img4};
  for (void* q : bufs)
    if (q) (void)hipFree(q);
  delete p;
  return KWS_OK;
}

int kws_stft_num_frames(const kws_stft_plan_t* p, int L) {
  if (!p || L < p->frame_len) return 0;
  return 1 + (L - p->frame_len) / p->frame_step;
}

int kws_stft_mel_f32(const kws_stft_plan_t* plan, const float* x, int B, int L, float* out, int out_kind,
                     void* stream) {
  KWS_REQUIRE(plan && x && out, "stft_mel: NULL pointer");
  KWS_REQUIRE(B > 0 && B <= 65535 && L >= plan->frame_len && L % 2 == 0, "stft_mel: B=%d L=%d", B, L);
  KWS_REQUIRE(out_kind >= 0 && out_kind <= 2, "stft_mel: out_kind %d", out_kind);
  StftArgs a;
  a.pl = *plan;
  a.x = x; a.out = out; a.L = L; a.out_kind = out_kind;
  a.F = kws_stft_num_frames(plan, L);
  hipStream_t st = (hipStream_t)stream;
  const int width = out_kind == 0 ? plan->n_out : (out_kind == 1 ? 257 : plan->n_mel);
  // ~5 N log2 N for the 256-point complex FFT + split + sparse mel + dense DCT, per frame
  const double fl = (double)B * a.F * (5.0 * 256 * 8 + 12.0 * 257 + 2.0 * plan->n_w + 2.0 * plan->n_mel * plan->n_out);
  KwsProfScope prof("stft_mel", fl, 4.0 * ((double)B * L + (double)B * a.F * width), st);
  // the feature form the reference's settings produce (train.py:38, settings.py:3, audio.py:20-23) runs stft4_kernel (first
  // radix-16 pass + DCT on the matrix pipe); any other table shape or output kind takes the generic kernel below
  if (out_kind == 0 && plan->n_out <= 64 && plan->n_mel <= 128 && plan->n_mel % 4 == 0 && plan->frame_len % 2 == 0 &&
      plan->mel_maxw > 0 && plan->img4 != nullptr && kws_stft4_lds_bytes(plan) <= 160 * 1024)
    return kws_stft4_launch(plan, x, B, L, a.F, out, st);
  if (a.F % 14 == 0) {
    a.run_samples = (7 * 2 - 1) * plan->frame_step + plan->frame_len;
    return launch_stft<7, 2>(a, B, st);
  }
  a.run_samples = (4 * 2 - 1) * plan->frame_step + plan->frame_len;
  return launch_stft<4, 2>(a, B, st);
}

}  // extern "C"
