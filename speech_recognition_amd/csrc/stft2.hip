// STFT -> |X| -> mel -> log -> DCT feature kernel, second generation.
//
// What the first kernel (stft.hip) got wrong for this chip, from its profile (0.46 ms / 1024 clips =
// 2.4 % of the HBM roof): one radix-4 butterfly per lane per stage means four LDS round trips per frame
// with a dependent wait each, one wave = one frame means 64 lanes share a 256-point transform, and every
// workgroup re-staged 25 KB of tables for 14 frames of work.
//
// Here a frame is owned by 16 lanes (one DPP row, four frames per wave) and the 256-point complex FFT of
// the even/odd-packed 512-sample frame is the 16 x 16 Cooley-Tukey split done in REGISTERS:
//     n = 16 n1 + n2,  k = k1 + 16 k2
//     lane n2: 16-point FFT over n1  ->  x W256^(n2 k1)  ->  [one 16x16 transpose through LDS]
//     lane k1: 16-point FFT over n2  ->  Z[k1 + 16 k2] in register k2
// so a frame crosses LDS once instead of four times.  The real-input split needs Z[256-k], which lives in
// lane 16-k1, register 15-k2: one row-local shuffle per register.  Magnitudes go to LDS once; every lane
// then gathers 5-6 triangular mel bands (CSR), takes the log, and produces 4 DCT outputs per lane from a
// 16-byte-wide read of the zero-padded [n_mel][64] DCT table, so one frame's feature row leaves as one
// contiguous 16-lane x 16-byte store.  Waves take frame quads grid-stride over the WHOLE batch and read
// their PCM straight from global memory (128-byte coalesced per 16-lane row; the 3x frame overlap is
// served by L2), so there is no per-clip staging, no workgroup barrier after the table load, and the
// tables are loaded once per persistent workgroup.
#include "stft_common.h"

namespace {

constexpr int NW2 = 12;            // waves per workgroup (3 per SIMD; 134 KB LDS incl. tables)
constexpr int TP = 17;             // padded row (complex) of the per-frame 16x16 transpose tile

using namespace kws_fft;

__global__ __launch_bounds__(NW2 * 64, 1) void stft2_kernel(Stft2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const kws_stft_plan& pl = a.pl;
  const int n_mel = pl.n_mel, n_out = pl.n_out;
  // ---- LDS carve (floats) --------------------------------------------------------------------------
  float* s_win = lds;                                              // [512]
  float2* s_tw = reinterpret_cast<float2*>(s_win + 512);           // [256]  W256^(n2 k1), index n2*16+k1
  float2* s_w512 = s_tw + 256;                                     // [258]
  float* s_dct = reinterpret_cast<float*>(s_w512 + 258);           // [n_mel][64]
  float* s_bw = s_dct + n_mel * 64;                                // [n_w]
  int* s_csr = reinterpret_cast<int*>(s_bw + ((pl.n_w + 3) & ~3)); // [3][128] start | cnt | ofs
  float* s_wave = reinterpret_cast<float*>(s_csr + 3 * 128);       // per-wave scratch
  constexpr int FR_FLOATS = 16 * TP * 2;                           // one frame's transpose tile (544 floats)
  constexpr int WAVE_FLOATS = 4 * FR_FLOATS;                       // also reused for mag[260] + logmel[128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, fq = lane >> 4;                       // lane in frame, frame in quad
  float* s_fr = s_wave + wave * WAVE_FLOATS + fq * FR_FLOATS;      // this frame's scratch

  for (int i = tid; i < 512; i += NW2 * 64) s_win[i] = pl.window[i];
  for (int i = tid; i < 256; i += NW2 * 64) s_tw[i] = pl.tw16[i];
  for (int i = tid; i < 257; i += NW2 * 64) s_w512[i] = pl.w512[i];
  for (int i = tid; i < n_mel * 64; i += NW2 * 64) s_dct[i] = pl.dct64[i];
  for (int i = tid; i < pl.n_w; i += NW2 * 64) s_bw[i] = pl.band_w[i];
  for (int i = tid; i < 128; i += NW2 * 64) {
    s_csr[i] = i < n_mel ? pl.band_start[i] : 0;
    s_csr[128 + i] = i < n_mel ? pl.band_cnt[i] : 0;
    s_csr[256 + i] = i < n_mel ? pl.band_ofs[i] : 0;
  }
  __syncthreads();

  const int64_t wave_global = (int64_t)blockIdx.x * NW2 + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * NW2;
  const int nb = (n_mel + 15) >> 4;   // mel bands per lane
  for (int64_t quad = wave_global; quad < a.total_quads; quad += wave_stride) {
    const int64_t b = quad / a.quads_per_clip;
    const int f = (int)(quad - b * a.quads_per_clip) * 4 + fq;
    const bool live = f < a.F;                      // 16-lane uniform
    const float* fx = a.x + b * (int64_t)a.L + (int64_t)(live ? f : 0) * pl.frame_step;
    // ---- load: lane n2 = l16 takes z[16 n1 + n2] = w x[2n] + i w x[2n+1], n1 = 0..15 ----------------
    float2 z[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      // Unconditional loads (a branch per load would serialise them behind vmcnt(0) waits): positions
      // past the frame read a clamped in-frame address and are zeroed by the zero-padded window.
      const int s = 2 * (16 * n1 + l16);
      const int sc = s < pl.frame_len ? s : pl.frame_len - 2;   // frame_len is even (checked on the host)
      const float2 xv = *reinterpret_cast<const float2*>(fx + sc);
      const float2 wv = *reinterpret_cast<const float2*>(s_win + s);
      z[n1] = make_float2(xv.x * wv.x, xv.y * wv.y);
    }
    fft16(z);                                       // over n1: z[k1] = Y[n2][k1]
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], s_tw[l16 * 16 + k1]);
    // ---- 16x16 transpose through LDS: write [k1][n2], read row k1 = l16 ------------------------------
    float2* tile = reinterpret_cast<float2*>(s_fr);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) tile[k1 * TP + l16] = z[k1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) z[n2] = tile[l16 * TP + n2];
    fft16(z);                                       // over n2: z[k2] = Z[k1 + 16 k2], k1 = l16
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                // tile reads done before the scratch is reused
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- real-input split + magnitude: X[k] = E + W512^k O, partner = Z[256-k] ----------------------
    float* s_mag = s_fr;                            // [260]
    const int partner = (16 - l16) & 15;            // lane holding bins 256-k (k1 > 0); lane 0 pairs with itself
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      // partner register: 15-k2 for k1 > 0, (16-k2)&15 for k1 = 0 (static indices, selected per lane)
      const float2 pa = z[15 - k2], pb = z[(16 - k2) & 15];
      const float px = __shfl(pa.x, partner, 16), py = __shfl(pa.y, partner, 16);
      const float2 zn0 = (l16 == 0) ? pb : make_float2(px, py);
      const float2 zk = z[k2];
      const float2 zn = make_float2(zn0.x, -zn0.y);
      const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
      const float2 dd = csub(zk, zn);
      const float2 O = make_float2(0.5f * dd.y, -0.5f * dd.x);
      const int k = l16 + 16 * k2;
      const float2 X = cadd(E, cmul(s_w512[k], O));
      s_mag[k] = __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y);
    }
    if (l16 == 0) s_mag[256] = fabsf(z[0].x - z[0].y);   // Nyquist bin: Re Z0 - Im Z0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- sparse mel bands + log: lane takes bands l16, l16+16, ... -----------------------------------
    float* s_lm = s_fr + 264;                       // [128]
    for (int i = 0; i < nb; ++i) {
      const int m = l16 + 16 * i;
      if (m < n_mel) {
        const int st0 = s_csr[m], cnt = s_csr[128 + m], ofs = s_csr[256 + m];
        // 4 independent (magnitude, weight) pairs per trip: the LDS latency is paid once per 4 taps
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int j = 0; j < cnt; j += 4) {
          const int j1 = j + 1 < cnt ? j + 1 : j, j2 = j + 2 < cnt ? j + 2 : j, j3 = j + 3 < cnt ? j + 3 : j;
          const float m0 = s_mag[st0 + j], m1 = s_mag[st0 + j1], m2 = s_mag[st0 + j2], m3 = s_mag[st0 + j3];
          const float w0 = s_bw[ofs + j];
          const float w1 = j + 1 < cnt ? s_bw[ofs + j1] : 0.f;
          const float w2 = j + 2 < cnt ? s_bw[ofs + j2] : 0.f;
          const float w3 = j + 3 < cnt ? s_bw[ofs + j3] : 0.f;
          s0 = fmaf(m0, w0, s0);
          s1 = fmaf(m1, w1, s1);
          s2 = fmaf(m2, w2, s2);
          s3 = fmaf(m3, w3, s3);
        }
        float sm = ((s0 + s1) + (s2 + s3)) + pl.log_offset;
        if (pl.log_floor > 0.f) sm = fmaxf(sm, pl.log_floor);
        s_lm[m] = __logf(sm);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- DCT: 4 outputs per lane (q = 4 l16 .. +3) ----------------------------------------------------
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f), o2 = o;
    {
      // two independent accumulator sets, 4 mel bins per LDS read of the log-mel row
      const float* dcol = s_dct + 4 * l16;
      int m = 0;
#pragma unroll 2
      for (; m + 3 < n_mel; m += 4) {
        const float4 lv = *reinterpret_cast<const float4*>(s_lm + m);
        const float4 d0 = *reinterpret_cast<const float4*>(dcol + (m + 0) * 64);
        const float4 d1 = *reinterpret_cast<const float4*>(dcol + (m + 1) * 64);
        const float4 d2 = *reinterpret_cast<const float4*>(dcol + (m + 2) * 64);
        const float4 d3 = *reinterpret_cast<const float4*>(dcol + (m + 3) * 64);
        o.x = fmaf(lv.x, d0.x, o.x); o.y = fmaf(lv.x, d0.y, o.y); o.z = fmaf(lv.x, d0.z, o.z); o.w = fmaf(lv.x, d0.w, o.w);
        o2.x = fmaf(lv.y, d1.x, o2.x); o2.y = fmaf(lv.y, d1.y, o2.y); o2.z = fmaf(lv.y, d1.z, o2.z); o2.w = fmaf(lv.y, d1.w, o2.w);
        o.x = fmaf(lv.z, d2.x, o.x); o.y = fmaf(lv.z, d2.y, o.y); o.z = fmaf(lv.z, d2.z, o.z); o.w = fmaf(lv.z, d2.w, o.w);
        o2.x = fmaf(lv.w, d3.x, o2.x); o2.y = fmaf(lv.w, d3.y, o2.y); o2.z = fmaf(lv.w, d3.z, o2.z); o2.w = fmaf(lv.w, d3.w, o2.w);
      }
      for (; m < n_mel; ++m) {
        const float lv = s_lm[m];
        const float4 d = *reinterpret_cast<const float4*>(dcol + m * 64);
        o.x = fmaf(lv, d.x, o.x); o.y = fmaf(lv, d.y, o.y); o.z = fmaf(lv, d.z, o.z); o.w = fmaf(lv, d.w, o.w);
      }
      o.x += o2.x; o.y += o2.y; o.z += o2.z; o.w += o2.w;
    }
    if (live) {
      float* orow = a.out + (b * a.F + f) * (int64_t)n_out;
      const int q0 = 4 * l16;
      if ((n_out & 3) == 0 && q0 + 3 < n_out) {
        *reinterpret_cast<float4*>(orow + q0) = o;
      } else {
        if (q0 < n_out) orow[q0] = o.x;
        if (q0 + 1 < n_out) orow[q0 + 1] = o.y;
        if (q0 + 2 < n_out) orow[q0 + 2] = o.z;
        if (q0 + 3 < n_out) orow[q0 + 3] = o.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                // scratch reads done before the next quad's writes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}


// ------------------------------------------------------------------------------------------------
// Third generation.  The second kernel is LDS-bandwidth bound (about 290 LDS instructions per lane and
// frame, 100 of them the 16-byte reads of the DCT table).  Here
//   * the per-lane constants of the FFT (window, W256 twiddles, W512 split factors: 94 registers) live in
//     registers - 8 waves per workgroup leave 256 VGPRs per lane;
//   * a wave takes FOUR quads (16 frames) per pass and parks their log-mel rows in LDS; the DCT of the 16
//     frames is then one [16 x n_mel] x [n_mel x 64] product on the matrix pipe (v_mfma_f32_16x16x4_f32,
//     exact f32 fma chain): 5 four-byte LDS reads per 4 MFMAs instead of 100 sixteen-byte reads and 320
//     FMAs per frame.
// LDS per workgroup (n_mel = 80): 8 waves x (4 transpose tiles + 16 x 81 log-mel) + DCT table [80][80]
// (row stride 80: the four k rows of an MFMA B operand fall on the two bank halves) + CSR = 141 KB.
constexpr int NW3 = 8;
constexpr int DSTR = 80;            // DCT table row stride (floats)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NW3 * 64, 1) void stft3_kernel(Stft2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const kws_stft_plan& pl = a.pl;
  const int n_mel = pl.n_mel, n_out = pl.n_out;
  const int LMS = n_mel + 1;                                       // log-mel row stride (odd: conflict-free columns)
  float* s_dct = lds;                                              // [n_mel][DSTR]
  float* s_bw = s_dct + n_mel * DSTR;                              // [n_w]
  int* s_csr = reinterpret_cast<int*>(s_bw + ((pl.n_w + 3) & ~3)); // [3][128] start | cnt | ofs
  float* s_wave = reinterpret_cast<float*>(s_csr + 3 * 128);
  constexpr int FR_FLOATS = 16 * TP * 2;                           // one frame's transpose tile (544 floats)
  const int wave_floats = 4 * FR_FLOATS + ((16 * LMS + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, fq = lane >> 4;
  float* s_fr = s_wave + wave * wave_floats + fq * FR_FLOATS;      // this frame's scratch
  float* s_lm16 = s_wave + wave * wave_floats + 4 * FR_FLOATS;     // [16][LMS] log-mel rows of the super-quad

  for (int i = tid; i < n_mel * DSTR; i += NW3 * 64) {
    const int m = i / DSTR, q = i - m * DSTR;
    s_dct[i] = q < 64 ? pl.dct64[m * 64 + q] : 0.f;
  }
  for (int i = tid; i < pl.n_w; i += NW3 * 64) s_bw[i] = pl.band_w[i];
  for (int i = tid; i < 128; i += NW3 * 64) {
    s_csr[i] = i < n_mel ? pl.band_start[i] : 0;
    s_csr[128 + i] = i < n_mel ? pl.band_cnt[i] : 0;
    s_csr[256 + i] = i < n_mel ? pl.band_ofs[i] : 0;
  }
  // per-lane FFT constants
  float2 r_win[16], r_tw[16], r_w512[8];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) r_win[n1] = *reinterpret_cast<const float2*>(pl.window + 2 * (16 * n1 + l16));
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) r_tw[k1] = pl.tw16[l16 * 16 + k1];
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) r_w512[k2] = pl.w512[l16 + 16 * k2];
  __syncthreads();

  const int64_t n_super = (a.total_quads + 3) / 4;
  const int64_t wave_global = (int64_t)blockIdx.x * NW3 + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * NW3;
  const int nb_mel = (n_mel + 15) >> 4;   // mel bands per lane
  for (int64_t sq = wave_global; sq < n_super; sq += wave_stride) {
#pragma unroll 1
    for (int qq = 0; qq < 4; ++qq) {
      const int64_t quad = sq * 4 + qq;
      const bool quad_ok = quad < a.total_quads;
      const int64_t b = quad_ok ? quad / a.quads_per_clip : 0;
      const int f = quad_ok ? (int)(quad - b * a.quads_per_clip) * 4 + fq : 0;
      const bool live = quad_ok && f < a.F;
      const float* fx = a.x + b * (int64_t)a.L + (int64_t)(live ? f : 0) * pl.frame_step;
      // ---- load + window: lane n2 = l16 takes z[16 n1 + n2] -----------------------------------------
      float2 z[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int s = 2 * (16 * n1 + l16);
        const int sc = s < pl.frame_len ? s : pl.frame_len - 2;   // clamped; the padded window zeroes the tail
        const float2 xv = *reinterpret_cast<const float2*>(fx + sc);
        z[n1] = make_float2(xv.x * r_win[n1].x, xv.y * r_win[n1].y);
      }
      fft16(z);
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], r_tw[k1]);
      float2* tile = reinterpret_cast<float2*>(s_fr);
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) tile[k1 * TP + l16] = z[k1];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) z[n2] = tile[l16 * TP + n2];
      fft16(z);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- real-input split + magnitude -------------------------------------------------------------
      float* s_mag = s_fr;                            // [260]
      // X[k] = E + T and X[256-k] = conj(E - T) with E = (Z[k] + conj Z[256-k]) / 2, T = W512^k (Z[k] - conj Z[256-k]) / 2i:
      // one twiddle product serves both bins, so a lane only walks its bins k = l16 + 16 k2 < 128 (k = 0 also
      // yields the Nyquist bin 256); bin 128 pairs with itself
      const int partner = (16 - l16) & 15;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        const float2 pa = z[15 - k2], pb = z[(16 - k2) & 15];
        const float px = __shfl(pa.x, partner, 16), py = __shfl(pa.y, partner, 16);
        const float2 zn0 = (l16 == 0) ? pb : make_float2(px, py);
        const float2 zk = z[k2];
        const float2 zn = make_float2(zn0.x, -zn0.y);
        const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y));
        const float2 dd = csub(zk, zn);
        const float2 O = make_float2(0.5f * dd.y, -0.5f * dd.x);
        const float2 T = cmul(r_w512[k2], O);
        const float2 Xp = cadd(E, T), Xm = csub(E, T);
        const int kk = l16 + 16 * k2;
        s_mag[kk] = __builtin_amdgcn_sqrtf(Xp.x * Xp.x + Xp.y * Xp.y);
        s_mag[256 - kk] = __builtin_amdgcn_sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);
      }
      if (l16 == 0) {                                 // bin 128: Z[128] with itself, W512^128 = -i
        const float2 zk = z[8];
        // E = Re Z (real part) .. general formula with zn = conj(zk): E = (Re zk, 0), O = (Im zk, 0), T = -i * Im zk
        s_mag[128] = __builtin_amdgcn_sqrtf(zk.x * zk.x + zk.y * zk.y);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- sparse mel bands + log -> row 4 qq + fq of the super-quad's log-mel block ----------------------
      float* lm_row = s_lm16 + (4 * qq + fq) * LMS;
      for (int i = 0; i < nb_mel; ++i) {
        const int m = l16 + 16 * i;
        if (m < n_mel) {
          const int st0 = s_csr[m], cnt = s_csr[128 + m], ofs = s_csr[256 + m];
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
          for (int j = 0; j < cnt; j += 4) {
            const int j1 = j + 1 < cnt ? j + 1 : j, j2 = j + 2 < cnt ? j + 2 : j, j3 = j + 3 < cnt ? j + 3 : j;
            const float m0 = s_mag[st0 + j], m1 = s_mag[st0 + j1], m2 = s_mag[st0 + j2], m3 = s_mag[st0 + j3];
            const float w0 = s_bw[ofs + j];
            const float w1 = j + 1 < cnt ? s_bw[ofs + j1] : 0.f;
            const float w2 = j + 2 < cnt ? s_bw[ofs + j2] : 0.f;
            const float w3 = j + 3 < cnt ? s_bw[ofs + j3] : 0.f;
            s0 = fmaf(m0, w0, s0);
            s1 = fmaf(m1, w1, s1);
            s2 = fmaf(m2, w2, s2);
            s3 = fmaf(m3, w3, s3);
          }
          float sm = ((s0 + s1) + (s2 + s3)) + pl.log_offset;
          if (pl.log_floor > 0.f) sm = fmaxf(sm, pl.log_floor);
          lm_row[m] = __logf(sm);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                // s_mag reads done before the next quad reuses the tile
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- DCT of the 16 frames on the matrix pipe: D[frame][q] = sum_m logmel[frame][m] dct[m][q] ----------
    // A: lane -> (frame = lane % 16, k = lane / 16); B: lane -> (k = lane / 16, q = 16 nb + lane % 16);
    // D: lane -> q = 16 nb + lane % 16, frames 4 (lane / 16) + v, i.e. quad lane/16, frame-in-quad v
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = s_lm16 + l16 * LMS + fq;
    const float* pb = s_dct + fq * DSTR + l16;
    // All four 16-column blocks every step (the table is zero padded to 64 columns: no per-block branch), operands of
    // step s+1 read while the MFMAs of step s issue.  (Written as `if (nb < n_blk) mfma(pa[ks], pb[...])` the loop
    // compiled to a read, an lgkmcnt(0) and a branch in front of every single MFMA.)
    {
      float av = pa[0], bv[4], an = 0.f, bn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) bv[nb] = pb[16 * nb];
      for (int ks = 0; ks < n_mel; ks += 4) {
        if (ks + 4 < n_mel) {
          an = pa[ks + 4];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) bn[nb] = pb[(ks + 4) * DSTR + 16 * nb];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nb], acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        av = an;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bv[nb] = bn[nb];
      }
    }
    {
      const int64_t quad = sq * 4 + fq;               // lane group fq holds the frames of quad fq
      if (quad < a.total_quads) {
        const int64_t b = quad / a.quads_per_clip;
        const int f0 = (int)(quad - b * a.quads_per_clip) * 4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (f0 + v < a.F) {
            float* orow = a.out + (b * a.F + f0 + v) * (int64_t)n_out + l16;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
              if (16 * nb + l16 < n_out) orow[16 * nb] = acc[nb][v];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                  // log-mel reads done before the next super-quad's writes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

}  // namespace

int kws_stft3_lds_bytes(const kws_stft_plan* pl) {
  const size_t floats = (size_t)pl->n_mel * DSTR + ((pl->n_w + 3) & ~3) + 3 * 128 +
                        (size_t)NW3 * (4 * (16 * TP * 2) + ((16 * (pl->n_mel + 1) + 3) & ~3));
  return (int)(floats * 4);
}

int kws_stft3_launch(const kws_stft_plan* pl, const float* x, int B, int L, int F, float* out, hipStream_t st) {
  KWS_REQUIRE(pl->n_out <= 64 && pl->n_mel % 4 == 0 && pl->n_mel <= 128, "stft3: n_mel=%d n_out=%d unsupported",
              pl->n_mel, pl->n_out);
  KWS_REQUIRE(F > 0 && (pl->frame_step % 2) == 0 && (L % 2) == 0, "stft3: bad geometry");
  Stft2Args a;
  a.pl = *pl;
  a.x = x; a.out = out; a.B = B; a.L = L; a.F = F;
  a.quads_per_clip = (F + 3) / 4;
  a.total_quads = (int64_t)B * a.quads_per_clip;
  const int bytes = kws_stft3_lds_bytes(pl);
  KWS_REQUIRE(bytes <= 160 * 1024, "stft3: LDS need %d B exceeds 160 KiB", bytes);
  static bool attr_done = false;
  if (!attr_done) {
    KWS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_done = true;
  }
  int64_t wgs = ((a.total_quads + 3) / 4 + NW3 - 1) / NW3;
  if (wgs > 256) wgs = 256;   // persistent: one workgroup per CU, tables staged once
  hipLaunchKernelGGL(stft3_kernel, dim3((unsigned)wgs), dim3(NW3 * 64), (size_t)bytes, st, a);
  KWS_LAUNCH_CHECK("stft3_kernel");
  return KWS_OK;
}

namespace {

}  // namespace

int kws_stft2_launch(const kws_stft_plan* pl, const float* x, int B, int L, int F, float* out, hipStream_t st) {
  KWS_REQUIRE(pl->n_out <= 64 && pl->n_mel <= 128, "stft2: n_mel=%d n_out=%d unsupported", pl->n_mel, pl->n_out);
  KWS_REQUIRE(F > 0 && (pl->frame_step % 2) == 0 && (L % 2) == 0, "stft2: bad geometry");
  Stft2Args a;
  a.pl = *pl;
  a.x = x; a.out = out; a.B = B; a.L = L; a.F = F;
  a.quads_per_clip = (F + 3) / 4;
  a.total_quads = (int64_t)B * a.quads_per_clip;
  const size_t floats = 512 + 2 * 256 + 2 * 258 + (size_t)pl->n_mel * 64 + ((pl->n_w + 3) & ~3) + 3 * 128 +
                        (size_t)NW2 * 4 * (16 * TP * 2);
  const size_t bytes = floats * 4;
  KWS_REQUIRE(bytes <= 160 * 1024, "stft2: LDS need %zu B exceeds 160 KiB", bytes);
  static bool attr_done = false;
  if (!attr_done) {
    KWS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_done = true;
  }
  int64_t wgs = (a.total_quads + NW2 - 1) / NW2;
  if (wgs > 256) wgs = 256;   // persistent: one workgroup (8 waves) per CU, tables staged once
  hipLaunchKernelGGL(stft2_kernel, dim3((unsigned)wgs), dim3(NW2 * 64), bytes, st, a);
  KWS_LAUNCH_CHECK("stft2_kernel");
  return KWS_OK;
}
