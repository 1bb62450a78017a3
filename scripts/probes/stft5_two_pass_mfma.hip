// EXPERIMENT, NOT BUILT INTO libkws_hip.so (round 3): measured 51.9 us against 49.2 us for the shipped stft4_kernel - see
// profiles/r03_stft5_experiment.txt and DESIGN.md section 5.  Kept as the record of what was measured (it compiles against
// the round-3 plan fields c5 / tw5 / img5, which the product no longer carries).
//
// STFT -> |X| -> mel -> log -> DCT feature kernel, fifth generation (reference
// input_data.py:361-381, audio.py:15-23).
//
// The 512-point real FFT of a frame is the 256-point complex FFT of its even/odd packed samples, split 16 x 16:
//
//   z[m] = w x[2m] + i w x[2m+1],  m = 16 n1 + n2
//   Y[k1][n2]      = sum_n1 z[16 n1 + n2] W16^(n1 k1)                    stage 1
//   Z[k1 + 16 k2]  = sum_n2 (Y[k1][n2] W256^(n2 k1)) W16^(n2 k2)         twiddle, stage 2
//
// The fourth kernel (round 2) ran stage 1 on the matrix pipe and stage 2 as a 16-point FFT in registers with window,
// twiddles and split factors read from LDS: per frame quad 637 vector and 73 LDS instructions, the vector pipe 63 % busy,
// the LDS pipe 51 % (38 % of that bank conflicts) - profiles/r03_stft_sq_before_rewrite.json.  Here BOTH stages are
// v_mfma_f32_16x16x32_f16 products (exact-to-f32 through two-way fp16 splits, three products each, as gemm_f16x2.hip),
// ONE frame per product chain, chained through registers:
//
//   stage 1   D1[n2][k1] = A1[n2][(n1, re/im)] . B1[(n1, re/im)][k1]     A1: windowed PCM gathered from global memory
//                                                                        (lane = (row n2, K group): 8-byte loads, 128
//                                                                        contiguous bytes per 16-lane row), B1 constant
//   twiddle   elementwise on D1's registers: lane (column k1, row group g) holds n2 = 4 g .. 4 g + 3 - FOUR complex
//             constants per lane, in registers
//   stage 2   D2[k2][k1] = A2[k2][(n2, re/im)] . B2[(n2, re/im)][k1]     B2 = the twiddled D1 AS IT LIES: the rows of a D
//                                                                        tile (4 g + i) are exactly the K elements a B
//                                                                        operand's lane group g wants; A2 constant
//   split     X[k], X[256 - k] from Z[k] and conj Z[256 - k]: with the columns in the order P1 (k1 <-> 16 - k1 in mirrored
//             lanes) and the rows in the order P2 (k2 <-> 15 - k2 two registers apart) the partner is a DPP row_mirror of
//             register i + 2; only the k1 = 0 / k1 = 8 columns pick other sources (one DPP row_bcast:15, blends).
// No window / twiddle / split-factor table in LDS any more (20 registers of per-lane constants instead of 19 ds_read2_b64
// per quad), no 16-point register FFT (160 vector instructions per quad) - scripts/emulate_stft5.py replays the index
// algebra in NumPy against numpy.fft.rfft.  Everything after the split - magnitudes to LDS, fixed tap-window mel bands, log,
// DCT of 16 frames on the matrix pipe, the LDS quad counter - is the fourth kernel's.
//
// Scales (all powers of two, exact): windowed samples x 2^10 and DFT constants x 2^14 into fp16 pairs (|x w| < 64), stage-1
// sums x 2^24; the twiddles carry 2^-24 x 2^9, so stage 2 splits Y' x 2^9 (|Y'| <= 11.4 |x w|: inputs up to |x| < 11, PCM is
// within [-1, 1]); stage-2 sums x 2^23 -> magnitudes x 2^23, undone by the mel weights (x 2^-23).
#include "stft_common.h"

#include <initializer_list>

using namespace kws_fft;

// -DKWS_STFT_STAMP builds (scripts/build_variant.sh, scripts/stamps_stft.py): wave 0 of every workgroup accumulates
// s_memtime deltas per phase: [0] loads issued -> stage-1 products done, [1] twiddle + stage 2, [2] split + magnitudes,
// [3] mel + log, [4] DCT + store, [5] passes, [6] total cycles, [7] total in 100 MHz ticks
#ifdef KWS_STFT_STAMP
__device__ unsigned long long g_stft_stamps[256][12];
extern "C" int kws_debug_read_stft_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stft_stamps), sizeof(g_stft_stamps));
}
#define ST_DECL unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_mark = __builtin_amdgcn_s_memtime(); \
  const unsigned long long st_t0 = st_mark, st_r0 = __builtin_amdgcn_s_memrealtime();
#define ST(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
  st_acc[i] += n_ - st_mark; st_mark = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ST_DECL
#define ST(i)
#endif

// -DKWS_STFT_ABL=<bits> builds (timing only, results wrong): 1 = without the stage-2 products, 2 = without the stage-2 operand
// splits (and twiddles), 4 = without the stage-1 products
#ifndef KWS_STFT_ABL
#define KWS_STFT_ABL 0
#endif

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int DSTR5 = 80;           // DCT table row stride (floats) of the f32 form
constexpr int MAGF = 260;           // floats of one frame's magnitude row that the mel tap windows may read (257 bins + zeros)
constexpr int MAGS = 264;           // row stride: + a spare slot (index MAGF) for the lanes that do not own bin 128
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float SCALE_IN = 1024.f;                        // 2^10 on the windowed samples
constexpr float SCALE_C = 16384.f;                        // 2^14 on the DFT constants of both stages
constexpr float SCALE_TW = 512.f / 16777216.f;            // twiddles: 2^-24 (stage-1 sums) x 2^9 (stage-2 input)
constexpr float SCALE_MAG_INV = 1.f / 8388608.f;          // mel weights: 2^-23 (stage-2 sums = 2^9 x 2^14)

__device__ __forceinline__ float row_mirror(float v) {   // value of lane 15 - (lane % 16) of the same 16-lane row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
}
// lanes of rows 1..3: lane 15 of the row above; row 0 keeps `keep` (DPP row_bcast:15 with row mask 0xE)
__device__ __forceinline__ float row_above_lane15(float keep, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x142, 0xE, 0xF, false));
}

// two f32 -> their fp16 parts, packed (low half = a): hi = rne(x), lo = rne(x - hi).  The residual comes from ONE
// v_fma_mix_f32 reading the fp16 half in place (the compiler's own sequence converts hi back first: 6 instructions per
// pair instead of 4; the kernel is bound by vector issue, section 5 of DESIGN.md)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  float la, lb;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(la), "v"(lb));
}

__device__ __forceinline__ float blend(unsigned m, float a, float b) {   // m all ones: a, m zero: b
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

// NB = mel bands per lane (ceil(n_mel / 16)); MCP packs, four bits per lane group i, the number of four-tap blocks the
// bands 16 i .. 16 i + 15 read (the widest of them decides; 80 mel bins: 1, 1, 2, 3, 4); MC = the largest of them (the row
// width of the weight table).  All compile-time, so the mel stage and the DCT are straight-line code whose LDS reads the
// compiler can put in flight together.
//
// The constant part of a workgroup's LDS - the quad counter, DCT operand and mel tap weights - as ONE image, made once per
// plan by stft5_image_kernel (kws_stft5_prepare) and copied by every workgroup of every launch with 16-byte loads.
template <int NB, int MC>
struct Stft5Lds {
  static constexpr int MAXW = 4 * MC;                              // taps of a mel band's window
  static constexpr int WSTR = MAXW + 4;                            // row stride of the weight table: 16-byte reads of 16
                                                                   // consecutive rows fall on disjoint banks
  static constexpr int CTR = 0, DCT = 4, WPAD = DCT + 16 * NB * DSTR5;
  __host__ __device__ static constexpr int image_floats(int n_mel) { return WPAD + n_mel * WSTR; }
  static constexpr int KB = (16 * NB + 31) / 32;                   // k-blocks of the f16 DCT product (32 mel bands each)
  static constexpr bool D16 = KB * 4 * 2 * 64 * 8 * 2 <= 16 * NB * DSTR5 * 4;   // the f16 DCT image must fit the f32 table's
                                                                               // LDS (80 bands: 24.6 of 25.6 KB; 40 bands keep f32)
};

template <int NB, int MC, int MCP>
__global__ __launch_bounds__(256) void stft5_image_kernel(kws_stft_plan pl, float* img) {
  using L = Stft5Lds<NB, MC>;
  constexpr int WSTR = L::WSTR;
  const int n_mel = pl.n_mel, tid = threadIdx.x, nthreads = blockDim.x;
  float* s_dct = img + L::DCT;                                     // [n_mel][DSTR5] or the f16 operand image
  float* s_wpad = img + L::WPAD;                                   // [n_mel][WSTR] band weights over the band's tap window
  if (L::D16) {
    // the DCT table as the B operands of v_mfma_f32_16x16x32_f16, ready to read: [kb][nb][plane][lane][8] fp16, element e of
    // lane (q = 16 nb + lane % 16, k group lane / 16) = dct[k = lane / 16 + 4 e + 32 kb][q] x 2^14, split in two parts
    _Float16* s_dh = reinterpret_cast<_Float16*>(s_dct);
    for (int i = tid; i < L::KB * 4 * 64 * 8; i += nthreads) {
      const int e = i & 7, ln = (i >> 3) & 63, nb = (i >> 9) & 3, kb = i >> 11;
      const int k = (ln >> 4) + 4 * e + 32 * kb, q = 16 * nb + (ln & 15);
      const float v = (k < n_mel ? pl.dct64[k * 64 + q] : 0.f) * 16384.f;
      const _Float16 h = (_Float16)v;
      s_dh[(((kb * 4 + nb) * 2 + 0) * 64 + ln) * 8 + e] = h;
      s_dh[(((kb * 4 + nb) * 2 + 1) * 64 + ln) * 8 + e] = (_Float16)(v - (float)h);
    }
  } else {
    for (int i = tid; i < 16 * NB * DSTR5; i += nthreads) {        // rows n_mel .. 16 NB - 1 are zero
      const int m = i / DSTR5, q = i - m * DSTR5;
      s_dct[i] = (q < 64 && m < n_mel) ? pl.dct64[m * 64 + q] : 0.f;
    }
  }
  for (int i = tid; i < n_mel * WSTR; i += nthreads) {
    // row m = the weights of bins win_m .. win_m + MAXW - 1, win_m = min(plan window start, MAGF - MAXW): the plan's
    // window (mel_maxw <= MAXW taps from mel_ws[m]) shifted right inside the row where the kernel's starts earlier.
    // The magnitudes reach the mel stage scaled by 2^23: the weights carry the inverse (exact).
    const int m = i / WSTR, q = i - m * WSTR;
    const int taps = 4 * ((MCP >> (4 * (m >> 4))) & 15);           // what the kernel reads for this band's group
    const int ws0 = pl.mel_ws[m];
    const int win = ws0 + taps <= MAGF ? ws0 : MAGF - taps;
    const int j = q - (ws0 - win);
    s_wpad[i] = (q < taps && j >= 0 && j < pl.mel_maxw) ? pl.mel_wpad[m * pl.mel_maxw + j] * SCALE_MAG_INV : 0.f;
  }
  for (int i = tid; i < 4; i += nthreads) img[L::CTR + i] = 0.f;  // the quad counter starts at zero
}

// NW5 waves per workgroup; a wave hands the log-mel rows of GQ quads (4 GQ frames) to one DCT.
template <int NB, int MC, int MCP, int NW5, int GQ>
__global__ __launch_bounds__(NW5 * 64, 1) void stft5_kernel(Stft2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef KWS_STFT_STAMP
  const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const kws_stft_plan& pl = a.pl;
  const int n_mel = pl.n_mel, n_out = pl.n_out;
  constexpr int LMS = 16 * NB + 1;                                 // log-mel row stride (odd: conflict-free columns); the
                                                                   // columns n_mel .. 16 NB - 1 hold finite values that meet zero DCT rows
  using LT = Stft5Lds<NB, MC>;                                     // the table image, then the waves' rows
  constexpr int WSTR = LT::WSTR;
  int* s_ctr = reinterpret_cast<int*>(lds + LT::CTR);              // [4] the workgroup's quad counter
  float* s_dct = lds + LT::DCT;
  float* s_wpad = lds + LT::WPAD;
  float* s_wave = lds + LT::image_floats(n_mel);
  constexpr int LMR = 4 * GQ;                                      // log-mel rows of a group
  constexpr int wave_floats = 4 * MAGS + ((LMR * LMS + 3) & ~3);
  constexpr int KB = LT::KB;
  constexpr bool D16 = LT::D16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform by construction; SAYING so keeps the quad
                                                                   // arithmetic and the buffer descriptors in scalar registers
  const int l16 = lane & 15, fq = lane >> 4;                       // column / row-in-tile l16, lane group fq
  float* s_magw = s_wave + wave * wave_floats;                     // [4 frames][MAGF] magnitudes of the current quad
  float* s_lm16 = s_magw + 4 * MAGS;                               // [16][LMS] log-mel rows of a group of four quads

  {
    // tables: one 16-byte copy of the plan's image
    const float4* src = reinterpret_cast<const float4*>(pl.img5);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int n4 = LT::image_floats(n_mel) / 4;
    constexpr int TRIPS = 4;
    for (int i0 = tid; i0 < n4; i0 += TRIPS * NW5 * 64) {
      float4 v[TRIPS];
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int i = i0 + u * NW5 * 64;
        v[u] = src[i < n4 ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int i = i0 + u * NW5 * 64;
        if (i < n4) dst[i] = v[u];
      }
    }
  }
  if (l16 < MAGF - 257) s_magw[fq * MAGS + 257 + l16] = 0.f;   // the tap windows may reach past the Nyquist bin: finite zeros there
  // log-mel rows start finite too: a partial last group multiplies rows it never wrote (their outputs are not stored)
  for (int i = lane; i < LMR * LMS; i += 64) s_lm16[i] = 0.f;

  // ---- per-lane constants --------------------------------------------------------------------------------------------
  // both DFT operands (stft.hip makes them in double: c5[stage][lane][j][tile]): element j of the lane's 8 <-> contraction index
  // q = 4 (lane / 16) + j / 2, part j % 2 (re / im of the data); x 2^14, two fp16 parts
  f16x8 hb1[2], hb2[2], ha1[2], ha2[2];
#pragma unroll
  for (int stg = 0; stg < 2; ++stg) {
    float c[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(pl.c5 + (stg * 64 + lane) * 16 + 4 * q);
      c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = c[2 * j + ct] * SCALE_C;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        if (stg == 0) { hb1[ct][j] = h; hb2[ct][j] = l; } else { ha1[ct][j] = h; ha2[ct][j] = l; }
      }
  }
  // window of the lane's A1 samples (row n2 = l16, K group fq): sample pairs 2 (16 (4 fq + jj) + l16), + 1; x 2^10
  float2 r_win[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const float2 w = *reinterpret_cast<const float2*>(pl.window + 128 * fq + 32 * jj + 2 * l16);
    r_win[jj] = make_float2(w.x * SCALE_IN, w.y * SCALE_IN);
  }
  // twiddles W256^(n2 k1) of the lane's D1 registers (column l16 <-> k1, rows n2 = 4 fq + i), with the scales folded in
  float2 r_tw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 t = pl.tw5[l16 * 16 + 4 * fq + i];
    r_tw[i] = make_float2(t.x * SCALE_TW, t.y * SCALE_TW);
  }
  const int k1 = l16 == 0 ? 8 : (l16 == 15 ? 0 : (l16 < 8 ? l16 : l16 + 1));   // P1[l16]
  // the lane's two primary bins k1 + 16 (2 fq + i), i = 0, 1, and their split factors W512^k
  float2 r_w5[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) r_w5[i] = pl.w512[k1 + 16 * (2 * fq + i)];
  const unsigned m0 = l16 == 0 ? 0xFFFFFFFFu : 0u, m15 = l16 == 15 ? 0xFFFFFFFFu : 0u;
  const int i128 = lane == 63 ? 128 : MAGF;                         // bin 128's owner; everyone else writes the spare slot
  // mel stage: first bin of the tap window and offset of the weight row of this lane's band l16 + 16 i (band 0's for
  // lanes past n_mel: they compute and do not store)
  int r_mws[NB], r_wofs[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int m = l16 + 16 * i < n_mel ? l16 + 16 * i : 0;
    const int ws0 = pl.mel_ws[m];
    const int taps = 4 * ((MCP >> (4 * i)) & 15);
    r_mws[i] = ws0 + taps <= MAGF ? ws0 : MAGF - taps;   // as in the image kernel
    r_wofs[i] = m * WSTR;
  }

  // Work items are QUADS of frames.  Every workgroup owns a contiguous range of them (neighbouring frames share samples
  // in L1 / L2) and its waves draw quads from a counter in LDS (the waves of a SIMD do not run at the same speed).  A wave
  // collects up to four quads and hands their 16 log-mel rows to one DCT.
  const int64_t q_lo = a.total_quads * blockIdx.x / gridDim.x, q_hi = a.total_quads * (blockIdx.x + 1) / gridDim.x;
  auto grab_issue = [&]() -> int {
    int v = 0;
    if (lane == 0) v = atomicAdd(s_ctr, 1);
    return v;
  };
  auto grab_value = [&](int v) -> int64_t { return q_lo + __builtin_amdgcn_readfirstlane(v); };
  // PCM of one quad into registers: frame u, K pair jj: the complex sample 16 (4 fq + jj) + l16 of frame u.  The buffer
  // descriptor spans ONE clip, so a frame's zero-padded tail and the frames past the clip's last one read zeros through the
  // range check (rows of frames >= F are never stored).
  float2 xv[4][4];
  auto issue_loads = [&](int64_t quad) {
    const unsigned qu = (unsigned)quad, qpc = (unsigned)a.quads_per_clip;   // < 2^31 quads (host-checked): 32-bit division
    const int64_t b = qu / qpc;
    const int f0 = (int)(qu - (unsigned)b * qpc) * 4;
    const int voff = (128 * fq + 2 * l16) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // one descriptor per frame (scalar work): base = the frame's first sample, range = what is left of the clip - the
      // zero-padded tail of the last frames reads zeros through the range check whatever the offset's parts are made of
      const int fu = f0 + u < a.F ? f0 + u : 0;            // frames past the last one re-read frame 0 (never stored)
      const int s0 = fu * pl.frame_step;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + b * (int64_t)a.L + s0), 0,
                                                                          (a.L - s0) * 4, 0x00020000);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 128 * jj, 0, 0);
        xv[u][jj] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
      }
    }
  };
  __syncthreads();                                  // tables copied, rows zeroed, counter at zero
  ST_DECL
  int64_t cur = grab_value(grab_issue());
  if (cur < q_hi) issue_loads(cur);
  while (cur < q_hi) {
    int gq[4] = {-1, -1, -1, -1};                   // the quads of this group (scalar registers)
#pragma unroll 1
    for (int qq = 0; qq < GQ; ++qq) {
      if (cur >= q_hi) break;                       // wave-uniform: the group is partial
      const int64_t quad = cur;
      if (qq == 0) gq[0] = (int)quad; else if (qq == 1) gq[1] = (int)quad; else if (qq == 2) gq[2] = (int)quad; else gq[3] = (int)quad;
#ifdef KWS_STFT_STAMP
      st_acc[5] += 1;
      st_mark = __builtin_amdgcn_s_memtime();
#endif
      const int ticket = grab_issue();              // the next quad's number: asked for now, needed after stage 1
      // ---- stage 1: one product chain per frame -------------------------------------------------------------
      f32x4 d1[4][2];
      {
        // the operands of all four frames first, then the three products of the eight (frame, tile) chains product by
        // product: a chain's dependent instructions are eight independent ones apart (chain by chain the compiler had to pad
        // every dependent pair with s_nop)
        f16x8 a1[4], a2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          u32x4 p1, p2;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            unsigned h, l;
            split2(xv[u][jj].x * r_win[jj].x, xv[u][jj].y * r_win[jj].y, h, l);
            p1[jj] = h; p2[jj] = l;
          }
          a1[u] = __builtin_bit_cast(f16x8, p1);
          a2[u] = __builtin_bit_cast(f16x8, p2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (KWS_STFT_ABL & 4) { d1[u][c] = __builtin_bit_cast(f32x4, c ? a1[u] : a2[u]); continue; }
            d1[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[u], hb1[c], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c) if (!(KWS_STFT_ABL & 4)) d1[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[u], hb2[c], d1[u][c], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c) if (!(KWS_STFT_ABL & 4)) d1[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[u], hb1[c], d1[u][c], 0, 0, 0);
      }
      // the PCM of this wave's NEXT quad is requested as soon as the products have consumed this quad's: it lands while
      // the rest of this quad runs
      __builtin_amdgcn_sched_barrier(0);
      cur = grab_value(ticket);
      if (cur < q_hi) issue_loads(cur);
      __builtin_amdgcn_sched_barrier(0);
#ifdef KWS_STFT_STAMP
      asm volatile("" :: "v"(d1[0][0][0]), "v"(d1[3][1][3]));   // the MFMA results have landed
#endif
      ST(0);
      // ---- twiddle + stage 2 ------------------------------------------------------------------------------------
      // lane (column l16 <-> k1, group fq): d1[u][0 / 1][i] = Re / Im Y[k1][n2 = 4 fq + i] x 2^24; B2 element 2 i + ri
      f32x4 d2[4][2];
      {
        f16x8 b1[4], b2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          u32x4 p1, p2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (KWS_STFT_ABL & 2) { p1[i] = __float_as_uint(d1[u][0][i]); p2[i] = __float_as_uint(d1[u][1][i]); continue; }
            const float2 y = cmul(make_float2(d1[u][0][i], d1[u][1][i]), r_tw[i]);
            unsigned h, l;
            split2(y.x, y.y, h, l);
            p1[i] = h; p2[i] = l;
          }
          b1[u] = __builtin_bit_cast(f16x8, p1);
          b2[u] = __builtin_bit_cast(f16x8, p2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (KWS_STFT_ABL & 1) { d2[u][c] = __builtin_bit_cast(f32x4, c ? b1[u] : b2[u]); continue; }
            d2[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1[c], b2[u], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c) if (!(KWS_STFT_ABL & 1)) d2[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha2[c], b1[u], d2[u][c], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 2; ++c) if (!(KWS_STFT_ABL & 1)) d2[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1[c], b1[u], d2[u][c], 0, 0, 0);
      }
#ifdef KWS_STFT_STAMP
      asm volatile("" :: "v"(d2[0][0][0]), "v"(d2[3][1][3]));
#endif
      ST(1);
      // ---- real-input split + magnitude ---------------------------------------------------------------------------
      // d2[u][0 / 1][i] = Re / Im Z[k1 + 16 P2[4 fq + i]] (x 2^23, the halves of the split formulas folded into stage 1):
      // registers 0, 1 are k2 = 2 fq, 2 fq + 1 (this lane's primary bins), registers 2, 3 are k2 = 15 - 2 fq, 14 - 2 fq.
      // X[k] = E + T and X[256-k] = conj(E - T) with E = Z[k] + conj Z[256-k], T = W512^k (Z[k] - conj Z[256-k]) / i.
      // Z[256-k] = column 16 - k1 (the mirrored lane), k2 -> 15 - k2 (two registers on) - except in the columns k1 = 8
      // (lane 0: its own partner column) and k1 = 0 (lane 15: k2 pairs with 16 - k2: register 2 of the lane itself for
      // i = 1; for i = 0 register 3 of lane 15 one row up, and bin 0 with itself).
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float* mrow = s_magw + u * MAGS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float own_x = d2[u][0][i + 2], own_y = d2[u][1][i + 2];
          const float mir_x = row_mirror(own_x), mir_y = row_mirror(own_y);
          float p15_x, p15_y;                          // the k1 = 0 column's partner
          if (i == 0) {
            p15_x = row_above_lane15(d2[u][0][0], d2[u][0][3]);
            p15_y = row_above_lane15(d2[u][1][0], d2[u][1][3]);
          } else {
            p15_x = d2[u][0][2];
            p15_y = d2[u][1][2];
          }
          const float zn_x = blend(m0, own_x, blend(m15, p15_x, mir_x));
          const float zn_y = -blend(m0, own_y, blend(m15, p15_y, mir_y));      // conj
          const float2 zk = make_float2(d2[u][0][i], d2[u][1][i]);
          const float2 zn = make_float2(zn_x, zn_y);
          const float2 E = cadd(zk, zn);
          const float2 dd = csub(zk, zn);
          const float2 O = make_float2(dd.y, -dd.x);
          const float2 T = cmul(r_w5[i], O);
          const float2 Xp = cadd(E, T), Xm = csub(E, T);
          const int kk = k1 + 16 * (2 * fq + i);
          mrow[kk] = __builtin_amdgcn_sqrtf(Xp.x * Xp.x + Xp.y * Xp.y);
          mrow[256 - kk] = __builtin_amdgcn_sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);
        }
        {
          // k1 = 0, k2 = 8 (lane 63, register 3): Z[128] pairs with itself, |X[128]| = 2 |Z[128]| (halved scale).  Every lane
          // computes and stores - the others into the row's spare slot - so that the four frames stay ONE scheduling region
          // (an exec-masked block per frame costs the same issue slots and fences the frames off from one another)
          const float zx = d2[u][0][3], zy = d2[u][1][3];
          mrow[i128] = 2.0f * __builtin_amdgcn_sqrtf(zx * zx + zy * zy);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      ST(2);
      // ---- mel bands + log -> row 4 qq + fq of the group's log-mel block ------------------------------------------
      // Lane (l16, fq) takes bands l16 + 16 i of frame fq.  Every band reads a window of 4 mel_mc[i] taps that starts at
      // mel_ws[m] and carries zero weights outside the band: all of a band's LDS reads are in flight before its first
      // multiply-add.
      const float* s_mag = s_magw + fq * MAGS;
      float* lm_row = s_lm16 + (4 * qq + fq) * LMS;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const float* mp = s_mag + r_mws[i];
        const float* wp = s_wpad + r_wofs[i];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        constexpr int CH = MC < 8 ? MC : 8;           // blocks in flight per request burst
        const int mci = (MCP >> (4 * i)) & 15;        // a constant after unrolling: this group's blocks
#pragma unroll
        for (int t0 = 0; t0 < MC; t0 += CH) {
          float4 wv[CH];
          float mv[CH][4];
#pragma unroll
          for (int t = 0; t < CH; ++t)
            if (t0 + t < mci) {
              wv[t] = *reinterpret_cast<const float4*>(wp + 4 * (t0 + t));
#pragma unroll
              for (int r = 0; r < 4; ++r) mv[t][r] = mp[4 * (t0 + t) + r];
            }
#pragma unroll
          for (int t = 0; t < CH; ++t)
            if (t0 + t < mci) {
            s0 = fmaf(mv[t][0], wv[t].x, s0);
            s1 = fmaf(mv[t][1], wv[t].y, s1);
            s2 = fmaf(mv[t][2], wv[t].z, s2);
            s3 = fmaf(mv[t][3], wv[t].w, s3);
            }
        }
        // floor: a plain maximum (log_floor = 0 leaves the non-negative sum as it is); v_log_f32 (log2, 1 ulp) x ln 2:
        // sm >= the offset / floor, so none of logf's denormal handling is needed; with the f16 DCT the row carries that
        // product's 2^9 as well (exact).  Lanes past n_mel store too: their columns meet zero rows of the DCT operand.
        const float sm = fmaxf(((s0 + s1) + (s2 + s3)) + pl.log_offset, pl.log_floor);
        lm_row[l16 + 16 * i] = __builtin_amdgcn_logf(sm) * (D16 ? 0.6931471805599453f * 512.f : 0.6931471805599453f);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                // s_mag reads done before the next quad overwrites the rows
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      ST(3);
    }
    // ---- DCT of the 16 frames on the matrix pipe: D[frame][q] = sum_m logmel[frame][m] dct[m][q] ----------
    // A: lane -> (frame = lane % 16, k = lane / 16); B: lane -> (k = lane / 16, q = 16 nb + lane % 16);
    // D: lane -> q = 16 nb + lane % 16, frames 4 (lane / 16) + v, i.e. quad lane/16, frame-in-quad v
    f32x4 dacc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) dacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = s_lm16 + (l16 & (LMR - 1)) * LMS + fq;        // GQ < 4: rows LMR .. 15 repeat (their outputs are not stored)
    if (D16) {
      // Log-mel values lie in [-14, 12]: scaled by 2^9, two fp16 parts, three products per (k block, column tile)
      const _Float16* s_dh = reinterpret_cast<const _Float16*>(s_dct);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        u32x4 p1, p2;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {               // k = fq + 4 e + 32 kb < 16 NB (the rows hold log-mel x 2^9)
          const float v0 = (4 * e + 32 * kb + 4 <= 16 * NB) ? pa[4 * e + 32 * kb] : 0.f;
          const float v1 = (4 * e + 32 * kb + 8 <= 16 * NB) ? pa[4 * e + 32 * kb + 4] : 0.f;
          unsigned h, l;
          split2(v0, v1, h, l);
          p1[e >> 1] = h; p2[e >> 1] = l;
        }
        const f16x8 a1 = __builtin_bit_cast(f16x8, p1), a2 = __builtin_bit_cast(f16x8, p2);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const f16x8 b1 = *reinterpret_cast<const f16x8*>(s_dh + (((kb * 4 + nb) * 2 + 0) * 64 + lane) * 8);
          const f16x8 b2 = *reinterpret_cast<const f16x8*>(s_dh + (((kb * 4 + nb) * 2 + 1) * 64 + lane) * 8);
          dacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, dacc[nb], 0, 0, 0);
          dacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, dacc[nb], 0, 0, 0);
          dacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, dacc[nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int v = 0; v < 4; ++v) dacc[nb][v] *= 1.1920928955078125e-07f;      // 2^-23: exact
    } else {
      const float* pb = s_dct + fq * DSTR5 + l16;
      // operands two steps ahead of the MFMAs that use them (one step = 4 MFMAs = 128 cycles of cover, an LDS round trip
      // under 12 waves takes longer)
      float av[3], bv[3][4];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        av[d] = pa[4 * d];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bv[d][nb] = pb[(4 * d) * DSTR5 + 16 * nb];
      }
#pragma unroll
      for (int ks = 0; ks < 16 * NB; ks += 4) {
        constexpr int KLAST = 16 * NB - 4;
        const int kn = ks + 8 <= KLAST ? ks + 8 : KLAST;          // clamped: the last two requests are not used
        av[2] = pa[kn];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bv[2][nb] = pb[kn * DSTR5 + 16 * nb];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) dacc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0][nb], dacc[nb], 0, 0, 0);
        av[0] = av[1]; av[1] = av[2];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          bv[0][nb] = bv[1][nb];
          bv[1][nb] = bv[2][nb];
        }
      }
    }
    {
      const int quad = fq == 0 ? gq[0] : (fq == 1 ? gq[1] : (fq == 2 ? gq[2] : gq[3]));   // lane group fq: the group's quad fq
                                                                                          // (-1 past GQ: nothing to store)
      if (quad >= 0) {
        const unsigned qu = (unsigned)quad, qpc = (unsigned)a.quads_per_clip;
        const int64_t b = qu / qpc;
        const int f0 = (int)(qu - (unsigned)b * qpc) * 4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (f0 + v < a.F) {
            float* orow = a.out + (b * a.F + f0 + v) * (int64_t)n_out + l16;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
              if (16 * nb + l16 < n_out) orow[16 * nb] = dacc[nb][v];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                  // log-mel reads done before the next group's writes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef KWS_STFT_STAMP
    st_mark = __builtin_amdgcn_s_memtime() - st_mark;   // (not a phase of the quad loop: the DCT + store of this group)
    st_acc[4] += st_mark;
#endif
  }
#ifdef KWS_STFT_STAMP
  if (tid == 0 && blockIdx.x < 256) {
    for (int i = 0; i < 6; ++i) g_stft_stamps[blockIdx.x][i] = st_acc[i];
    g_stft_stamps[blockIdx.x][6] = __builtin_amdgcn_s_memtime() - st_t0;
    g_stft_stamps[blockIdx.x][7] = __builtin_amdgcn_s_memrealtime() - st_r0;
    g_stft_stamps[blockIdx.x][8] = st_entry;            // absolute 100 MHz ticks: kernel entry, loop start, loop end
    g_stft_stamps[blockIdx.x][9] = st_r0;
    g_stft_stamps[blockIdx.x][10] = __builtin_amdgcn_s_memrealtime();
  }
  if (lane == 0 && blockIdx.x < 256) {                  // latest wave of the workgroup to finish
    atomicMax(&g_stft_stamps[blockIdx.x][11], (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
#endif
}

}  // namespace

// instantiated band shapes: 80 mel bins over 257 bins (input_data.py:366-373 as train.py sets it: blocks 1, 1, 2, 3, 4),
// 40 mel bins (2, 5, 8), and the same lane counts with every group at the widest width; anything else declines (the generic
// kernel of stft.hip)
static int stft5_shape(const kws_stft_plan* pl) {
  if (pl->mel_maxw <= 0) return 0;
  const int nb = (pl->n_mel + 15) / 16;
  auto fits = [&](std::initializer_list<int> mc) {
    if ((int)mc.size() != nb) return false;
    int i = 0;
    for (int v : mc)
      if (pl->mel_mc[i++] > v) return false;
    return true;
  };
  if (fits({1, 1, 2, 3, 4})) return 1;
  if (fits({4, 4, 4, 4, 4})) return 2;
  if (fits({2, 5, 8})) return 3;
  if (fits({8, 8, 8})) return 4;
  return 0;
}

static int stft5_lds_bytes(const kws_stft_plan* pl, int nw, int gq) {
  const int sh = stft5_shape(pl);
  if (sh == 0) return 1 << 30;                                       // declines: the caller falls back to the generic kernel
  const int nb = sh <= 2 ? 5 : 3, mc = sh <= 2 ? 4 : 8;
  const size_t floats = 4 + (size_t)16 * nb * DSTR5 + (size_t)pl->n_mel * (4 * mc + 4) +
                        (size_t)nw * (4 * MAGS + ((4 * gq * (16 * nb + 1) + 3) & ~3));
  return (int)(floats * 4);
}
constexpr int STFT5_NW = 12, STFT5_GQ = 4;
int kws_stft5_lds_bytes(const kws_stft_plan* pl) { return stft5_lds_bytes(pl, STFT5_NW, STFT5_GQ); }

template <int NB, int MC, int MCP>
static int stft5_image_t(kws_stft_plan* pl) {
  const size_t bytes = (size_t)Stft5Lds<NB, MC>::image_floats(pl->n_mel) * sizeof(float);
  KWS_HIP(hipMalloc(reinterpret_cast<void**>(&pl->img5), bytes));
  hipLaunchKernelGGL((stft5_image_kernel<NB, MC, MCP>), dim3(1), dim3(256), 0, nullptr, *pl, pl->img5);
  KWS_LAUNCH_CHECK("stft5_image_kernel");
  KWS_HIP(hipStreamSynchronize(nullptr));
  return KWS_OK;
}

// called once by kws_stft_plan_create after the tables are uploaded: the LDS image of this plan's kernel instance
int kws_stft5_prepare(kws_stft_plan* pl) {
  pl->img5 = nullptr;
  const int sh = stft5_shape(pl);
  if (sh == 0 || pl->n_mel % 4 != 0) return KWS_OK;                  // stft5 declines this plan: nothing to prepare
  if (sh == 1) return stft5_image_t<5, 4, 0x43211>(pl);
  if (sh == 2) return stft5_image_t<5, 4, 0x44444>(pl);
  if (sh == 3) return stft5_image_t<3, 8, 0x852>(pl);
  return stft5_image_t<3, 8, 0x888>(pl);
}

template <int NB, int MC, int MCP>
static int stft5_launch_t(const Stft2Args& a, hipStream_t st) {
  const int bytes = stft5_lds_bytes(&a.pl, STFT5_NW, STFT5_GQ);
  KWS_REQUIRE(bytes <= 160 * 1024, "stft5: LDS need %d B exceeds 160 KiB", bytes);
  // per device and cheap: set on every launch (a process-wide "done" flag would miss the second device of a process)
  KWS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft5_kernel<NB, MC, MCP, STFT5_NW, STFT5_GQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int64_t wgs = (a.total_quads + STFT5_NW - 1) / STFT5_NW;
  if (wgs > 256) wgs = 256;   // persistent: one workgroup per CU, tables copied once
  hipLaunchKernelGGL((stft5_kernel<NB, MC, MCP, STFT5_NW, STFT5_GQ>), dim3((unsigned)wgs), dim3(STFT5_NW * 64), (size_t)bytes, st, a);
  KWS_LAUNCH_CHECK("stft5_kernel");
  return KWS_OK;
}

int kws_stft5_launch(const kws_stft_plan* pl, const float* x, int B, int L, int F, float* out, hipStream_t st) {
  KWS_REQUIRE(pl->n_out <= 64 && pl->n_mel % 4 == 0 && pl->n_mel <= 128, "stft5: n_mel=%d n_out=%d unsupported",
              pl->n_mel, pl->n_out);
  KWS_REQUIRE(F > 0 && (pl->frame_step % 2) == 0 && (L % 2) == 0 && (pl->frame_len % 2) == 0 &&
              (reinterpret_cast<uintptr_t>(x) & 7) == 0, "stft5: bad geometry (8-byte aligned frames)");
  const int sh = stft5_shape(pl);
  KWS_REQUIRE(pl->img5 != nullptr, "stft5: the plan carries no table image (kws_stft5_prepare)");
  KWS_REQUIRE(sh != 0, "stft5: mel band shape (n_mel=%d, up to %d taps) is not instantiated", pl->n_mel, pl->mel_maxw);
  KWS_REQUIRE((int64_t)B * ((F + 3) / 4) < (1ll << 31), "stft5: %d clips x %d frames exceed 2^31 frame quads", B, F);
  KWS_REQUIRE((int64_t)F * pl->frame_step * 4 < (1ll << 31), "stft5: clip too long for 32-bit byte offsets");
  Stft2Args a;
  a.pl = *pl;
  a.x = x; a.out = out; a.B = B; a.L = L; a.F = F;
  a.quads_per_clip = (F + 3) / 4;
  a.total_quads = (int64_t)B * a.quads_per_clip;
  if (sh == 1) return stft5_launch_t<5, 4, 0x43211>(a, st);
  if (sh == 2) return stft5_launch_t<5, 4, 0x44444>(a, st);
  if (sh == 3) return stft5_launch_t<3, 8, 0x852>(a, st);
  return stft5_launch_t<3, 8, 0x888>(a, st);
}
