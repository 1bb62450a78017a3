// STFT -> |X| -> mel -> log -> DCT feature kernel, fourth generation (reference
// input_data.py:361-381, audio.py:15-23).
//
// Round 1's kernel (stft3_kernel, removed in round 3) was bound by vector-instruction issue at 2 waves per SIMD: both
// radix-16 passes of the 16 x 16 Cooley-Tukey split ran on the vector pipe, joined by a transpose through LDS, with 94
// registers per lane of FFT constants.  Here the FIRST pass is a matrix product on the matrix pipe:
//
//   z[m] = w x[2m] + i w x[2m+1],  m = 16 n1 + n2,  Z[k1 + 16 k2] = sum_n2 W256^(n2 k1) W16^(n2 k2) Y[k1][n2]
//   Y[k1][n2] = sum_n1 z[16 n1 + n2] W16^(n1 k1)                      <- 16-point DFTs over n1 = [rows x 32] x [32 x 32]
//
// as v_mfma_f32_16x16x4_f32 products D = A B (exact f32 fma chains):
//   A  (data)      row r = 4 u + i  <->  frame u of the wave's four frames, n2 = 4 t + i of row tile t = 0..3;
//                  K index (8 chunks of 4) <-> (n1, re/im): one 8-byte load per lane feeds two chunks;
//   B  (constant)  [K][col]: col c <-> k1 = KPERM[c]; column tile 0 = Re Y, 1 = Im Y; entries 0.5 cos / 0.5 sin
//                  (the 0.5 of the real-input split is folded in: a power of two, so the products stay exact);
//   D  layout      lane (g = lane / 16, c = lane % 16), register i of tile t: row 4 g + i = (frame g, n2 = 4 t + i)
// so after 64 MFMAs lane (g, c) holds Y[k1][n2 = 0..15] of frame g in registers - exactly what the SECOND pass (a
// 16-point FFT in registers, twiddles W256^(n2 k1) first) wants.  The matrix layouts do the transpose: no LDS round
// trip between the passes, no window / twiddle constants in registers (32 instead of 94: they are LDS tables), and the column
// order KPERM = 0..7, 9..15, 8 puts the real-input split's partner bin Z[256 - k] in the MIRRORED lane of the 16-lane
// row, so the partner exchange is a DPP row_mirror operand instead of a ds_bpermute.
// Everything after the split (magnitudes to LDS, CSR mel bands, log, DCT of 16 frames on the matrix pipe, one
// 16-lane x 16-byte store per feature row) is the third kernel's, with 12 waves per workgroup (3 per SIMD) so that one
// wave's matrix work runs beside another's vector work.
// scripts/emulate_stft4.py replays this index algebra in NumPy against numpy.fft.rfft.
#include "stft_common.h"

#include <initializer_list>

using namespace kws_fft;

// -DKWS_STFT_STAMP builds (scripts/build_variant.sh, scripts/stamps_stft.py): wave 0 of every workgroup accumulates
// s_memtime deltas per phase: [0] loads issued -> first-pass MFMAs done, [1] twiddle + second pass, [2] split + magnitudes,
// [3] mel + log, [4] DCT + store, [5] passes, [6] total cycles, [7] total in 100 MHz ticks
#ifdef KWS_STFT_STAMP
__device__ unsigned long long g_stft_stamps[256][12];
extern "C" __attribute__((visibility("default"))) int kws_debug_read_stft_stamps(unsigned long long* out) {
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

// -DKWS_STFT_ABL=<bits> builds (timing only, results wrong): the kernel without 1 the first-pass products, 2 twiddle +
// second pass, 4 split + magnitudes, 8 mel + log, 16 the DCT products, 32 the output stores, 64 the PCM loads - what a
// phase costs IN the overlap of twelve waves, which its stamp (one wave's latency) does not say
#ifndef KWS_STFT_ABL
#define KWS_STFT_ABL 0
#endif
#define KEEP2(v) asm volatile("" :: "v"((v).x), "v"((v).y))

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int DSTR4 = 80;           // DCT table row stride (floats)
constexpr int MAGF = 260;           // floats of one frame's magnitude row
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row_mirror(float v) {   // value of lane 15 - (lane % 16) of the same 16-lane row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
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
// H1: the first pass on the f16 matrix instruction (see the first-pass comment in the kernel); false = v_mfma_f32_16x16x4_f32
//
// The constant part of a workgroup's LDS - window, twiddles, split factors, the quad counter, DCT operand and mel tap
// weights - as ONE image, made once per plan by stft4_image_kernel (kws_stft4_prepare) and copied by every workgroup
// of every launch with 16-byte loads.  Staged table by table in the prologue of the feature kernel it cost 5.6 us of a
// 61 us launch: eleven trips of dependent global loads (band start -> weight row) before the first frame.
template <int NB, int MC>
struct Stft4Lds {
  static constexpr int MAXW = 4 * MC;                              // taps of a mel band's window
  static constexpr int WSTR = MAXW + 4;                            // row stride of the weight table: 16-byte reads of 16
                                                                   // consecutive rows fall on disjoint banks
  static constexpr int WIN = 0, TW = 512, W5 = 1024, CTR = 1280, DCT = 1284, WPAD = DCT + 16 * NB * DSTR4;
  __host__ __device__ static constexpr int image_floats(int n_mel) { return WPAD + n_mel * WSTR; }
};

template <int NB, int MC, int MCP, bool H1>
__device__ __forceinline__ void stft4_tables(float* lds, const kws_stft_plan& pl, int tid, int nthreads) {
  using L = Stft4Lds<NB, MC>;
  constexpr int WSTR = L::WSTR;
  const int n_mel = pl.n_mel;
  float* s_win = lds + L::WIN;                                     // [512] zero padded window
  float2* s_tw = reinterpret_cast<float2*>(lds + L::TW);           // [16 n2][16 c] second-pass twiddles W256^(n2 k1)
  float2* s_w5 = reinterpret_cast<float2*>(lds + L::W5);           // [8 k2][16 c] split factors W512^(k1 + 16 k2)
  float* s_dct = lds + L::DCT;                                     // [n_mel][DSTR4]
  float* s_wpad = lds + L::WPAD;                                   // [n_mel][WSTR] band weights over the band's tap window
  // H1: the power-of-two scales of the f16 first pass live in the tables (exact): the window carries the samples' 2^10,
  // the twiddles of n2 >= 1 the products' 2^-24
  for (int i = tid; i < 512; i += nthreads) s_win[i] = H1 ? pl.window[i] * 1024.f : pl.window[i];
  for (int i = tid; i < 256; i += nthreads) {      // [c][n2] -> [n2][c]: a row's 16 lanes read 16 consecutive 8-byte
    float2 v = pl.tw4[(i & 15) * 16 + (i >> 4)];   // entries (conflict-free)
    if (H1) { v.x *= 5.9604644775390625e-08f; v.y *= 5.9604644775390625e-08f; }
    s_tw[i] = v;
  }
  for (int i = tid; i < 128; i += nthreads) s_w5[i] = pl.w512p[(i & 15) * 8 + (i >> 4)];  // [c][k2] -> [k2][c]
  constexpr int KB = (16 * NB + 31) / 32;                           // H1: k-blocks of the DCT product (32 mel bands each)
  constexpr bool D16 = H1 && KB * 4 * 2 * 64 * 8 * 2 <= 16 * NB * DSTR4 * 4;   // the f16 DCT image must fit the f32 table's LDS
                                                                                // (80 bands: 24.6 of 25.6 KB; 40 bands keep the f32 DCT)
  if (D16) {
    // the DCT table as the B operands of v_mfma_f32_16x16x32_f16, ready to read: [kb][nb][plane][lane][8] fp16, element e of
    // lane (q = 16 nb + lane % 16, k group lane / 16) = dct[k = lane / 16 + 4 e + 32 kb][q] x 2^14, split in two parts
    _Float16* s_dh = reinterpret_cast<_Float16*>(s_dct);
    for (int i = tid; i < KB * 4 * 64 * 8; i += nthreads) {
      const int e = i & 7, ln = (i >> 3) & 63, nb = (i >> 9) & 3, kb = i >> 11;
      const int k = (ln >> 4) + 4 * e + 32 * kb, q = 16 * nb + (ln & 15);
      const float v = (k < n_mel ? pl.dct64[k * 64 + q] : 0.f) * 16384.f;
      const _Float16 h = (_Float16)v;
      s_dh[(((kb * 4 + nb) * 2 + 0) * 64 + ln) * 8 + e] = h;
      s_dh[(((kb * 4 + nb) * 2 + 1) * 64 + ln) * 8 + e] = (_Float16)(v - (float)h);
    }
  } else {
  for (int i = tid; i < 16 * NB * DSTR4; i += nthreads) {          // rows n_mel .. 16 NB - 1 are zero
    const int m = i / DSTR4, q = i - m * DSTR4;
    s_dct[i] = (q < 64 && m < n_mel) ? pl.dct64[m * 64 + q] : 0.f;
  }
  }
  for (int i = tid; i < n_mel * WSTR; i += nthreads) {
    // row m = the weights of bins win_m .. win_m + MAXW - 1, win_m = min(plan window start, MAGF - MAXW): the plan's
    // window (mel_maxw <= MAXW taps from mel_ws[m]) shifted right inside the row where the kernel's starts earlier
    const int m = i / WSTR, q = i - m * WSTR;
    const int taps = 4 * ((MCP >> (4 * (m >> 4))) & 15);           // what the kernel reads for this band's group
    const int ws0 = pl.mel_ws[m];
    const int win = ws0 + taps <= MAGF ? ws0 : MAGF - taps;
    const int j = q - (ws0 - win);
    s_wpad[i] = (q < taps && j >= 0 && j < pl.mel_maxw) ? pl.mel_wpad[m * pl.mel_maxw + j] : 0.f;
  }
  for (int i = tid; i < 4; i += nthreads) lds[L::CTR + i] = 0.f;  // the quad counter starts at zero
}

template <int NB, int MC, int MCP, bool H1>
__global__ __launch_bounds__(256) void stft4_image_kernel(kws_stft_plan pl, float* img) {
  stft4_tables<NB, MC, MCP, H1>(img, pl, threadIdx.x, blockDim.x);
}

// NW4 waves per workgroup; a wave hands the log-mel rows of GQ quads (4 GQ frames) to one DCT.  GQ = 4 fills the 16 rows
// of the matrix instruction and needs 5.2 KB of log-mel rows per wave, which caps the workgroup at 12 waves; GQ = 2 (half
// of the DCT's rows idle) fits 16.  Measured at batch 1024 / 80 bands: 50.3 us with 12 waves, 51.4 us with 16 - the fourth
// wave per SIMD buys what the idle DCT rows cost - so 12 x 4 is what the launcher instantiates.
// V4: rows of the first-pass product are dealt so that a lane's row in the tiles 2p and 2p + 1 holds the complex samples n2 and
// n2 + 1 (n2 = 2 (i + 4 p) + (t & 1)): ONE 16-byte buffer load / window read feeds two tiles - 8 instead of 16 loads per
// quad (the ablation prices the 16 at 4.3 us of the launch).  Needs 16-byte aligned frames: L % 4 = 0, frame_step % 4 = 0.
template <int NB, int MC, int MCP, bool H1, int NW4, int GQ, bool V4>
__global__ __launch_bounds__(NW4 * 64, 1) void stft4_kernel(Stft2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef KWS_STFT_STAMP
  const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const kws_stft_plan& pl = a.pl;
  const int n_mel = pl.n_mel, n_out = pl.n_out;
  constexpr int LMS = 16 * NB + 1;                                 // log-mel row stride (odd: conflict-free columns); the
                                                                   // columns n_mel .. 16 NB - 1 hold finite values that meet zero DCT rows
  using LT = Stft4Lds<NB, MC>;                                     // the table image (stft4_tables), then the waves' rows
  constexpr int WSTR = LT::WSTR;
  float* s_win = lds + LT::WIN;
  float2* s_tw = reinterpret_cast<float2*>(lds + LT::TW);
  float2* s_w5 = reinterpret_cast<float2*>(lds + LT::W5);
  int* s_ctr = reinterpret_cast<int*>(lds + LT::CTR);              // [4] the workgroup's quad counter
  float* s_dct = lds + LT::DCT;
  float* s_wpad = lds + LT::WPAD;
  float* s_wave = lds + LT::image_floats(n_mel);
  constexpr int LMR = 4 * GQ;                                      // log-mel rows of a group
  constexpr int wave_floats = 4 * MAGF + ((LMR * LMS + 3) & ~3);
  constexpr int KB = (16 * NB + 31) / 32;                           // H1: k-blocks of the DCT product (32 mel bands each)
  constexpr bool D16 = H1 && KB * 4 * 2 * 64 * 8 * 2 <= 16 * NB * DSTR4 * 4;   // the f16 DCT image must fit the f32 table's LDS
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform by construction; SAYING so keeps the quad
                                                                   // arithmetic and the buffer descriptors in scalar registers
                                                                   // (left to itself the compiler built a waterfall loop per load)
  const int l16 = lane & 15, fq = lane >> 4;                       // D role: column c = l16, frame g = fq
  const int ar_u = l16 >> 2, ar_i = l16 & 3;                       // A role: row l16 = (frame u, n2 % 4), k slot fq
  float* s_mag = s_wave + wave * wave_floats + fq * MAGF;          // this lane row's frame
  float* s_lm16 = s_wave + wave * wave_floats + 4 * MAGF;          // [16][LMS] log-mel rows of a group of four quads

  {
    // tables: one 16-byte copy of the plan's image (stft4_tables above)
    const float4* src = reinterpret_cast<const float4*>(pl.img4);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int n4 = LT::image_floats(n_mel) / 4;
    // every thread's loads are requested before its first LDS store (as a plain loop the compiler waits for each load
    // before the next: one L2 round trip per 12 KB of the 37 KB image)
    constexpr int TRIPS = 4;
    for (int i0 = tid; i0 < n4; i0 += TRIPS * NW4 * 64) {
      float4 v[TRIPS];
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int i = i0 + u * NW4 * 64;
        v[u] = src[i < n4 ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < TRIPS; ++u) {
        const int i = i0 + u * NW4 * 64;
        if (i < n4) dst[i] = v[u];
      }
    }
  }
  if (l16 < MAGF - 257) s_mag[257 + l16] = 0.f;    // the tap windows may reach past the Nyquist bin: finite zeros there
  // log-mel rows start finite too: a partial last group multiplies rows it never wrote (their outputs are not stored)
  for (int i = lane; i < LMR * LMS; i += 64) s_lm16[i] = 0.f;
  // per-lane constants: B operand of the 16 (k-chunk, column tile) products
  float r_b[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(pl.b4 + lane * 16 + 4 * q);
    r_b[4 * q] = v.x; r_b[4 * q + 1] = v.y; r_b[4 * q + 2] = v.z; r_b[4 * q + 3] = v.w;
  }
  // H1: the same 16 constants as the B operands of two v_mfma_f32_16x16x32_f16 (column tiles c = 0, 1): element e = 2 j' + s
  // of the lane's 8 is the constant that multiplied (s = 0: the even / s = 1: the odd sample of) k-chunk j' - the sum over
  // the 32 k of one f16 instruction is the sum of the eight K = 4 f32 instructions it replaces, term for term.  Scaled by
  // 2^14 (|b| <= 0.5) and split into two fp16 parts; the samples are scaled by 2^10 (|x w| < 64) and split the same way:
  // three products h2.h1 + h1.h2 + h1.h1 carry 22 bits (csrc/gemm_f16x2.hip has the argument), the result is multiplied by 2^-24.
  f16x8 hb1[2], hb2[2];
  if (H1) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = r_b[4 * (e >> 1) + 2 * (e & 1) + c] * 16384.f;
        const _Float16 h = (_Float16)v;
        hb1[c][e] = h;
        hb2[c][e] = (_Float16)(v - (float)h);
      }
  }
  // mel stage: first bin of the tap window and offset of the weight row of this lane's band l16 + 16 i (band 0's for
  // lanes past n_mel: they compute and do not store)
  int r_mws[NB], r_wofs[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int m = l16 + 16 * i < n_mel ? l16 + 16 * i : 0;
    const int ws0 = pl.mel_ws[m];
    const int taps = 4 * ((MCP >> (4 * i)) & 15);
    r_mws[i] = ws0 + taps <= MAGF ? ws0 : MAGF - taps;   // as in the staging loop above
    r_wofs[i] = m * WSTR;
  }
  const int k1 = l16 < 8 ? l16 : (l16 < 15 ? l16 + 1 : 8);         // KPERM[l16]
  const unsigned m0 = l16 == 0 ? 0xFFFFFFFFu : 0u, m15 = l16 == 15 ? 0xFFFFFFFFu : 0u;
  // sample offsets of this lane's A-operand loads: 2 (16 (4 j' + s) + 4 t + i), s = fq, i = ar_i
  // sample offset (floats) of the lane's A-operand data in tile t, k-chunk j': the complex sample 16 (4 j' + fq) + n2 with
  // n2 = 4 t + i, or V4: n2 = 2 (i + 4 (t >> 1)) + (t & 1)
  const int a_off0 = V4 ? 32 * fq + 4 * ar_i : 2 * (16 * fq + ar_i);
  auto a_off = [&](int t, int jp) { return a_off0 + 128 * jp + (V4 ? 16 * (t >> 1) + 2 * (t & 1) : 8 * t); };

  // Work items are QUADS of frames.  Every workgroup owns a contiguous range of them (neighbouring frames share samples
  // in L1 / L2) and its waves draw quads from a counter in LDS: the waves of a SIMD do not run at the same speed - the
  // oldest wave wins the issue arbitration and, with a fixed 8 or 9 quads per wave, finished at 56 us while the youngest
  // ran alone, latency-bound, until 94 us.  A wave collects up to four quads and hands their 16 log-mel rows to one DCT.
  const int64_t q_lo = a.total_quads * blockIdx.x / gridDim.x, q_hi = a.total_quads * (blockIdx.x + 1) / gridDim.x;
  // the draw is split in two so that the LDS round trip of the atomic runs under the first-pass MFMAs: grab_issue early in
  // a pass, grab_value (which waits for it) when the next quad's loads are about to be issued
  auto grab_issue = [&]() -> int {
    int v = 0;
    if (lane == 0) v = atomicAdd(s_ctr, 1);
    return v;
  };
  auto grab_value = [&](int v) -> int64_t { return q_lo + __builtin_amdgcn_readfirstlane(v); };
  // A role: PCM of one quad into registers (frames past the clip's last one re-read frame 0: their rows are never stored)
  float2 xv[4][4];
  // Buffer loads: the descriptor spans ONE clip, so a frame's zero-padded tail (samples 480..511 of the last frames run
  // past the clip) reads as zeros through the range check instead of needing a clamped address per load - the 16 clamped
  // 64-bit offsets were loop invariants that cost 32 registers - and a lane addresses all 16 loads with one register plus
  // immediates.
  auto issue_loads = [&](int64_t quad) {
    const unsigned qu = (unsigned)quad, qpc = (unsigned)a.quads_per_clip;   // < 2^31 quads (host-checked): 32-bit division
    const int64_t b = qu / qpc;
    const int f0 = (int)(qu - (unsigned)b * qpc) * 4;
    const int fu = f0 + ar_u < a.F ? f0 + ar_u : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + b * (int64_t)a.L), 0,
                                                                        a.L * 4, 0x00020000);
    const int voff = (fu * pl.frame_step + a_off0) * 4;
    if (V4) {
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          if (KWS_STFT_ABL & 64) { xv[2 * p2][jp] = xv[2 * p2 + 1][jp] = make_float2(__int_as_float(voff + p2), __int_as_float(voff + jp)); continue; }
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 512 * jp + 64 * p2, 0, 0);
          xv[2 * p2][jp] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
          xv[2 * p2 + 1][jp] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
        }
      return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        if (KWS_STFT_ABL & 64) { xv[t][jp] = make_float2(__int_as_float(voff + t), __int_as_float(voff + jp)); continue; }
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 512 * jp + 32 * t, 0, 0);
        xv[t][jp] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
      }
  };
  __syncthreads();                                  // tables copied, rows zeroed, counter at zero
  // the eight split factors of the lane stay in registers for all its quads (16 of the 40 the budget of three waves per SIMD
  // leaves: 48.0 -> 47.5 us; the twiddles in registers as well - 4 / 8 / 12 of the 15 - bought nothing more and cost the
  // 40-band instance its occupancy)
  float2 r_w5[8];
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) r_w5[k2] = s_w5[k2 * 16 + l16];
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
      const int ticket = grab_issue();              // the next quad's number: asked for now, needed after the MFMAs
      // ---- first pass on the matrix pipe ----------------------------------------------------------------
      f32x4 acc[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (KWS_STFT_ABL & 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) { acc[t][0][jp] = xv[t][jp].x; acc[t][1][jp] = xv[t][jp].y; }
      } else if (H1) {
        // An f32 matrix instruction holds the SIMD's vector issue for all of its 32 cycles (probe):
        // the 64 of a quad were 2,048 cycles nothing else could use.  The f16 form holds it for 8 of its 16: 24 instructions
        // (3 products x 4 row blocks x 2 column tiles) + the split of the lane's 32 windowed samples.
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          u32x4 p1, p2;
#pragma unroll
          for (int jp = 0; jp < 4; ++jp) {
            const float2 wv = *reinterpret_cast<const float2*>(s_win + a_off(t, jp));   // window x 2^10
            unsigned h, l;
            split2(xv[t][jp].x * wv.x, xv[t][jp].y * wv.y, h, l);
            p1[jp] = h; p2[jp] = l;
          }
          const f16x8 a1 = __builtin_bit_cast(f16x8, p1), a2 = __builtin_bit_cast(f16x8, p2);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, hb1[c], acc[t][c], 0, 0, 0);
            acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, hb2[c], acc[t][c], 0, 0, 0);
            acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, hb1[c], acc[t][c], 0, 0, 0);
          }
        }
        acc[0][0][0] *= 5.9604644775390625e-08f;     // 2^-24 (exact): n2 = 0 has no twiddle to carry it
        acc[0][1][0] *= 5.9604644775390625e-08f;
      } else {
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float2 wv = *reinterpret_cast<const float2*>(s_win + a_off(t, jp));
          const float ar = xv[t][jp].x * wv.x, ai = xv[t][jp].y * wv.y;
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar, r_b[4 * jp + 0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar, r_b[4 * jp + 1], acc[t][1], 0, 0, 0);
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, r_b[4 * jp + 2], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, r_b[4 * jp + 3], acc[t][1], 0, 0, 0);
        }
      }
      }
      // the PCM of this wave's NEXT quad is requested as soon as the MFMAs have consumed this quad's: it lands while
      // the vector work of this quad runs
      __builtin_amdgcn_sched_barrier(0);
      cur = grab_value(ticket);
      if (cur < q_hi) issue_loads(cur);
      __builtin_amdgcn_sched_barrier(0);
      // ---- second pass in registers: lane (g, c) holds Y[k1][n2] of frame g ------------------------------
      float2 z[16];
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) {               // register i of tile t is the row (frame, i): n2 as dealt above
        const int t = V4 ? 2 * (n2 >> 3) + (n2 & 1) : n2 >> 2, i = V4 ? (n2 >> 1) & 3 : n2 & 3;
        z[n2] = make_float2(acc[t][0][i], acc[t][1][i]);
      }
#ifdef KWS_STFT_STAMP
      asm volatile("" :: "v"(z[0].x), "v"(z[15].y));   // the MFMA results have landed
#endif
      ST(0);
      if (!(KWS_STFT_ABL & 2)) {
#pragma unroll
      for (int n2 = 1; n2 < 16; ++n2) z[n2] = cmul(z[n2], s_tw[n2 * 16 + l16]);
      fft16(z);                                       // z[k2] = Z[k1 + 16 k2] / 2
      }
#ifdef KWS_STFT_STAMP
      asm volatile("" :: "v"(z[0].x), "v"(z[15].y));
#endif
      ST(1);
      // ---- real-input split + magnitude -------------------------------------------------------------
      // X[k] = E + T and X[256-k] = conj(E - T) with E = (Z[k] + conj Z[256-k]) / 2, T = W512^k (Z[k] - conj Z[256-k]) / 2i
      // (the halves are in z already).  Z[256-k] sits in the mirrored lane's register 15-k2; k1 = 0 (lane 0) and k1 = 8
      // (lane 15) are their own partners, with registers (16-k2)&15 and 15-k2.
      if (KWS_STFT_ABL & 4) {
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) KEEP2(z[n2]);
      } else {
      const float2 (&w5)[8] = r_w5;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        const float2 pa = z[15 - k2], pb = z[(16 - k2) & 15];
        const float mx = row_mirror(pa.x), my = row_mirror(pa.y);
        // lane-dependent source picked by bit masks: written as `l16 == 0 ? pb : ...` it compiled to two EXEC-masked
        // branches per bin pair
        const float2 zn0 = make_float2(blend(m0, pb.x, blend(m15, pa.x, mx)), blend(m0, pb.y, blend(m15, pa.y, my)));
        const float2 zk = z[k2];
        const float2 zn = make_float2(zn0.x, -zn0.y);
        const float2 E = cadd(zk, zn);
        const float2 dd = csub(zk, zn);
        const float2 O = make_float2(dd.y, -dd.x);
        const float2 T = cmul(w5[k2], O);
        const float2 Xp = cadd(E, T), Xm = csub(E, T);
        const int kk = k1 + 16 * k2;
        s_mag[kk] = __builtin_amdgcn_sqrtf(Xp.x * Xp.x + Xp.y * Xp.y);
        s_mag[256 - kk] = __builtin_amdgcn_sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);
      }
      if (l16 == 0) {                                 // bin 128: Z[128] pairs with itself, |X[128]| = |Z[128]|
        const float2 zk = z[8];
        s_mag[128] = 2.0f * __builtin_amdgcn_sqrtf(zk.x * zk.x + zk.y * zk.y);
      }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      ST(2);
      // ---- mel bands + log -> row 4 qq + fq of the group's log-mel block ------------------------------------------
      // Lane l16 takes bands l16 + 16 i.  Every band reads a window of 4 mel_mc[i] taps (mel_mc[i] = the widest band of
      // the 16 that share i, so the trip count is wave-uniform) that starts at mel_ws[m] and carries zero weights outside
      // the band: all of a band's LDS reads are in flight before its first multiply-add - the CSR walk of the third
      // kernel (a dependent wait per four taps, trip counts that differ lane by lane) took 40 % of a pass.
      float* lm_row = s_lm16 + (4 * qq + fq) * LMS;
#pragma unroll
      for (int i = 0; i < ((KWS_STFT_ABL & 8) ? 0 : NB); ++i) {
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
              if (NB >= 5) {                     // the windows start on even bins (stft.hip): 8-byte reads
                const float2 m01 = *reinterpret_cast<const float2*>(mp + 4 * (t0 + t));
                const float2 m23 = *reinterpret_cast<const float2*>(mp + 4 * (t0 + t) + 2);
                mv[t][0] = m01.x; mv[t][1] = m01.y; mv[t][2] = m23.x; mv[t][3] = m23.y;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) mv[t][r] = mp[4 * (t0 + t) + r];
              }
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
        // floor: a plain maximum (log_floor = 0 leaves the non-negative sum as it is) - no select on a scalar mask, which
        // issues at 1/7 of the rate of a v_max_f32 here (profiles/r02_probe_valu_rates.txt)
        const float sm = fmaxf(((s0 + s1) + (s2 + s3)) + pl.log_offset, pl.log_floor);
        // v_log_f32 (log2, 1 ulp) x ln 2: sm >= the offset / floor, so none of logf's denormal handling (15 instructions
        // per band) is needed; with the f16 DCT the row carries that product's 2^9 as well (exact)
        // lanes past n_mel (band 0's weights, see r_mws) store as well: their columns meet zero rows of the DCT operand, and
        // an unguarded store keeps the five bands straight-line code whose LDS reads overlap (guarded, each band's reads
        // were issued inside its own EXEC region, one round trip after the other)
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
    if (KWS_STFT_ABL & 16) {
      dacc[0][0] = pa[0];
    } else if (D16) {
      // as the first pass: the 20 (K = 80) x 4 f32 instructions of a group held the vector issue for 2,560 cycles; the f16 form
      // needs KB x 4 x 3 instructions that hold it for 8 each.  Log-mel values lie in [-14, 12]: scaled by 2^9.
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
      const float* pb = s_dct + fq * DSTR4 + l16;
    {
      // operands two steps ahead of the MFMAs that use them (one step = 4 MFMAs = 128 cycles of cover, an LDS round trip
      // under 12 waves takes longer)
      float av[3], bv[3][4];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        av[d] = pa[4 * d];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bv[d][nb] = pb[(4 * d) * DSTR4 + 16 * nb];
      }
#pragma unroll
      for (int ks = 0; ks < 16 * NB; ks += 4) {
        constexpr int KLAST = 16 * NB - 4;
        const int kn = ks + 8 <= KLAST ? ks + 8 : KLAST;          // clamped: the last two requests are not used
        av[2] = pa[kn];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bv[2][nb] = pb[kn * DSTR4 + 16 * nb];
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
              if (16 * nb + l16 < n_out && (!(KWS_STFT_ABL & 32) || dacc[nb][v] == 123.456f)) orow[16 * nb] = dacc[nb][v];
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

// instantiated band shapes (four-tap blocks per lane group): 80 mel bins over 257 bins (input_data.py:366-373 as train.py sets it,
// windows starting on even bins: 1, 2, 2, 3, 4), 40 mel bins (2, 5, 8; audio.py's 40 bands fit: 1, 3, 4), and the same lane counts
// with every group at the widest width; anything else declines (the generic kernel of stft.hip)
static int stft4_shape(const kws_stft_plan* pl) {
  if (pl->mel_maxw <= 0) return 0;
  const int nb = (pl->n_mel + 15) / 16;
  auto fits = [&](std::initializer_list<int> mc) {
    if ((int)mc.size() != nb) return false;
    int i = 0;
    for (int v : mc)
      if (pl->mel_mc[i++] > v) return false;
    return true;
  };
  if (fits({1, 2, 2, 3, 4})) return 1;
  if (fits({4, 4, 4, 4, 4})) return 2;
  if (fits({2, 5, 8})) return 3;
  if (fits({8, 8, 8})) return 4;
  return 0;
}

static int stft4_lds_bytes(const kws_stft_plan* pl, int nw, int gq) {
  const int sh = stft4_shape(pl);
  if (sh == 0) return 1 << 30;                                       // declines: the caller falls back to the generic kernel
  const int nb = sh <= 2 ? 5 : 3, mc = sh <= 2 ? 4 : 8;
  const size_t floats = 512 + 512 + 256 + 4 + (size_t)16 * nb * DSTR4 + (size_t)pl->n_mel * (4 * mc + 4) +
                        (size_t)nw * (4 * MAGF + ((4 * gq * (16 * nb + 1) + 3) & ~3));
  return (int)(floats * 4);
}
int kws_stft4_lds_bytes(const kws_stft_plan* pl) { return stft4_lds_bytes(pl, 12, 4); }   // the default form

template <int NB, int MC, int MCP, bool H1>
static int stft4_image_h(kws_stft_plan* pl) {
  const size_t bytes = (size_t)Stft4Lds<NB, MC>::image_floats(pl->n_mel) * sizeof(float);
  KWS_HIP(hipMalloc(reinterpret_cast<void**>(&pl->img4), bytes));
  hipLaunchKernelGGL((stft4_image_kernel<NB, MC, MCP, H1>), dim3(1), dim3(256), 0, nullptr, *pl, pl->img4);
  KWS_LAUNCH_CHECK("stft4_image_kernel");
  KWS_HIP(hipStreamSynchronize(nullptr));
  return KWS_OK;
}
template <int NB, int MC, int MCP>
static int stft4_image_t(kws_stft_plan* pl) { return stft4_image_h<NB, MC, MCP, true>(pl); }

// called once by kws_stft_plan_create after the tables are uploaded: the LDS image of this plan's kernel instance
int kws_stft4_prepare(kws_stft_plan* pl) {
  pl->img4 = nullptr;
  const int sh = stft4_shape(pl);
  if (sh == 0 || pl->n_mel % 4 != 0) return KWS_OK;                  // stft4 declines this plan: nothing to prepare
  if (sh == 1) return stft4_image_t<5, 4, 0x43221>(pl);
  if (sh == 2) return stft4_image_t<5, 4, 0x44444>(pl);
  if (sh == 3) return stft4_image_t<3, 8, 0x852>(pl);
  return stft4_image_t<3, 8, 0x888>(pl);
}

template <int NB, int MC, int MCP, bool H1, int NW4, int GQ, bool V4>
static int stft4_launch_w(const Stft2Args& a, hipStream_t st) {
  const int bytes = stft4_lds_bytes(&a.pl, NW4, GQ);
  KWS_REQUIRE(bytes <= 160 * 1024, "stft4: LDS need %d B exceeds 160 KiB", bytes);
  // per device and cheap: set on every launch (a process-wide "done" flag would miss the second device of a process)
  KWS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft4_kernel<NB, MC, MCP, H1, NW4, GQ, V4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int64_t wgs = (a.total_quads + NW4 - 1) / NW4;
  if (wgs > 256) wgs = 256;   // persistent: one workgroup per CU, tables copied once
  hipLaunchKernelGGL((stft4_kernel<NB, MC, MCP, H1, NW4, GQ, V4>), dim3((unsigned)wgs), dim3(NW4 * 64), (size_t)bytes, st, a);
  KWS_LAUNCH_CHECK("stft4_kernel");
  return KWS_OK;
}
// 12 waves per workgroup, DCT groups of 4 quads (16 x 2 was measured - 51.4 us against 50.3 - and is not instantiated);
// 16-byte PCM loads whenever every frame starts on a 16-byte boundary, 8-byte loads otherwise
template <int NB, int MC, int MCP>
static int stft4_launch_t(const Stft2Args& a, hipStream_t st) {
  const bool v4 = a.L % 4 == 0 && a.pl.frame_step % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  return v4 ? stft4_launch_w<NB, MC, MCP, true, 12, 4, true>(a, st) : stft4_launch_w<NB, MC, MCP, true, 12, 4, false>(a, st);
}

int kws_stft4_launch(const kws_stft_plan* pl, const float* x, int B, int L, int F, float* out, hipStream_t st) {
  KWS_REQUIRE(pl->n_out <= 64 && pl->n_mel % 4 == 0 && pl->n_mel <= 128, "stft4: n_mel=%d n_out=%d unsupported",
              pl->n_mel, pl->n_out);
  KWS_REQUIRE(F > 0 && (pl->frame_step % 2) == 0 && (L % 2) == 0 && (pl->frame_len % 2) == 0, "stft4: bad geometry");
  const int sh = stft4_shape(pl);
  KWS_REQUIRE(pl->img4 != nullptr, "stft4: the plan carries no table image (kws_stft4_prepare)");
  KWS_REQUIRE(sh != 0, "stft4: mel band shape (n_mel=%d, up to %d taps) is not instantiated", pl->n_mel, pl->mel_maxw);
  KWS_REQUIRE((int64_t)B * ((F + 3) / 4) < (1ll << 31), "stft4: %d clips x %d frames exceed 2^31 frame quads", B, F);
  Stft2Args a;
  a.pl = *pl;
  a.x = x; a.out = out; a.B = B; a.L = L; a.F = F;
  a.quads_per_clip = (F + 3) / 4;
  a.total_quads = (int64_t)B * a.quads_per_clip;
  if (sh == 1) return stft4_launch_t<5, 4, 0x43221>(a, st);
  if (sh == 2) return stft4_launch_t<5, 4, 0x44444>(a, st);
  if (sh == 3) return stft4_launch_t<3, 8, 0x852>(a, st);
  return stft4_launch_t<3, 8, 0x888>(a, st);
}
