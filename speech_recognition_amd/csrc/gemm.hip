// f32 MFMA GEMM family for the pointwise / first convolutions.
//
//   NN : C[M,N]  = A[M,K] * W[K,N]          forward (A = activations, optionally gathered) and
//                                            dgrad (A = dY, W = W^T)
//   TN : dW[K,N] = A^T[K,M] * G[M,N]        wgrad, split over M, partial slabs reduced in a fixed
//                                            order (bit-reproducible, no float atomics)
//
// gfx950 mapping: v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 cycles per SIMD).  Three kernels:
//   gemm_nn_ws_kernel / gemm_tn_ws_kernel  the default paths: ONE workgroup per CU (NN) or per work item (TN)
//       whose waves are specialised - MFMA waves, loader waves (buffer loads -> registers -> LDS) and, for
//       NN, storer waves (LDS staging tile -> global); see the comments at the kernels for the measurements
//       that shaped them;
//   gemm_nn_persist_kernel / gemm_tn_kernel  4-wave kernels (every wave loads, computes and stores) kept for
//       the gathered first convolution and ragged K / N.
// Common layout choices: A tiles row-major in LDS with a 4-float row pad so that one ds_read_b128 per lane
// fetches 4 consecutive k of its row conflict-free; the k order inside an 8-wide group is permuted
// consistently for A and B (lane half h owns k = 8q+4h+r), which the MFMA sum does not care about.
// -DKWS_GEMM_STAMP builds add s_memtime stamps to the wave-specialised NN kernel (scripts/stamps_ws.py).
#include "common.h"
#include "internal.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef KWS_GEMM_STAMP
__device__ unsigned long long g_stamps[8192][8];
extern "C" __attribute__((visibility("default"))) int kws_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif

namespace {

constexpr int NXCD = 8;

struct NNArgs {
  const float* A;
  const float* W;
  float* C;
  int64_t M;
  int K, N;
  float* stats;  // [m_tiles][2][N] or nullptr
  kws_gather_t g;
  int m_tiles, n_tiles;
  int half_tail;  // wave-specialised kernel: the tiles of a short last round are walked as 64-row halves
  int last_rows;  // rows of the last row tile: M - 128 (m_tiles - 1)
  unsigned inv_n_tiles;   // ceil(2^32 / n_tiles) (0 for n_tiles = 1): tile / n_tiles as one s_mul_hi_u32
  int lda;        // wave-specialised kernel only: row pitch of A in floats (0 = K).  Round 6: a 1 x 1 convolution with stride s over an
                  // even-length input is a plain GEMM over every s-th row - the shortcut convolutions of the residual nets
};

// 16 bytes of zeros that masked-out lanes load from instead of branching around their load.
// hipcc turns `ok ? *p : 0` into a branch per load and waits vmcnt(0) behind each one, which serialises
// a tile's global loads (measured: 2.3 k cycles to "issue" 8 loads); selecting the ADDRESS keeps every
// load unconditional and in flight together.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
  return *reinterpret_cast<const float4*>(ok ? p : g_zero16);
}

// gathered load of 4 consecutive elements starting at element index pos of one clip (0 outside
// [0, x_len)).  The common case (fully inside, 8-byte aligned: checked on the host) is two unconditional
// 8-byte loads; only rows that straddle the clip boundary take the element-wise branch.
__device__ __forceinline__ float4 gather4(const float* xb, int pos, int x_len) {
  const bool inside = pos >= 0 && pos + 3 < x_len;
  const float* p = inside ? xb + pos : g_zero16;
  const float2 lo = *reinterpret_cast<const float2*>(p);
  const float2 hi = *reinterpret_cast<const float2*>(p + 2);
  float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
  if (!inside && pos + 3 >= 0 && pos < x_len) {
    if (pos >= 0 && pos < x_len) v.x = xb[pos];
    if (pos + 1 >= 0 && pos + 1 < x_len) v.y = xb[pos + 1];
    if (pos + 2 >= 0 && pos + 2 < x_len) v.z = xb[pos + 2];
    if (pos + 3 >= 0 && pos + 3 < x_len) v.w = xb[pos + 3];
  }
  return v;
}

// ------------------------------------------------------------------------------------------------
// Persistent 4-wave NN kernel (the gathered first convolution and ragged K / N; the wave-specialised kernel
// below takes everything else).  One-tile-per-workgroup kernels showed, in in-kernel stamps, (6.1 k cycles exposed first-load
// latency + 9.4 k cycles of 4-byte-per-lane stores per 16.4 k cycles of MFMA work, and co-resident
// workgroups in lockstep so none of it overlapped):
//   * each workgroup walks a list of tiles; while it computes the LAST K-slab of tile t it already has
//     the first K-slab of tile t+1 in flight (register prefetch), so the load latency hides under MFMA;
//   * the accumulators leave through LDS: every wave stages its 64 x (TN*32) sub-tile in the (now idle)
//     pipeline buffers and writes it out as 16-byte row-major stores - 4x fewer store instructions and
//     256-byte contiguous segments instead of 128-byte ones; the stores drain while the next tile's
//     main loop runs.
constexpr int PBK = 32;
constexpr int PLDA = PBK + 4;

template <int BM, int BN, int WM, int WN, bool GATHER, bool STATS>
__global__ __launch_bounds__(256, 2) void gemm_nn_persist_kernel(NNArgs p) {
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_F4 = BM * PBK / 4 / 256;
  constexpr int B_F4 = PBK * BN / 4 / 256;
  constexpr int BN4 = BN / 4;
  constexpr int PBK4 = PBK / 4;
  constexpr int STAGE = BM * PLDA + PBK * BN;       // floats per pipeline stage
  constexpr int WROWS = TM * 32, WCOLS = TN * 32;   // one wave's sub-tile
  constexpr int C4_PER_ROW = WCOLS / 4;             // float4 per staged row
  constexpr int RED_OFF = 4 * WROWS * WCOLS;        // stats scratch behind the 4 staging regions
  static_assert(WM * WN == 4, "4 waves");
  static_assert(RED_OFF + 2 * WM * BN <= 2 * STAGE, "staging + stats scratch must fit the pipeline LDS");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;
  const int nk = (K + PBK - 1) / PBK;

  // XCD-aware persistent tile walk: workgroups b and b+8 share an XCD (round-robin dispatch, speed only);
  // XCD x owns the row panels tile_m = x (mod 8) and its workgroups take consecutive local tiles, so the
  // n-tiles of one row panel run at the same time on one L2.
  const int xcd = blockIdx.x % NXCD;
  const int wg_in_xcd = blockIdx.x / NXCD;
  const int wgs_per_xcd = gridDim.x / NXCD;
  const int panels = (p.m_tiles - xcd + NXCD - 1) / NXCD;   // row panels owned by this XCD
  const int local_tiles = panels * p.n_tiles;
  int local = wg_in_xcd;
  if (local >= local_tiles) return;

  const float* a_row[A_F4];
  bool a_ok[A_F4];
  int a_t[A_F4];
  int64_t m0 = 0;
  int n0 = 0, tile_m = 0;
  auto setup = [&](int loc) {
    tile_m = (loc / p.n_tiles) * NXCD + xcd;
    m0 = (int64_t)tile_m * BM;
    n0 = (loc % p.n_tiles) * BN;
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256;
      const int64_t gm = m0 + idx / PBK4;
      a_ok[r] = gm < M;
      if (GATHER) {
        const int64_t b = a_ok[r] ? gm / p.g.L_out : 0;
        a_t[r] = a_ok[r] ? (int)(gm - b * p.g.L_out) : 0;
        a_row[r] = p.A + b * p.g.x_batch_stride;
      } else {
        a_t[r] = 0;
        a_row[r] = p.A + (a_ok[r] ? gm : 0) * (int64_t)K;
      }
    }
  };
  float4 ra[A_F4], rb[B_F4];
  auto load_global = [&](int k0) {
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256;
      const int gk = k0 + (idx % PBK4) * 4;
      if (GATHER) {
        if (a_ok[r] && gk < K) {
          const int j = gk / p.g.cin;
          const int c = gk - j * p.g.cin;
          ra[r] = gather4(a_row[r], a_t[r] * p.g.stride_t + j * p.g.stride_j + c + p.g.base_off, p.g.x_len);
        } else {
          ra[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        ra[r] = ld4_or_zero(a_row[r] + gk, a_ok[r] && gk < K);
      }
    }
#pragma unroll
    for (int r = 0; r < B_F4; ++r) {
      const int idx = tid + r * 256;
      const int gk = k0 + idx / BN4, gn = n0 + (idx % BN4) * 4;
      rb[r] = ld4_or_zero(p.W + (int64_t)gk * N + gn, gk < K && gn < N);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + (idx / PBK4) * PLDA + (idx % PBK4) * 4]) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < B_F4; ++r) {
      const int idx = tid + r * 256;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + BM * PLDA + (idx / BN4) * BN + (idx % BN4) * 4]) = rb[r];
    }
  };

  f32x16 acc[TM][TN];
  setup(local);
  load_global(0);
  store_lds(0);
  __syncthreads();
  while (true) {
    const int64_t cm0 = m0;   // the tile being computed (setup() for the prefetch overwrites m0/n0/tile_m)
    const int cn0 = n0, ctile_m = tile_m;
    const int next = local + wgs_per_xcd;
    const bool has_next = next < local_tiles;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) {
        load_global((kt + 1) * PBK);
      } else if (has_next) {
        setup(next);        // first K-slab of the NEXT tile flies while this tile's last slab is computed
        load_global(0);
      }
      const float* cA = smem + cur * STAGE + (wm * TM * 32 + li) * PLDA + lh * 4;
      const float* cB = smem + cur * STAGE + BM * PLDA + (lh * 4) * BN + wn * TN * 32 + li;
#pragma unroll
      for (int q = 0; q < PBK / 8; ++q) {
        float4 a[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(cA + i * 32 * PLDA + q * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float b[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j] = cB[(q * 8 + r) * BN + j * 32];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float av = r == 0 ? a[i].x : (r == 1 ? a[i].y : (r == 2 ? a[i].z : a[i].w));
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j], acc[i][j], 0, 0, 0);
          }
        }
      }
      if (kt + 1 < nk) store_lds(cur ^ 1);
      __syncthreads();   // after the last slab: every wave is done reading the pipeline buffers
    }

    // ---- epilogue: registers -> this wave's LDS staging region -> 16-byte global stores --------------
    float* stg = smem + wave * (WROWS * WCOLS);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v)
          stg[(i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh) * WCOLS + j * 32 + li] = acc[i][j][v];
    if (STATS) {
      float* red = smem + RED_OFF;  // [2][WM][BN]
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float x = acc[i][j][v];
            s += x;
            ss += x * x;
          }
        s += __shfl_xor(s, 32);
        ss += __shfl_xor(ss, 32);
        if (lh == 0) {
          const int c = wn * WCOLS + j * 32 + li;
          red[(0 * WM + wm) * BN + c] = s;
          red[(1 * WM + wm) * BN + c] = ss;
        }
      }
    }
    // same-wave LDS accesses are ordered; only the compiler has to be kept from reordering them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const int64_t row_base = cm0 + wm * WROWS;
      const int col_base = cn0 + wn * WCOLS;
      const int c4 = lane % C4_PER_ROW, r_in = lane / C4_PER_ROW;
      constexpr int ROWS_PER_PASS = 64 / C4_PER_ROW;
#pragma unroll
      for (int ps = 0; ps < WROWS / ROWS_PER_PASS; ++ps) {
        const int r = ps * ROWS_PER_PASS + r_in;
        const float4 v = *reinterpret_cast<const float4*>(stg + r * WCOLS + c4 * 4);
        const int64_t row = row_base + r;
        const int col = col_base + c4 * 4;
        if (row < M && col < N) *reinterpret_cast<float4*>(p.C + row * N + col) = v;
      }
    }
    if (STATS) {
      __syncthreads();
      const float* red = smem + RED_OFF;
      if (tid < BN && cn0 + tid < N) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s += red[(0 * WM + w) * BN + tid];
          ss += red[(1 * WM + w) * BN + tid];
        }
        p.stats[((int64_t)ctile_m * 2 + 0) * N + cn0 + tid] = s;
        p.stats[((int64_t)ctile_m * 2 + 1) * N + cn0 + tid] = ss;
      }
    }
    if (!has_next) break;
    __syncthreads();   // staging / stats scratch fully consumed before the pipeline buffers are refilled
    store_lds(0);
    __syncthreads();
    local = next;
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised NN kernel.  The stamps of the persistent kernel show that one wave cannot overlap its
// own memory phases with MFMA issue (per 128x128 tile: 15.8 k cycles MFMA, 3.7 k issuing loads, 2.4 k
// waiting for them, 5.5 k moving C out), and co-resident workgroups drift into lockstep.  Here the roles
// are split inside ONE workgroup per CU:
//   4 compute waves: MFMA only - read the A/B slab of iteration g from LDS slot g&1 with the fragments
//              of the next k-group prefetched under the current MFMAs.  The matrix pipe queues ~48 MFMAs,
//              so a wave finishes ISSUING an iteration ~1.5 k cycles before the pipe finishes executing
//              it; that slack pays for the epilogue: tiles alternate between two accumulator sets and the
//              finished set is written to the LDS staging tile after the FIRST iteration of the next tile
//              has been issued, so the pipe never drains.  The MFMA operands are swapped
//              (D = W^T-fragment x A-fragment) so that a lane's 4 consecutive accumulator registers are 4
//              consecutive COLUMNS of one C row: 16 ds_write_b128 per lane instead of 64 ds_write_b32;
//   NLW loader waves: wave w owns the K-slabs s = w (mod NLW); it writes slab s into slot s&1 during
//              iteration s-1 and immediately re-issues its loads for slab s+NLW, so the loads have almost
//              NLW full iterations to land and every wait is a plain vmcnt(0) on the wave's own register
//              set.  Loads are buffer loads: the tile base and K offset live in SGPRs (no per-load VALU
//              address math) and rows past M read as zeros through the descriptor's range check;
//   NSW storer waves: move the staging tile to global memory as 16-byte row stores (buffer stores, rows
//              past M dropped by the range check) in nk-1 equal chunks spread over the next tile's
//              iterations (a VALU instruction on a SIMD whose matrix pipe is saturated costs ~60 cycles,
//              and all CUs finish tiles together, so one burst per tile stalls the barrier), and
//              accumulate the BN column sums on the way (fixed order).
// One __syncthreads per iteration orders everything: slot (g+1)&1 was last read in iteration g-1; the
// staging tile is written at the end of iteration kt=0 of a tile, read during kt=1..nk-1, and its column
// sums are finished at kt=0 of the following tile.
// Host-checked preconditions: K % 32 == 0, K >= 64, N % BN == 0, 32-bit byte offsets inside a tile view.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// -DKWS_WS_ABL=<bits> (timing only, wrong results): the kernel without 1 the storers' row moves, 2 the loaders' LDS writes,
// 4 the loaders' global loads, 8 the MFMA waves' staging writes; 16 = WITH five dependent vector instructions per A element
// in the loader waves (what a depthwise -> pointwise loader fusion would issue there: BN scale/shift + ReLU6 on the loaded
// y element, three FIR taps)
#ifndef KWS_WS_ABL
#define KWS_WS_ABL 0
#endif
// -DKWS_WS_PRIO_MMA=<0..3> / -DKWS_WS_PRIO_LD=<0..3> / -DKWS_WS_PRIO_ST=<0..3>: s_setprio of the MFMA / loader / storer waves
// (round 4 experiment: a SIMD hosts one MFMA wave and one mover; the default leaves all at 0)
#ifndef KWS_WS_PRIO_MMA
#define KWS_WS_PRIO_MMA 0
#endif
#ifndef KWS_WS_PRIO_LD
#define KWS_WS_PRIO_LD 0
#endif
#ifndef KWS_WS_PRIO_ST
#define KWS_WS_PRIO_ST 0
#endif
constexpr int KWS_WS_MAX_N = 1024;                  // widest N the per-workgroup statistics row supports
constexpr int KWS_BUFFER_RSRC_FLAGS = 0x00020000;   // raw buffer, 32-bit data format (gfx9 family)

__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// KB = K-slab depth per barrier: 32 for the 128-wide column tiles; the 64-wide tiles (N = 192, 320) take 64 so
// that an MFMA wave still issues 64 MFMAs per barrier (LDS: 2 x 50.8 KB slots + 34.8 KB staging).
// LDS floats of the kernel below (the same arithmetic as its layout constants)
template <int BN, int KB, bool STATS>
constexpr int nn_ws_smem_floats() {
  return 2 * (128 * (KB + 4) + KB * BN) + 128 * (BN + 4) + (256 / (BN / 4)) * 2 * BN + (STATS ? 2 * KWS_WS_MAX_N : 0);
}
// The kernel as a device function of (block id, number of blocks): gemm_nn_ws_kernel runs it on its own grid,
// gemm_dgrad_wgrad_kernel (further down) on the first blocks of a grid that also holds a weight-gradient GEMM.
template <int BN, int KB, int NLW, int NSW, bool STATS>
__device__ __forceinline__ void nn_ws_body(const NNArgs& p, float* const smem, const int bid, const int nblk) {
  constexpr int PBK = KB, PLDA = KB + 4;            // shadow the 32-deep constants of the 4-wave kernels
  constexpr int NQ = KB / 8;                        // 8-deep k-groups per slab
  constexpr int BM = 128, WM = 2, WN = 2;
  constexpr int NCT = WM * WN * 64;                 // compute threads
  constexpr int NLT = NLW * 64;                     // loader threads
  constexpr int NST = NSW * 64;                     // storer threads
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_F4 = BM * PBK / 4 / 64;           // 16 per loader lane
  constexpr int B_F4 = PBK * BN / 4 / 64;           // 16 or 8
  constexpr int BN4 = BN / 4;
  constexpr int PBK4 = PBK / 4;
  constexpr int STAGE = BM * PLDA + PBK * BN;
  constexpr int SLD = BN + 4;                       // staging row stride (floats): conflict-free b128 writes
  constexpr int STG_OFF = 2 * STAGE;
  constexpr int RGROUPS = NST / BN4;                // storer row groups
  constexpr int NPASS = BM / RGROUPS;               // row passes per tile and storer thread
  constexpr int RED_OFF = STG_OFF + BM * SLD;
  constexpr int WACC_OFF = RED_OFF + (NCT / BN4) * 2 * BN;   // [2][KWS_WS_MAX_N] column sums of ALL my tiles
  constexpr int SMEM = WACC_OFF + (STATS ? 2 * KWS_WS_MAX_N : 0);
  static_assert(SMEM * 4 <= 160 * 1024, "LDS budget");
  static_assert(SMEM == nn_ws_smem_floats<BN, KB, STATS>(), "nn_ws_smem_floats out of step with the layout");

  const int tid = threadIdx.x;
  const int K = p.K, N = p.N;
  const int nk = K / PBK;                           // >= 2

  const int xcd = bid % NXCD;
  const int wg_in_xcd = bid / NXCD;
  const int wgs_per_xcd = nblk / NXCD;
  const int panels = (p.m_tiles - xcd + NXCD - 1) / NXCD;
  const int local_tiles = panels * p.n_tiles;
  // The walk of an XCD's workgroups over its tiles ends with a partial round: e_x of the wgs_per_xcd workgroups would
  // compute one more 128-row tile while the others wait for the kernel to end (6.06 tiles per CU cost 7 rounds).  When
  // the leftover fits twice (2 e_x <= wgs_per_xcd) those tiles are walked as 64-row HALVES by twice as many workgroups:
  // a half tile is the same tile view cut at 64 rows (the descriptors' range checks zero-fill / drop the rest) whose MFMA
  // waves skip their second row block - rows are dealt to the waves as 64 i + 32 wm + lane so that every wave owns one block
  // of each half.  Each output element is still one wave's fmaf chain over k in the same order: bit-identical results.
  const int full_rounds = local_tiles / wgs_per_xcd;
  const int e_x = local_tiles - full_rounds * wgs_per_xcd;
  const bool halves = p.half_tail && e_x > 0 && 2 * e_x <= wgs_per_xcd;
  const int first_half = halves ? full_rounds * wgs_per_xcd : local_tiles;   // local index of the first half tile
  const int local_count = halves ? first_half + 2 * e_x : local_tiles;
  if (wg_in_xcd >= local_count) {                   // (only when local_tiles < wgs_per_xcd: unreachable with halves)
    if (STATS)
      for (int c = tid; c < 2 * N; c += (4 + NLW + NSW) * 64) p.stats[(int64_t)bid * 2 * N + c] = 0.f;
    return;
  }
  // local item -> tile view: row tile, first row inside it (0 / 64), rows it holds, first column.  Everything here is 32-bit
  // scalar arithmetic on purpose: the loader waves run it in front of every slab's loads, and a first version that compared
  // 64-bit row counts (v_cmp_*_i64: there is no scalar form) and selected 64-bit row offsets cost the 64-wide-tile kernels
  // 1,800 cycles of barrier wait per iteration - a vector instruction of a non-MFMA wave waits for the matrix pipe's gaps
  auto decode = [&](int loc, int& tile_m, int& row0, int& rows, int& n0) {
    const int h = loc - first_half;                 // >= 0: a half tile
    const int tile = h >= 0 ? first_half + (h >> 1) : loc;
    // tile / n_tiles by the host's reciprocal (exact for tile < 2^32 / n_tiles): the compiler's integer division goes
    // through v_rcp_iflag_f32 - five dependent vector instructions in front of the loads
    const int tq = p.inv_n_tiles ? (int)__umulhi((unsigned)tile, p.inv_n_tiles) : tile;
    tile_m = tq * NXCD + xcd;
    n0 = (tile - tq * p.n_tiles) * BN;
    row0 = h >= 0 ? 64 * (h & 1) : 0;
    const int cap = h >= 0 ? 64 : BM;
    int r = (tile_m < p.m_tiles - 1 ? BM : p.last_rows) - row0;   // rows of the tile from row0 on
    r = r < 0 ? 0 : r;
    rows = r < cap ? r : cap;
  };
  const int n_my = (local_count - wg_in_xcd + wgs_per_xcd - 1) / wgs_per_xcd;
  const int G = n_my * nk;
  // barriers executed by every wave: 1 (prologue) + G (iterations) + 2 (last tile staged / moved out)
  // + 1 when STATS (column sums complete)

  if (tid < NCT) {
    // ------------------------------------------------------------------ MFMA waves
    if (KWS_WS_PRIO_MMA) __builtin_amdgcn_s_setprio(KWS_WS_PRIO_MMA);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    struct Frag {
      float4 a[TM];
      float b[4][TN];
    };
    struct Acc {
      f32x16 t[TM][TN];
    };
    Acc accA, accB;
    // C[m][n]: m = i*64 + wm*32 + li, n = wn*TN*32 + j*32 + 8*(v>>2) + 4*lh + (v&3)
    static_assert(TM == 2 && WM == 2, "row blocks are dealt as 64 i + 32 wm");
    float* const stg = smem + STG_OFF + (wm * 32 + li) * SLD + wn * TN * 32 + 4 * lh;
    auto stage = [&](const Acc& c) {
      if (KWS_WS_ABL & 8) { asm volatile("" :: "v"(c.t[0][0][0])); return; }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v4 = 0; v4 < 4; ++v4)
            *reinterpret_cast<float4*>(stg + i * 64 * SLD + j * 32 + 8 * v4) =
                make_float4(c.t[i][j][4 * v4], c.t[i][j][4 * v4 + 1], c.t[i][j][4 * v4 + 2], c.t[i][j][4 * v4 + 3]);
    };
    // BN column sums of a staged tile: VALU work is only cheap inside the MFMA waves' own instruction
    // stream (a VALU instruction of another wave waits ~60 cycles for an issue slot while the matrix pipe
    // is saturated), so it lives here, in the slack after an iteration's MFMAs have been issued.
    constexpr int CRG = NCT / BN4;                  // row groups of the column-sum pass
    const int sc4 = tid % BN4, srg = tid / BN4;
    int st_n0 = 0;                                  // first column of the staged tile
    constexpr int SPASS = BM / CRG, SHALF = SPASS / 2;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), css = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* const sp = smem + STG_OFF + srg * SLD + sc4 * 4;
    // the LDS reads of half a tile are issued BEFORE an iteration's MFMAs and consumed after them, so their
    // latency never shows; rows >= M hold exact zeros (their A rows were loaded as zeros)
    auto stats_read = [&](float4 (&sv)[SHALF], int half) {
#pragma unroll
      for (int ps = 0; ps < SHALF; ++ps) sv[ps] = *reinterpret_cast<const float4*>(sp + (half * SHALF + ps) * CRG * SLD);
    };
    auto stats_add = [&](const float4 (&sv)[SHALF], int half) {
      if (half == 0) {
        cs = make_float4(0.f, 0.f, 0.f, 0.f);
        css = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int ps = 0; ps < SHALF; ++ps) {
        const float4 v = sv[ps];
        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        css.x += v.x * v.x; css.y += v.y * v.y; css.z += v.z * v.z; css.w += v.w * v.w;
      }
      if (half == 1) {
        float* red = smem + RED_OFF + (srg * 2) * BN + sc4 * 4;   // [CRG][2][BN]
        *reinterpret_cast<float4*>(red) = cs;
        *reinterpret_cast<float4*>(red + BN) = css;
      }
    };
    auto stats_partial = [&]() {
      float4 sv[SHALF];
      stats_read(sv, 0);
      stats_add(sv, 0);
      stats_read(sv, 1);
      stats_add(sv, 1);
    };
    auto stats_finish = [&]() {                     // >= one barrier after stats_partial: fixed-order sum
      if (tid < BN) {
        const float* red = smem + RED_OFF + tid;
        float cs = 0.f, css = 0.f;
#pragma unroll
        for (int w = 0; w < CRG; ++w) {
          cs += red[(w * 2 + 0) * BN];
          css += red[(w * 2 + 1) * BN];
        }
        // the workgroup's tiles are folded in tile order into ONE statistics row per workgroup (<= 256
        // rows for kws_bn_stats_finalize instead of one per 128-row tile); column st_n0 + tid is always
        // updated by this same thread
        smem[WACC_OFF + st_n0 + tid] += cs;
        smem[WACC_OFF + KWS_WS_MAX_N + st_n0 + tid] += css;
      }
    };
    auto set_staged_tile = [&](int ordinal) {
      int tile_m, row0, rows;
      decode(wg_in_xcd + ordinal * wgs_per_xcd, tile_m, row0, rows, st_n0);
    };
    if (STATS)
      for (int c = tid; c < 2 * KWS_WS_MAX_N; c += NCT) smem[WACC_OFF + c] = 0.f;
    __syncthreads();
#ifdef KWS_GEMM_STAMP
    unsigned long long t_mma = 0, t_bar = 0, t_stage = 0, t_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_mark, r_begin = __builtin_amdgcn_s_memrealtime();
#define WT(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - t_mark; t_mark = now_; } while (0)
#else
#define WT(acc_)
#endif
    int g = 0, tile_ord = 0;
    auto run_tile_t = [&](auto half_c, Acc& acc, const Acc& prev, bool have_prev) {
      constexpr int TMR = decltype(half_c)::value ? 1 : TM;   // row blocks this tile computes
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc.t[i][j][v] = 0.f;
      for (int kt = 0; kt < nk; ++kt, ++g) {
        const int cur = g & 1;
        const float* cA = smem + cur * STAGE + (wm * 32 + li) * PLDA + lh * 4;
        const float* cB = smem + cur * STAGE + BM * PLDA + (lh * 4) * BN + wn * TN * 32 + li;
        auto load_frag = [&](Frag& f, int q) {
#pragma unroll
          for (int i = 0; i < TMR; ++i) f.a[i] = *reinterpret_cast<const float4*>(cA + i * 64 * PLDA + q * 8);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < TN; ++j) f.b[r][j] = cB[(q * 8 + r) * BN + j * 32];
        };
        auto mma = [&](const Frag& f) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < TMR; ++i) {
              const float av = r == 0 ? f.a[i].x : (r == 1 ? f.a[i].y : (r == 2 ? f.a[i].z : f.a[i].w));
#pragma unroll
              for (int j = 0; j < TN; ++j)   // swapped operands: lane <-> C row, register <-> C column
                acc.t[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[r][j], av, acc.t[i][j], 0, 0, 0);
            }
        };
        // column sums of the staged tile: first half at kt = 1, second half at kt = 2 (both at kt = 1 when nk = 2)
        const bool st_a = STATS && have_prev && kt == 1, st_b = STATS && have_prev && kt == 2;
        float4 sv[SHALF];
        if (st_a) stats_read(sv, 0);
        if (st_b) stats_read(sv, 1);
        Frag f0, f1;
        load_frag(f0, 0);
        load_frag(f1, 1);
#pragma unroll
        for (int q = 0; q < NQ; q += 2) {           // fragments of k-group q+2 / q+3 fly under the MFMAs of q / q+1
          __builtin_amdgcn_sched_barrier(0);
          mma(f0);
          __builtin_amdgcn_sched_barrier(0);
          if (q + 2 < NQ) load_frag(f0, q + 2);
          __builtin_amdgcn_sched_barrier(0);
          mma(f1);
          __builtin_amdgcn_sched_barrier(0);
          if (q + 3 < NQ) load_frag(f1, q + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        WT(t_mma);
        if (have_prev) {
          if (kt == 0) {
            if (STATS && tile_ord >= 2) stats_finish();   // tile tile_ord-2: partial sums written one tile ago
            stage(prev);        // previous tile's accumulators: finished long ago, pipe stays fed by this tile
            set_staged_tile(tile_ord - 1);
            WT(t_stage);
          } else if (st_a) {
            stats_add(sv, 0);
            if (nk == 2) {
              stats_read(sv, 1);
              stats_add(sv, 1);
            }
            WT(t_stage);
          } else if (st_b) {
            stats_add(sv, 1);
            WT(t_stage);
          }
        }
        __syncthreads();
        WT(t_bar);
      }
      ++tile_ord;
    };
    auto run_tile = [&](Acc& acc, const Acc& prev, bool have_prev) {
      if (wg_in_xcd + tile_ord * wgs_per_xcd >= first_half) run_tile_t(std::true_type(), acc, prev, have_prev);
      else run_tile_t(std::false_type(), acc, prev, have_prev);
    };
    for (int t = 0; t < n_my; t += 2) {
      run_tile(accA, accB, t > 0);
      if (t + 1 < n_my) run_tile(accB, accA, true);
    }
    if (STATS && n_my >= 2) stats_finish();         // tile n_my-2
    if (n_my & 1) stage(accA); else stage(accB);
    set_staged_tile(n_my - 1);
#ifdef KWS_GEMM_STAMP
    if ((tid & 63) == 0 && bid < 256) g_stamps[bid + 512][tid >> 6] = t_bar;   // every wave's barrier wait
    if (tid == 0 && bid < 8192) {
      g_stamps[bid][0] = t_mma; g_stamps[bid][1] = t_bar; g_stamps[bid][2] = t_stage;
      g_stamps[bid][3] = (unsigned long long)G;
      g_stamps[bid][4] = __builtin_amdgcn_s_memtime() - t_begin;
      g_stamps[bid][5] = __builtin_amdgcn_s_memrealtime() - r_begin;
    }
#endif
    __syncthreads();   // last tile staged
    if (STATS) stats_partial();
    __syncthreads();   // last tile moved out, its partial sums in LDS
    if (STATS) {
      stats_finish();
      __syncthreads();
      for (int c = tid; c < N; c += NCT) {
        p.stats[((int64_t)bid * 2 + 0) * N + c] = smem[WACC_OFF + c];
        p.stats[((int64_t)bid * 2 + 1) * N + c] = smem[WACC_OFF + KWS_WS_MAX_N + c];
      }
    }
  } else if (tid < NCT + NLT) {
    // ------------------------------------------------------------------ loader waves
    if (KWS_WS_PRIO_LD) __builtin_amdgcn_s_setprio(KWS_WS_PRIO_LD);
    const int lane = tid & 63;
    const int lw = __builtin_amdgcn_readfirstlane((tid - NCT) >> 6);   // my slabs: s = lw (mod NLW)
    constexpr int AROWS = 64 / PBK4;                // A rows covered by one wave instruction: 8 or 4
    const int arow = lane / PBK4;                   // + AROWS r
    const int acol = (lane % PBK4) * 4;
    const int brow = lane / BN4;                    // + BROWS r
    const int bcol = (lane % BN4) * 4;
    constexpr int BROWS = 64 / BN4;                 // B rows covered by one wave instruction
    const int lda = p.lda ? p.lda : K;              // row pitch of A (a strided row view: every s-th row of a wider matrix)
    const int a_voff = (arow * lda + acol) * 4;     // byte offsets inside the tile's buffer views
    const int b_voff = (brow * N + bcol) * 4;
    int ld_i = lw / nk, ld_kt = lw % nk;            // my next slab: tile ordinal, K-slab
    float4 ra[A_F4], rb[B_F4];
    auto issue = [&]() {
      const bool tile_ok = ld_i < n_my;
      int tile_m, row0, rows_t, n0;
      decode(wg_in_xcd + ld_i * wgs_per_xcd, tile_m, row0, rows_t, n0);
      const int64_t m0 = (int64_t)tile_m * BM + row0;
      const int rows = tile_ok ? rows_t : 0;
      const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.A + m0 * lda), 0, rows > 0 ? ((rows - 1) * lda + K) * 4 : 0, KWS_BUFFER_RSRC_FLAGS);
      const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.W + n0), 0, tile_ok ? (K * N - n0) * 4 : 0, KWS_BUFFER_RSRC_FLAGS);
      const int k0 = ld_kt * PBK;
      if (!(KWS_WS_ABL & 4)) {
#pragma unroll
      for (int r = 0; r < A_F4; ++r) ra[r] = buf_ld4(ares, a_voff, (k0 + AROWS * r * lda) * 4);
#pragma unroll
      for (int r = 0; r < B_F4; ++r) rb[r] = buf_ld4(bres, b_voff, (k0 + BROWS * r) * N * 4);
      }
      ld_kt += NLW;
      while (ld_kt >= nk) {
        ld_kt -= nk;
        ++ld_i;
      }
    };
    auto write_lds = [&](int slot) {
      if (KWS_WS_ABL & 2) { asm volatile("" :: "v"(ra[0].x), "v"(rb[0].x)); return; }
      if (KWS_WS_ABL & 16) {
#pragma unroll
        for (int r = 0; r < A_F4; ++r) {
          float* e = &ra[r].x;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int o = 0; o < 5; ++o) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(e[c]) : "v"(rb[0].x));
        }
      }
      float* sA = smem + slot * STAGE + arow * PLDA + acol;
#pragma unroll
      for (int r = 0; r < A_F4; ++r) *reinterpret_cast<float4*>(sA + AROWS * r * PLDA) = ra[r];
      float* sB = smem + slot * STAGE + BM * PLDA + brow * BN + bcol;
#pragma unroll
      for (int r = 0; r < B_F4; ++r) *reinterpret_cast<float4*>(sB + BROWS * r * BN) = rb[r];
    };
    issue();                     // slab lw
    if (lw == 0) {
      write_lds(0);
      issue();                   // slab NLW
    }
    __syncthreads();
#ifdef KWS_GEMM_STAMP
    unsigned long long t_write = 0, t_issue = 0, t_lbar = 0, t_mark = 0;
#endif
    for (int g = 0; g < G; ++g) {
      if ((g + 1) % NLW == lw) {
#ifdef KWS_GEMM_STAMP
        t_mark = __builtin_amdgcn_s_memtime();
#endif
        write_lds((g + 1) & 1);  // slab g+1 (zeros past the end)
        WT(t_write);
        issue();                 // slab g+1+NLW
        WT(t_issue);
      }
#ifdef KWS_GEMM_STAMP
      t_mark = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
      WT(t_lbar);
    }
#ifdef KWS_GEMM_STAMP
    if (tid == NCT && bid < 8192) {
      g_stamps[bid][6] = t_write; g_stamps[bid][7] = t_issue;
      g_stamps[bid + 256][2] = t_lbar;
    }
    if ((tid & 63) == 0 && bid < 256) g_stamps[bid + 512][tid >> 6] = t_lbar;
#endif
    __syncthreads();
    __syncthreads();
    if (STATS) __syncthreads();
  } else {
    // ------------------------------------------------------------------ storer waves
    // pure data movers (LDS read + buffer store, one address add per pass): the tile staged at kt = 0 is
    // moved out in nk-1 equal chunks at kt = 1..nk-1
    if (KWS_WS_PRIO_ST) __builtin_amdgcn_s_setprio(KWS_WS_PRIO_ST);
    const int stt = tid - NCT - NLT;
    const int c4 = stt % BN4, r_in = stt / BN4;
    const int c_voff = (r_in * N + c4 * 4) * 4;
    const int ppi = (NPASS + nk - 2) / (nk - 1);    // passes per iteration
    int st_i = 0;                                   // ordinal of the next tile to move out
    __amdgpu_buffer_rsrc_t cres = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0, KWS_BUFFER_RSRC_FLAGS);
    auto begin_tile = [&]() {
      int tile_m, row0, rows, n0;
      decode(wg_in_xcd + st_i * wgs_per_xcd, tile_m, row0, rows, n0);
      const int64_t m0 = (int64_t)tile_m * BM + row0;
      // view of C starting at (m0, n0): rows past M fall outside rows*N floats and are dropped
      cres = __builtin_amdgcn_make_buffer_rsrc(p.C + m0 * N + n0, 0, rows > 0 ? (rows * N - n0) * 4 : 0,
                                               KWS_BUFFER_RSRC_FLAGS);
      ++st_i;
    };
    auto move_rows = [&](int lo, int hi) {          // passes [lo, hi) of the staged tile
      if (KWS_WS_ABL & 1) return;
      const float* stg = smem + STG_OFF + c4 * 4 + r_in * SLD;
      constexpr int GRP = 8;                        // all LDS reads of a group are in flight before its first store
      for (int p0 = lo; p0 < hi; p0 += GRP) {
        float4 v[GRP];
#pragma unroll
        for (int i = 0; i < GRP; ++i)
          if (p0 + i < hi) v[i] = *reinterpret_cast<const float4*>(stg + (p0 + i) * RGROUPS * SLD);
#pragma unroll
        for (int i = 0; i < GRP; ++i)
          if (p0 + i < hi) {
            u32x4 u;
            u.x = __float_as_uint(v[i].x); u.y = __float_as_uint(v[i].y); u.z = __float_as_uint(v[i].z); u.w = __float_as_uint(v[i].w);
            __builtin_amdgcn_raw_buffer_store_b128(u, cres, c_voff, (p0 + i) * RGROUPS * N * 4, 0);
          }
      }
    };
    __syncthreads();
#ifdef KWS_GEMM_STAMP
    unsigned long long t_store = 0, t_sbar = 0, t_mark = 0;
#endif
    int kt = 0;
    for (int g = 0; g < G; ++g) {
      if (g >= nk) {                                // a previous tile exists
        if (kt == 0) {
          begin_tile();       // descriptors only: the compute waves stage this tile during this iteration
        } else {
#ifdef KWS_GEMM_STAMP
          t_mark = __builtin_amdgcn_s_memtime();
#endif
          const int lo = (kt - 1) * ppi, hi = lo + ppi < NPASS ? lo + ppi : NPASS;
          if (lo < NPASS) move_rows(lo, hi);
          WT(t_store);
        }
      }
      if (++kt == nk) kt = 0;
#ifdef KWS_GEMM_STAMP
      t_mark = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
      WT(t_sbar);
    }
#ifdef KWS_GEMM_STAMP
    if (stt == 0 && bid < 8192) {
      g_stamps[bid + 256][0] = t_store;
      g_stamps[bid + 256][1] = t_sbar;
    }
    if ((tid & 63) == 0 && bid < 256) g_stamps[bid + 512][tid >> 6] = t_sbar;
#endif
    begin_tile();
    __syncthreads();   // last tile staged
    move_rows(0, NPASS);
    __syncthreads();
    if (STATS) __syncthreads();
  }
}

template <int BN, int KB, int NLW, int NSW, bool STATS>
__global__ __launch_bounds__((4 + NLW + NSW) * 64, 1) void gemm_nn_ws_kernel(NNArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[nn_ws_smem_floats<BN, KB, STATS>()];
  nn_ws_body<BN, KB, NLW, NSW, STATS>(p, smem, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
struct TNArgs {
  const float* A;   // [M,K] or gathered X
  const float* G;   // [M,N]
  float* ws;        // [S][K][N]
  int64_t M;
  int K, N;
  int64_t chunk;    // rows per split (multiple of 32)
  int k_tiles, n_tiles, S;
  kws_gather_t g;
  int lda = 0;      // wave-specialised kernel only: row pitch of A in floats (0 = K), as NNArgs::lda
};

constexpr int MS = 32;  // rows of M per LDS stage

template <int BKO, int BNO, bool GATHER>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TNArgs p) {
  constexpr int TK = BKO / 2 / 32;
  constexpr int TN = BNO / 2 / 32;
  constexpr int Z_F4 = MS * BKO / 4 / 256;
  constexpr int G_F4 = MS * BNO / 4 / 256;
  constexpr int BKO4 = BKO / 4, BNO4 = BNO / 4;
  __shared__ __attribute__((aligned(16))) float smem[2 * MS * (BKO + BNO)];
  constexpr int STAGE = MS * (BKO + BNO);  // floats per pipeline stage: Z tile then G tile

  // XCD-aware order (PMC: 2.7x the algorithmic bytes left L2 with tile-major order): all output tiles of
  // ONE M-split re-read the same [chunk, K] and [chunk, N] row panels, so they are given to consecutive
  // slots of one XCD (workgroups b, b+8, ... share an L2) and run back to back there.
  const int n_out_tiles = p.k_tiles * p.n_tiles;
  const int xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const int tile = slot % n_out_tiles;
  const int split = (slot / n_out_tiles) * NXCD + xcd;
  if (split >= p.S) return;   // whole workgroup leaves together (before any barrier)
  const int tile_k = tile / p.n_tiles, tile_n = tile % p.n_tiles;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int k0 = tile_k * BKO, n0 = tile_n * BNO;
  const int K = p.K, N = p.N;
  const int64_t m_begin = (int64_t)split * p.chunk;
  const int64_t m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;

  float4 rz[Z_F4], rg[G_F4];
  // gathered A: a thread's columns never change and its rows advance by MS per stage, so the clip / row split
  // (an integer division) is done once and carried incrementally
  int gz_koff[Z_F4], gz_t[Z_F4];
  int64_t gz_b[Z_F4];
  bool gz_kok[Z_F4];
  if (GATHER) {
#pragma unroll
    for (int r = 0; r < Z_F4; ++r) {
      const int idx = tid + r * 256;
      const int row = idx / BKO4, c4 = idx % BKO4;
      const int gk = k0 + c4 * 4;
      const int j = gk / p.g.cin;
      gz_kok[r] = gk < K;
      gz_koff[r] = j * p.g.stride_j + (gk - j * p.g.cin) + p.g.base_off;
      const int64_t gm = m_begin + row;
      gz_b[r] = gm / p.g.L_out;
      gz_t[r] = (int)(gm - gz_b[r] * p.g.L_out);
    }
  }
  auto load_global = [&](int64_t mb) {
#pragma unroll
    for (int r = 0; r < Z_F4; ++r) {
      const int idx = tid + r * 256;
      const int row = idx / BKO4, c4 = idx % BKO4;
      const int64_t gm = mb + row;
      const int gk = k0 + c4 * 4;
      const bool ok = gm < m_end && gk < K;
      if (GATHER) {
        if (gm < m_end && gz_kok[r]) {
          rz[r] = gather4(p.A + gz_b[r] * p.g.x_batch_stride, gz_t[r] * p.g.stride_t + gz_koff[r], p.g.x_len);
        } else {
          rz[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        gz_t[r] += MS;                               // next stage: MS rows further
        while (gz_t[r] >= p.g.L_out) {
          gz_t[r] -= p.g.L_out;
          ++gz_b[r];
        }
      } else {
        rz[r] = ld4_or_zero(p.A + (ok ? gm : 0) * (int64_t)K + gk, ok);
      }
    }
#pragma unroll
    for (int r = 0; r < G_F4; ++r) {
      const int idx = tid + r * 256;
      const int row = idx / BNO4, c4 = idx % BNO4;
      const int64_t gm = mb + row;
      const int gn = n0 + c4 * 4;
      const bool ok = gm < m_end && gn < N;
      rg[r] = ld4_or_zero(p.G + (ok ? gm : 0) * (int64_t)N + gn, ok);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int r = 0; r < Z_F4; ++r) {
      const int idx = tid + r * 256;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + (idx / BKO4) * BKO + (idx % BKO4) * 4]) = rz[r];
    }
#pragma unroll
    for (int r = 0; r < G_F4; ++r) {
      const int idx = tid + r * 256;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + MS * BKO + (idx / BNO4) * BNO + (idx % BNO4) * 4]) = rg[r];
    }
  };

  f32x16 acc[TK][TN];
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int stages = (int)((m_end - m_begin + MS - 1) / MS);
  if (stages > 0) {
    load_global(m_begin);
    store_lds(0);
  }
  __syncthreads();
  for (int st = 0; st < stages; ++st) {
    const int cur = st & 1;
    if (st + 1 < stages) load_global(m_begin + (int64_t)(st + 1) * MS);
    const float* cZ = smem + cur * STAGE + lh * BKO + wk * TK * 32 + li;
    const float* cG = smem + cur * STAGE + MS * BKO + lh * BNO + wn * TN * 32 + li;
#pragma unroll
    for (int s = 0; s < MS / 2; ++s) {
      float a[TK], b[TN];
#pragma unroll
      for (int i = 0; i < TK; ++i) a[i] = cZ[(2 * s) * BKO + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = cG[(2 * s) * BNO + j * 32];
#pragma unroll
      for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < stages) store_lds(cur ^ 1);
    __syncthreads();
  }

  float* out = p.ws + (int64_t)split * K * N;
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + li;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = k0 + wk * TK * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
        if (row < K && col < N) out[(int64_t)row * N + col] = acc[i][j][v];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised TN (wgrad) kernel: the same role split as gemm_nn_ws_kernel.  One workgroup = one
// (M-split, output tile) work item:
//   4 MFMA waves: each owns a 64 x 64 accumulator block and 32 rows of every M-stage (a 128 x 128 tile
//              is 2 x 2 blocks over a 32-row stage; a 128 x 64 / 64 x 128 tile is 2 blocks x 2 row
//              halves of a 64-row stage; a 64 x 64 tile is 4 row quarters of a 128-row stage), so every
//              wave issues 64 MFMAs per barrier whatever the tile shape; blocks that were split over rows
//              are summed through LDS in wave order at the end;
//   NLW loader waves: a stage is U = 1 / 2 / 4 "units" of 32 rows x (BKO + BNO) columns; unit x is owned by
//              loader wave x % NLW, written into LDS during the stage before it is used and re-issued
//              (buffer loads, SGPR row offsets, descriptor range check zero-fills rows past the split)
//              right after, so each load has about a full stage to land.
// Host-checked: K % BKO == 0, N % BNO == 0, chunk = whole stages (32 U rows), 32-bit byte offsets inside a split.
template <int BKO, int BNO>
constexpr int tn_ws_smem_floats() {
  constexpr int U = 4 / ((BKO / 64) * (BNO / 64));
  return 2 * U * 32 * (BKO + BNO) > (U > 1 ? 4 * 64 * 64 : 0) ? 2 * U * 32 * (BKO + BNO) : 4 * 64 * 64;
}
template <int BKO, int BNO>
constexpr int tn_ws_threads() { return (4 + ((128 / BKO) * (128 / BNO) > 2 ? 4 : 2)) * 64; }
// As for the NN kernel: a device function of the block id; NTHREADS = threads of the launching kernel (waves past the
// kernel's own 4 + NLW only keep the barrier count).
template <int BKO, int BNO, int NTHREADS>
__device__ __forceinline__ void tn_ws_body(const TNArgs& p, float* const smem, const int bid) {
  constexpr int WK = BKO / 64, WN = BNO / 64, WS = 4 / (WK * WN);
  constexpr int U = WS;                             // 32-row units per stage
  constexpr bool IL = WS != 2;                      // interleaved column blocks (see the fragment reads)
  constexpr int NLW = U > 2 ? 4 : 2;
  constexpr int NCT = 256;
  constexpr int COLS = BKO + BNO;
  constexpr int UNIT = 32 * COLS;                   // floats per unit: Z rows then G rows, row-major [32][COLS]
  constexpr int SLOT = U * UNIT;
  constexpr int RED = WS > 1 ? 4 * 64 * 64 : 0;
  constexpr int SMEM = 2 * SLOT > RED ? 2 * SLOT : RED;
  static_assert(SMEM * 4 <= 160 * 1024, "LDS budget");
  static_assert(SMEM == tn_ws_smem_floats<BKO, BNO>(), "tn_ws_smem_floats out of step with the layout");
  static_assert(NTHREADS >= (4 + NLW) * 64, "the launching kernel must bring the kernel's own waves");

  const int n_out_tiles = p.k_tiles * p.n_tiles;
  const int xcd = bid % NXCD, slot_id = bid / NXCD;
  const int tile = slot_id % n_out_tiles;
  const int split = (slot_id / n_out_tiles) * NXCD + xcd;
  if (split >= p.S) return;
  const int tile_k = tile / p.n_tiles, tile_n = tile % p.n_tiles;
  const int tid = threadIdx.x;
  const int k0 = tile_k * BKO, n0 = tile_n * BNO;
  const int K = p.K, N = p.N;
  const int64_t m_begin = (int64_t)split * p.chunk;
  const int64_t m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
  const int rows = (int)(m_end - m_begin);
  const int G = (rows + 32 * U - 1) / (32 * U);     // stages; barriers per wave: 1 + G (+ 1 when WS > 1)

  if (tid < NCT) {
    // ------------------------------------------------------------------ MFMA waves
    const int lane = tid & 63, wave = tid >> 6;
    const int wms = wave % WS, wkn = wave / WS;
    const int wk = wkn / WN, wn = wkn % WN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    __syncthreads();
#ifdef KWS_GEMM_STAMP
    unsigned long long tn_mma = 0, tn_bar = 0, tn_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long tn_begin = tn_mark, tn_rbegin = __builtin_amdgcn_s_memrealtime();
#define TNT(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - tn_mark; tn_mark = now_; } while (0)
#else
#define TNT(acc_)
#endif
    for (int g = 0; g < G; ++g) {
      // my 32 rows of this stage = unit wms of slot g & 1; lane half lh takes the odd / even row of a pair
      // The two 32-wide blocks of a wave's 64 Z (G) columns are INTERLEAVED: block i (j) is the columns 2 li + i, so one
      // 8-byte LDS read feeds both blocks - 2 reads per 4 MFMAs instead of 4.  (Stamps: with one 4-byte read per MFMA the
      // wave needed 4,780 cycles to issue the 64 MFMAs of a stage - every non-MFMA instruction in an MFMA wave's stream
      // costs ~12 matrix cycles.)  The epilogue maps accumulator (block, row / lane) back to 2 x + block.
      // IL: not for the two-unit stages (128 x 64 / 64 x 128 tiles) - measured, they get 6 - 10 % SLOWER with it: with the
      // shorter issue their stage is paced by the loads (5,200 - 5,400 cycles, two or four loader waves alike; 4,350 without
      // the loaders): these tiles re-read z per column tile and sit on the memory system's rate
      const float* cz = smem + (g & 1) * SLOT + wms * UNIT + lh * COLS + wk * 64 + (IL ? 2 * li : li);
      const float* cg = cz + BKO - wk * 64 + wn * 64;
      float a0[2], b0[2], a1[2], b1[2];
      auto ld = [&](float (&a)[2], float (&b)[2], int s) {
        if (IL) {
          const float2 av = *reinterpret_cast<const float2*>(cz + 2 * s * COLS);
          const float2 bv = *reinterpret_cast<const float2*>(cg + 2 * s * COLS);
          a[0] = av.x; a[1] = av.y;
          b[0] = bv.x; b[1] = bv.y;
        } else {
          a[0] = cz[2 * s * COLS];
          a[1] = cz[2 * s * COLS + 32];
          b[0] = cg[2 * s * COLS];
          b[1] = cg[2 * s * COLS + 32];
        }
      };
      auto mm = [&](const float (&a)[2], const float (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      };
      ld(a0, b0, 0);
      ld(a1, b1, 1);
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < 16) ld(a0, b0, s + 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < 16) ld(a1, b1, s + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      TNT(tn_mma);
      __syncthreads();
      TNT(tn_bar);
    }
#ifdef KWS_GEMM_STAMP
    const unsigned long long tn_loop_end = __builtin_amdgcn_s_memtime();
#endif
    float* out = p.ws + (int64_t)split * K * N;
    if (WS == 1) {
      // accumulator (i, j) register v of lane (li, lh) is dW[k0 + wk 64 + 2 r + i][n0 + wn 64 + 2 li + j], r = the MFMA's row of
      // register v: the two column blocks of a row leave as one 8-byte store (256 contiguous bytes per row and wave)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = k0 + wk * 64 + 2 * ((v & 3) + 8 * (v >> 2) + 4 * lh) + i;
          *reinterpret_cast<float2*>(out + (int64_t)row * N + n0 + wn * 64 + 2 * li) = make_float2(acc[i][0][v], acc[i][1][v]);
        }
    } else {
      // the pipeline slots are idle now: every wave parks its block, then the WS waves of a block share its
      // rows and add the WS copies in wave order
      float* mine = smem + wave * 4096;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int r = (v & 3) + 8 * (v >> 2) + 4 * lh;
            mine[IL ? (2 * r + i) * 64 + 2 * li + j : (i * 32 + r) * 64 + j * 32 + li] = acc[i][j][v];
          }
      __syncthreads();
      const float* blk = smem + wkn * WS * 4096;
      const int t = wms * 64 + lane;
      for (int f4 = t; f4 < 1024; f4 += WS * 64) {
        float4 s = *reinterpret_cast<const float4*>(blk + f4 * 4);
#pragma unroll
        for (int w = 1; w < WS; ++w) {
          const float4 u = *reinterpret_cast<const float4*>(blk + w * 4096 + f4 * 4);
          s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
        }
        const int row = k0 + wk * 64 + f4 / 16, col = n0 + wn * 64 + (f4 % 16) * 4;
        *reinterpret_cast<float4*>(out + (int64_t)row * N + col) = s;
      }
    }
#ifdef KWS_GEMM_STAMP
    if (tid == 0 && bid < 4096) {
      g_stamps[bid][0] = tn_mma; g_stamps[bid][1] = tn_bar; g_stamps[bid][3] = (unsigned long long)G;
      g_stamps[bid][2] = __builtin_amdgcn_s_memrealtime() - tn_rbegin;   // 100 MHz ticks of the whole item (clock = cycles / ticks)
      g_stamps[bid][4] = tn_loop_end - tn_begin; g_stamps[bid][5] = __builtin_amdgcn_s_memtime() - tn_loop_end;
    }
#endif
  } else if (tid < NCT + NLW * 64) {
    // ------------------------------------------------------------------ loader waves
    const int lane = tid & 63;
    const int lw = __builtin_amdgcn_readfirstlane((tid - NCT) >> 6);
    // one wave instruction loads RZ (RG) whole rows of the unit's Z (G) part: lane -> (row in group, column)
    constexpr int ZC4 = BKO / 4, GC4 = BNO / 4;     // float4 per Z / G row
    constexpr int RZ = 64 / ZC4, RG = 64 / GC4;     // rows per instruction: 2 or 4
    constexpr int Z_F4 = 32 / RZ, G_F4 = 32 / RG;   // instructions per unit: 16 or 8 each
    const int zrow = lane / ZC4, zc = (lane % ZC4) * 4;
    const int grow = lane / GC4, gc = (lane % GC4) * 4;
    const int lda = p.lda ? p.lda : K;              // row pitch of A (a strided row view, NNArgs::lda)
    const int z_voff = (zrow * lda + k0 + zc) * 4, g_voff = (grow * N + n0 + gc) * 4;
    const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.A + m_begin * lda), 0, rows > 0 ? ((rows - 1) * lda + K) * 4 : 0, KWS_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.G + m_begin * N), 0, rows * N * 4, KWS_BUFFER_RSRC_FLAGS);
    float4 rz[Z_F4], rg[G_F4];
    auto issue = [&](int x) {                       // unit x = rows [32 x, 32 x + 32) of my split
      if (KWS_WS_ABL & 4) return;
#pragma unroll
      for (int r = 0; r < Z_F4; ++r) rz[r] = buf_ld4(zres, z_voff, (x * 32 + RZ * r) * lda * 4);
#pragma unroll
      for (int r = 0; r < G_F4; ++r) rg[r] = buf_ld4(gres, g_voff, (x * 32 + RG * r) * N * 4);
    };
    auto write_lds = [&](int x) {
      if (KWS_WS_ABL & 2) { asm volatile("" :: "v"(rz[0].x), "v"(rg[0].x)); return; }
      float* dst = smem + ((x / U) & 1) * SLOT + (x % U) * UNIT;
      float* dz = dst + zrow * COLS + zc;
      float* dg = dst + grow * COLS + BKO + gc;
#pragma unroll
      for (int r = 0; r < Z_F4; ++r) *reinterpret_cast<float4*>(dz + RZ * r * COLS) = rz[r];
#pragma unroll
      for (int r = 0; r < G_F4; ++r) *reinterpret_cast<float4*>(dg + RG * r * COLS) = rg[r];
    };
    int x = lw;                                     // my next unit
    if (x < U) {
      issue(x);
      write_lds(x);
      x += NLW;
    }
    issue(x);
    __syncthreads();
#ifdef KWS_GEMM_STAMP
    unsigned long long tl_work = 0, tl_bar = 0, tl_mark = __builtin_amdgcn_s_memtime();
#define TLT(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - tl_mark; tl_mark = now_; } while (0)
#else
#define TLT(acc_)
#endif
    for (int g = 0; g < G; ++g) {
      if (x / U == g + 1) {                         // my unit belongs to the next stage: write it, re-issue
        write_lds(x);
        x += NLW;
        issue(x);
      }
      TLT(tl_work);
      __syncthreads();
      TLT(tl_bar);
    }
#ifdef KWS_GEMM_STAMP
    if ((tid & 63) == 0 && lw == 0 && bid < 4096) { g_stamps[bid][6] = tl_work; g_stamps[bid][7] = tl_bar; }
#endif
    if (WS > 1) __syncthreads();
  } else {
    // ------------------------------------------------------------------ spare waves of a wider launching kernel: the barriers only
    __syncthreads();
    for (int g = 0; g < G; ++g) __syncthreads();
    if (WS > 1) __syncthreads();
  }
}

template <int BKO, int BNO>
__global__ __launch_bounds__((4 + ((128 / BKO) * (128 / BNO) > 2 ? 4 : 2)) * 64, 1) void gemm_tn_ws_kernel(TNArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[tn_ws_smem_floats<BKO, BNO>()];
  constexpr int NT = tn_ws_threads<BKO, BNO>();
  tn_ws_body<BKO, BNO, NT>(p, smem, blockIdx.x);
}

// One launch = the input-gradient GEMM (dZ = dY W^T: the first nn_grid blocks, the persistent NN walk) AND the weight-gradient
// GEMM of the same layer (dW slabs = Z^T dY: one block per work item behind them).  The two are independent; in one grid the
// hardware starts weight-gradient workgroups on a CU the moment its NN workgroup has ended - the NN kernel's last partial
// round (stamps, round 4: 1 - 12 us of idle CUs per launch) and the second launch's ramp-up are filled without an event
// (two streams cost 30 us per layer in events, profiles/r04_dgrad_wgrad_two_streams.txt).  Same code paths, same per-element
// arithmetic as the two separate launches: bit-identical dZ and slabs.
template <int BN, int KB, int BKO, int BNO>
__global__ __launch_bounds__(512, 1) void gemm_dgrad_wgrad_kernel(NNArgs a, TNArgs t, int nn_grid) {
  constexpr int SM_NN = nn_ws_smem_floats<BN, KB, false>(), SM_TN = tn_ws_smem_floats<BKO, BNO>();
  __shared__ __attribute__((aligned(16))) float smem[SM_NN > SM_TN ? SM_NN : SM_TN];
  if ((int)blockIdx.x < nn_grid) nn_ws_body<BN, KB, 2, 2, false>(a, smem, blockIdx.x, nn_grid);
  else tn_ws_body<BKO, BNO, 512>(t, smem, (int)blockIdx.x - nn_grid);
}

// out[i] = sum_k ws[k][i], k ascending within 4 interleaved groups that are combined in a fixed order
// (bit-reproducible).  64 float4 columns x 4 slab groups per workgroup so that the S slabs of the small
// K x N outputs are read by S/4-deep loops on many workgroups instead of S-deep loops on a few.
// Sums S slabs that lie `stride4` float4 apart (the plain case: stride4 = n4).  blockIdx.y selects a GROUP of `per_group`
// slabs and the sum goes to out + blockIdx.y * out_stride4: the first stage of a two-stage sum writes every group's
// result over the group's own first slab (each thread reads its column of all slabs before it writes it).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* ws, float* out, int64_t n4, int S,
                                                           int64_t stride4 = -1, int per_group = 0,
                                                           int64_t out_stride4 = 0) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + col;
  if (stride4 < 0) stride4 = n4;
  const float4* w = reinterpret_cast<const float4*>(ws);
  if (per_group > 0) {
    const int first = blockIdx.y * per_group;
    w += (int64_t)first * stride4;
    out += (int64_t)blockIdx.y * out_stride4 * 4;
    S = S - first < per_group ? S - first : per_group;
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
  if (i < n4) {
    int k = grp;
    for (; k + 4 < S; k += 8) {
      const float4 v = w[(int64_t)k * stride4 + i];
      const float4 u = w[(int64_t)(k + 4) * stride4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
    }
    if (k < S) {
      const float4 v = w[(int64_t)k * stride4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i < n4) {
    float4 t = red[0][col];
#pragma unroll
    for (int g = 1; g < 4; ++g) {
      const float4 v = red[g][col];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = t;
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int y = by + r, x = bx + tx;
    tile[r][tx] = (y < rows && x < cols) ? in[(int64_t)y * cols + x] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int y = bx + r, x = by + tx;  // out is [cols][rows]
    if (y < cols && x < rows) out[(int64_t)y * rows + x] = tile[tx][r];
  }
}

// up to KWS_TRANSPOSE_BATCH matrices in one launch (the dgrad GEMMs' transposed pointwise kernels: eleven
// 5 us launches otherwise)
struct TransposeBatch {
  const float* in[KWS_TRANSPOSE_BATCH];
  float* out[KWS_TRANSPOSE_BATCH];
  int rows[KWS_TRANSPOSE_BATCH], cols[KWS_TRANSPOSE_BATCH], tile_end[KWS_TRANSPOSE_BATCH];
  int n;
};
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch b) {
  __shared__ float tile[32][33];
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.tile_end[m]) ++m;
  const int t = blockIdx.x - (m ? b.tile_end[m - 1] : 0);
  const int rows = b.rows[m], cols = b.cols[m];
  const int tx_n = (cols + 31) / 32;
  const int bx = (t % tx_n) * 32, by = (t / tx_n) * 32;
  const float* in = b.in[m];
  float* out = b.out[m];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int y = by + r, x = bx + tx;
    tile[r][tx] = (y < rows && x < cols) ? in[(int64_t)y * cols + x] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int y = bx + r, x = by + tx;  // out is [cols][rows]
    if (y < cols && x < rows) out[(int64_t)y * rows + x] = tile[tx][r];
  }
}

// The slab sums of SEVERAL weight-gradient GEMMs in one launch (net.hip: the eleven pointwise layers of a training step wrote
// their slabs into regions of their own; summing them per layer cost fourteen 5 - 10 us launches on the dependency chain).
// One workgroup = 64 float4 columns of one GEMM; 4 slab groups (k = grp, grp + 4, ...) with four loads in flight each,
// combined in a fixed order: bit-reproducible.
__global__ __launch_bounds__(256) void reduce_slabs_batch_kernel(SlabBatch b) {
  __shared__ float4 red[4][64];
  kws_reduce_slabs_batch_body(b, blockIdx.x, red);
}

// split heuristic of the TN kernel: enough workgroups to fill 256 CUs, slabs no larger than needed
struct TNPlan {
  int bko, bno;  // 128 or 64 each
  int k_tiles, n_tiles, S;
  int64_t chunk;
};
// ws: the wave-specialised kernel (tile width chosen per dimension); otherwise the 4-wave kernel (square tiles)
bool tn_ws_eligible(int K, int N, bool gather) { return !gather && K % 64 == 0 && N % 64 == 0; }
// How M is cut into splits.  A work item = (output tile, split); the kernel deals split s to XCD s % 8 (all tiles of a split
// re-read the same row panels: one L2), an XCD has 32 CUs, and a CU holds one workgroup (two for the 64 KB 128 x 128 kernel).
// Round 2 took S = 512 / tiles: for 256 -> 320 that is 49 splits = 7 on XCD 0 and 6 on the others - 70 items on one XCD's 32
// CUs is THREE rounds where 60 is two, and the launch ran 126 us instead of 85 (320 -> 320: 75 against 50 items, 135 against
// 91 us; in-kernel stamps and rocprofv3 agree, profiles/r03_tn_plan.txt).  The split size is now searched: every candidate
// (whole stages of the tile shape) is priced as the busiest XCD's items per CU x (stages x cycles per stage + the fixed
// cost of an item: first loads, epilogue) + the slab sum's traffic, and the cheapest wins.
#ifndef KWS_TN_MAX_S
#define KWS_TN_MAX_S 256
#endif
#ifndef KWS_TN_PAIR_PCT
#define KWS_TN_PAIR_PCT 0     // experiment (round 4): price two co-resident 128 x 128 items (64 KB of LDS each) at this % of two in a row
#endif
int64_t tn_cost(int64_t M, int K, int N, int tiles, int U, int64_t chunk, int* S_out) {
  const int64_t S = ceil_div64(M, chunk);
  const int64_t stages = ceil_div64(chunk, 32 * U);
  const int64_t t_stage = U == 1 ? 4370 : (U == 2 ? 4870 : 4550);   // measured shader cycles per stage (64 MFMAs per wave)
  int64_t worst = 0;
  for (int x = 0; x < NXCD; ++x) {
    const int64_t cnt = x < S ? (S - x + NXCD - 1) / NXCD : 0;
    const int64_t per_cu = ceil_div64(tiles * cnt, 32);
    int64_t t = per_cu * (stages * t_stage + 4500);
    if (KWS_TN_PAIR_PCT && U == 1) {      // the CU holds two such workgroups at once
      const int64_t pairs = per_cu / 2, single = per_cu % 2;
      t = pairs * (2 * (stages * t_stage + 4500) * KWS_TN_PAIR_PCT / 100) + single * (stages * t_stage + 4500);
    }
    if (t > worst) worst = t;
  }
  *S_out = (int)S;
  return worst + (int64_t)((double)S * K * N * 4.0 * 0.8e-3) + 6000;   // + slab sum: bytes at ~2.5 TB/s, one launch
}
TNPlan tn_plan_search(int64_t M, int K, int N, bool ws);
// The search below walks s_lo * 15 + 64 candidates with an 8-XCD inner loop and is asked for the same few shapes ~33 times per
// training step (launch, workspace size, layout): memoised per (M, K, N, ws) - a step's shapes are a dozen entries.
TNPlan tn_plan(int64_t M, int K, int N, bool ws) {
  static std::mutex mu;
  static std::map<std::tuple<int64_t, int, int, bool>, TNPlan> cache;
  const auto key = std::make_tuple(M, K, N, ws);
  {
    std::lock_guard<std::mutex> lk(mu);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  const TNPlan pl = tn_plan_search(M, K, N, ws);
  std::lock_guard<std::mutex> lk(mu);
  if (cache.size() > 4096) cache.clear();           // (a caller sweeping shapes: keep the map bounded)
  cache[key] = pl;
  return pl;
}
TNPlan tn_plan_search(int64_t M, int K, int N, bool ws) {
  TNPlan pl;
  if (ws) {
    pl.bko = (K % 128 == 0) ? 128 : 64;
    pl.bno = (N % 128 == 0) ? 128 : 64;
  } else {
    pl.bko = pl.bno = (K % 128 == 0 && N % 128 == 0) ? 128 : 64;
  }
  pl.k_tiles = ceil_div(K, pl.bko);
  pl.n_tiles = ceil_div(N, pl.bno);
  const int tiles = pl.k_tiles * pl.n_tiles;
  if (!ws) {
    // 4-wave kernel (gathered operands, ragged widths): ~3 workgroups per CU in flight
    int64_t S = ceil_div64(768, tiles);
    if (S > 256) S = 256;               // bounds the partial-slab traffic (S * K * N floats)
    const int64_t maxS = M / 256 > 1 ? M / 256 : 1;
    if (S > maxS) S = maxS;
    int64_t chunk = ceil_div64(ceil_div64(M, S), 128) * 128;
    if (chunk < 128) chunk = 128;
    pl.chunk = chunk;
    pl.S = (int)ceil_div64(M > 0 ? M : 1, chunk);
    return pl;
  }
  const int U = 4 / ((pl.bko / 64) * (pl.bno / 64));   // 32-row units per stage of the wave-specialised kernel
  const int64_t g = 32 * U;
  const int64_t s_lo = std::max<int64_t>(1, ceil_div64(M, KWS_TN_MAX_S * g));   // S <= 256: bounds the slab traffic
  int64_t best = -1, best_chunk = s_lo * g;
  const int64_t max_chunk = ((1ll << 31) - 1) / (4ll * (K > N ? K : N));   // 32-bit byte offsets inside a split's buffer views
  for (int64_t st = s_lo; st < s_lo * 16 + 64; ++st) {
    if (st * g > max_chunk && st > s_lo) break;
    int S = 0;
    const int64_t c = tn_cost(M, K, N, tiles, U, st * g, &S);
    if (best < 0 || c < best) {
      best = c;
      best_chunk = st * g;
    }
    if (S <= 1) break;
  }
  pl.chunk = best_chunk;
  pl.S = (int)ceil_div64(M > 0 ? M : 1, best_chunk);
  return pl;
}

// which kernel kws_gemm_nn_f32 runs for a shape, and how many statistics rows it writes
struct NNPlan {
  bool ws;          // wave-specialised kernel
  int bn;           // column-tile width: 128 or 64
  int kb;           // its K-slab depth
  int wgs;          // its grid (= statistics rows: one per workgroup)
  int m_tiles;      // statistics rows of the tile-per-row kernels
};
constexpr bool nn_half_tail() { return true; }   // a short last round is walked in 64-row half tiles
NNPlan nn_plan(int64_t M, int K, int N, bool gather) {
  NNPlan pl;
  pl.m_tiles = (int)ceil_div64(M, 128);
  int BN = (N % 128 == 0) ? 128 : 64;
  if (BN == 128 && !gather && K % 64 == 0 && K >= 128) {
    // One workgroup per CU walks ceil(tiles / 256) rounds of tiles; the small late layers have 1.1 - 2.3
    // 128-wide tiles per CU and lose 25 - 44 % to the last partial round.  64-wide tiles (0.55 of the time of
    // a 128-wide one with the 64-deep K-slabs) quantise finer: take them when they make the walk shorter.
    // (a short last round is walked in half tiles, see the kernel: ~0.55 of a round when the leftover fits twice)
    const int64_t t128 = (int64_t)pl.m_tiles * (N / 128);
    auto rounds = [](int64_t t) {
      const int64_t e = t % 256;
      return (double)(t / 256) + (e == 0 ? 0.0 : (nn_half_tail() && 2 * e <= 256 ? 0.55 : 1.0));
    };
    const double cost128 = rounds(t128), cost64 = 0.55 * rounds(2 * t128);
    if (cost64 < cost128) BN = 64;
  }
  pl.bn = BN;
  pl.kb = (BN == 64 && K % 64 == 0 && K >= 128) ? 64 : 32;
#ifdef KWS_NN_KB32_K128   // experiment (round 5): a 64-wide tile of K = 128 is TWO 64-deep iterations long and carries its staging + statistics
  if (BN == 64 && K == 128) pl.kb = 32;   // epilogue in both; four 32-deep iterations spread it (profiles/r05_nn_fwd_l1.txt)
#endif
  const int64_t slots = ceil_div64(pl.m_tiles, NXCD) * ceil_div(N, BN);
  // wave-specialised kernel (default): needs whole K-slabs and column tiles, and 32-bit byte offsets
  // inside one tile's buffer views (128 rows of A / C, all of W); everything else (the gathered first
  // convolution, ragged K or N) takes the persistent kernel
  pl.ws = !gather && K % pl.kb == 0 && K >= 2 * pl.kb && N % BN == 0 && N <= KWS_WS_MAX_N &&
          (int64_t)K * N * 4 < (1ll << 31) && 128ll * K * 4 < (1ll << 31) && 128ll * N * 4 < (1ll << 31);
  int per_xcd = (int)(slots < 32 ? slots : 32);    // one 8-wave workgroup per CU (153 KB LDS), 32 CUs per XCD
  pl.wgs = per_xcd * NXCD;
  return pl;
}

template <bool GATHER>
int launch_nn(const NNArgs& a0, hipStream_t st) {
  NNArgs a = a0;
  const NNPlan pl = nn_plan(a.M, a.K, a.N, GATHER);
  const bool wide = pl.bn == 128;
  const int BN = pl.bn;
  a.m_tiles = (int)ceil_div64(a.M, 128);
  a.n_tiles = ceil_div(a.N, BN);
  a.half_tail = nn_half_tail() ? 1 : 0;
  a.last_rows = (int)(a.M - (int64_t)(a.m_tiles - 1) * 128);
  a.inv_n_tiles = a.n_tiles > 1 ? (unsigned)(((1ull << 32) + a.n_tiles - 1) / a.n_tiles) : 0u;
  const int64_t slots = ceil_div64(a.m_tiles, NXCD) * a.n_tiles;
  const int64_t grid = slots * NXCD;
  if (grid <= 0 || grid > 0x7FFFFFFF) {
    kws_set_error("gemm_nn: grid %lld out of range", (long long)grid);
    return KWS_E_INVALID;
  }
  const bool stats = a.stats != nullptr;
  if (pl.ws) {
    dim3 gp((unsigned)pl.wgs), bp(8 * 64);
    if (wide) {
      if (stats) hipLaunchKernelGGL((gemm_nn_ws_kernel<128, 32, 2, 2, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_ws_kernel<128, 32, 2, 2, false>), gp, bp, 0, st, a);
    } else if (pl.kb == 64) {
      if (stats) hipLaunchKernelGGL((gemm_nn_ws_kernel<64, 64, 2, 2, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_ws_kernel<64, 64, 2, 2, false>), gp, bp, 0, st, a);
    } else {
#ifdef KWS_NN_NLW_K32     // experiment (round 6): more loader waves for the 32-deep slabs of the 64-wide tiles (K = 64: C3's first blocks)
      const dim3 bq((4 + KWS_NN_NLW_K32 + 2) * 64);
      if (stats) hipLaunchKernelGGL((gemm_nn_ws_kernel<64, 32, KWS_NN_NLW_K32, 2, true>), gp, bq, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_ws_kernel<64, 32, KWS_NN_NLW_K32, 2, false>), gp, bq, 0, st, a);
#else
      if (stats) hipLaunchKernelGGL((gemm_nn_ws_kernel<64, 32, 2, 2, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_ws_kernel<64, 32, 2, 2, false>), gp, bp, 0, st, a);
#endif
    }
    KWS_LAUNCH_CHECK("gemm_nn_ws_kernel");
    return KWS_OK;
  }
  {
    // persistent: 2 workgroups per CU (69.6 KB LDS each), 32 CUs per XCD
    int per_xcd = (int)(slots < 64 ? slots : 64);
      dim3 gp((unsigned)(per_xcd * NXCD)), bp(256);
    if (wide) {
      if (stats) hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 128, 2, 2, GATHER, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 128, 2, 2, GATHER, false>), gp, bp, 0, st, a);
    } else {
      if (stats) hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 64, 2, 2, GATHER, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 64, 2, 2, GATHER, false>), gp, bp, 0, st, a);
    }
    KWS_LAUNCH_CHECK("gemm_nn_persist_kernel");
  }
  return KWS_OK;
}

// dW == nullptr: the slabs only (the caller sums them later: kws_reduce_slabs_batch); *S_out = how many were written
template <bool GATHER>
int launch_tn(TNArgs a, float* dW, hipStream_t st, int* S_out = nullptr) {
  bool ws_ok = tn_ws_eligible(a.K, a.N, GATHER);
  TNPlan pl = tn_plan(a.M, a.K, a.N, ws_ok);
  if (ws_ok && pl.chunk * (int64_t)(a.K > a.N ? a.K : a.N) * 4 >= (1ll << 31)) {   // 32-bit offsets inside a split
    ws_ok = false;                                     // (a split of > 1 M rows: M > 2^28): the 4-wave kernel and ITS plan
    pl = tn_plan(a.M, a.K, a.N, false);
  }
  a.chunk = pl.chunk;
  a.k_tiles = pl.k_tiles;
  a.n_tiles = pl.n_tiles;
  a.S = pl.S;
  dim3 g((unsigned)(pl.k_tiles * pl.n_tiles * ceil_div(pl.S, NXCD) * NXCD)), b(256);
  if (ws_ok) {
    if (pl.bko == 128 && pl.bno == 128) hipLaunchKernelGGL((gemm_tn_ws_kernel<128, 128>), g, dim3(6 * 64), 0, st, a);
    else if (pl.bko == 128) hipLaunchKernelGGL((gemm_tn_ws_kernel<128, 64>), g, dim3(6 * 64), 0, st, a);
    else if (pl.bno == 128) hipLaunchKernelGGL((gemm_tn_ws_kernel<64, 128>), g, dim3(6 * 64), 0, st, a);
    else hipLaunchKernelGGL((gemm_tn_ws_kernel<64, 64>), g, dim3(8 * 64), 0, st, a);
  } else if (pl.bko == 128) {
    hipLaunchKernelGGL((gemm_tn_kernel<128, 128, GATHER>), g, b, 0, st, a);
  } else {
    hipLaunchKernelGGL((gemm_tn_kernel<64, 64, GATHER>), g, b, 0, st, a);
  }
  KWS_LAUNCH_CHECK("gemm_tn_kernel");
  if (S_out) *S_out = pl.S;
  if (dW == nullptr) return KWS_OK;
  const int64_t n4 = (int64_t)a.K * a.N / 4;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div64(n4, 64)), dim3(256), 0, st, a.ws, dW, n4, pl.S,
                     (int64_t)-1, 0, (int64_t)0);
  KWS_LAUNCH_CHECK("reduce_slabs_kernel");
  return KWS_OK;
}

// The fused launch of a layer's input-gradient and weight-gradient GEMMs (gemm_dgrad_wgrad_kernel).  Returns 1 when the pair is
// not eligible (either plan falls to a 4-wave kernel, or a split would need 64-bit offsets): the caller then launches the two
// GEMMs separately, exactly as before.
template <int BN, int KB>
void launch_pair_tn(const NNArgs& a, const TNArgs& t, const TNPlan& tp, int nn_grid, unsigned grid, hipStream_t st) {
  if (tp.bko == 128 && tp.bno == 128) hipLaunchKernelGGL((gemm_dgrad_wgrad_kernel<BN, KB, 128, 128>), dim3(grid), dim3(512), 0, st, a, t, nn_grid);
  else if (tp.bko == 128) hipLaunchKernelGGL((gemm_dgrad_wgrad_kernel<BN, KB, 128, 64>), dim3(grid), dim3(512), 0, st, a, t, nn_grid);
  else if (tp.bno == 128) hipLaunchKernelGGL((gemm_dgrad_wgrad_kernel<BN, KB, 64, 128>), dim3(grid), dim3(512), 0, st, a, t, nn_grid);
  else hipLaunchKernelGGL((gemm_dgrad_wgrad_kernel<BN, KB, 64, 64>), dim3(grid), dim3(512), 0, st, a, t, nn_grid);
}
// NN arguments completed from the plan + the launch of the pair kernel's instantiation for (NN tile form, TN tile form)
void launch_pair(NNArgs a, const NNPlan& np, const TNArgs& t, const TNPlan& tp, unsigned grid, hipStream_t st) {
  a.m_tiles = (int)ceil_div64(a.M, 128);
  a.n_tiles = ceil_div(a.N, np.bn);
  a.half_tail = nn_half_tail() ? 1 : 0;
  a.last_rows = (int)(a.M - (int64_t)(a.m_tiles - 1) * 128);
  a.inv_n_tiles = a.n_tiles > 1 ? (unsigned)(((1ull << 32) + a.n_tiles - 1) / a.n_tiles) : 0u;
  const int nn_grid = np.wgs;
  if (np.bn == 128) launch_pair_tn<128, 32>(a, t, tp, nn_grid, grid, st);
  else if (np.kb == 64) launch_pair_tn<64, 64>(a, t, tp, nn_grid, grid, st);
  else launch_pair_tn<64, 32>(a, t, tp, nn_grid, grid, st);
}
// eligibility of the pair launch, apart from the launch itself: the profiler's books open only for a launch that happens
bool dgrad_wgrad_eligible(const NNArgs& a, const TNArgs& t, NNPlan* np_out, TNPlan* tp_out, int64_t* tn_grid_out) {
  const NNPlan np = nn_plan(a.M, a.K, a.N, false);
  if (!np.ws || a.stats != nullptr) return false;
  if (!tn_ws_eligible(t.K, t.N, false)) return false;
  const TNPlan tp = tn_plan(t.M, t.K, t.N, true);
  if (tp.chunk * (int64_t)(t.K > t.N ? t.K : t.N) * 4 >= (1ll << 31)) return false;
  const int64_t tn_grid = (int64_t)tp.k_tiles * tp.n_tiles * ceil_div(tp.S, NXCD) * NXCD;
  if (np.wgs % NXCD != 0 || tn_grid + np.wgs > 0x7FFFFFFF) return false;   // (the weight-gradient blocks keep their XCD: nn_grid is 8 x per-XCD)
  *np_out = np; *tp_out = tp; *tn_grid_out = tn_grid;
  return true;
}
int launch_dgrad_wgrad(const NNArgs& a, TNArgs t, const NNPlan& np, const TNPlan& tp, int64_t tn_grid, hipStream_t st, int* S_out) {
  t.chunk = tp.chunk; t.k_tiles = tp.k_tiles; t.n_tiles = tp.n_tiles; t.S = tp.S;
  launch_pair(a, np, t, tp, (unsigned)(np.wgs + tn_grid), st);
  KWS_LAUNCH_CHECK("gemm_dgrad_wgrad_kernel");
  if (S_out) *S_out = tp.S;
  return KWS_OK;
}

int check_gather(const kws_gather_t* g, int B, int N) {
  KWS_REQUIRE(g != nullptr, "gather descriptor is NULL");
  KWS_REQUIRE(g->L_out > 0 && g->cin > 0 && g->taps > 0 && g->cin % 4 == 0,
              "gather: need L_out>0, taps>0, cin%%4==0 (cin=%d)", g->cin);
  KWS_REQUIRE(g->x_len > 0 && g->x_batch_stride >= g->x_len, "gather: bad x_len/x_batch_stride");
  KWS_REQUIRE(g->x_batch_stride % 2 == 0 && g->stride_t % 2 == 0 && g->stride_j % 2 == 0 && g->base_off % 2 == 0,
              "gather: strides/offsets must be even (8-byte aligned 2-float loads)");
  KWS_REQUIRE(B > 0 && N > 0 && N % 4 == 0, "gather: B=%d N=%d (N%%4 must be 0)", B, N);
  return KWS_OK;
}

}  // namespace

extern "C" {

int kws_gemm_num_row_tiles(int64_t M) {
  const int64_t t = ceil_div64(M, 128);
  return (int)(t > 256 ? t : 256);
}

int kws_gemm_nn_stats_rows(int64_t M, int K, int N) {
  const NNPlan pl = nn_plan(M, K, N, false);
  return pl.ws ? pl.wgs : pl.m_tiles;
}

int kws_gemm_gather_stats_rows(int64_t M) { return (int)ceil_div64(M, 128); }

int kws_gemm_nn_f32(const float* A, const float* W, float* C, int64_t M, int K, int N, float* stats_part,
                    void* stream) {
  KWS_REQUIRE(A && W && C, "gemm_nn: NULL pointer");
  KWS_REQUIRE(M > 0 && K > 0 && N > 0 && K % 4 == 0 && N % 4 == 0, "gemm_nn: M=%lld K=%d N=%d (K,N %% 4)",
              (long long)M, K, N);
  NNArgs a{};
  a.A = A; a.W = W; a.C = C; a.M = M; a.K = K; a.N = N; a.stats = stats_part;
  KwsProfScope prof("gemm_nn", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)K * N + (double)M * N), (hipStream_t)stream);
  return launch_nn<false>(a, (hipStream_t)stream);
}

int kws_gemm_gather_f32(const float* X, const kws_gather_t* g, const float* W, float* C, int B, int N,
                        float* stats_part, void* stream) {
  KWS_REQUIRE(X && W && C, "gemm_gather: NULL pointer");
  KWS_TRY(check_gather(g, B, N));
  NNArgs a{};
  a.A = X; a.W = W; a.C = C; a.M = (int64_t)B * g->L_out; a.K = g->taps * g->cin; a.N = N;
  a.stats = stats_part; a.g = *g;
  KwsProfScope prof("gemm_nn", 2.0 * a.M * a.K * N, 4.0 * ((double)B * g->x_len + (double)a.K * N + (double)a.M * N), (hipStream_t)stream);
  return launch_nn<true>(a, (hipStream_t)stream);
}

int64_t kws_gemm_tn_workspace_floats(int64_t M, int K, int N) {
  // callers size one buffer for the plain and the gathered call: the larger of the two plans
  const int s_plain = tn_plan(M, K, N, tn_ws_eligible(K, N, false)).S, s_gather = tn_plan(M, K, N, false).S;
  return (int64_t)(s_plain > s_gather ? s_plain : s_gather) * K * N;
}

int kws_gemm_tn_f32(const float* A, const float* G, float* dW, int64_t M, int K, int N, float* workspace,
                    void* stream) {
  KWS_REQUIRE(A && G && dW && workspace, "gemm_tn: NULL pointer");
  KWS_REQUIRE(M > 0 && K > 0 && N > 0 && K % 4 == 0 && N % 4 == 0, "gemm_tn: M=%lld K=%d N=%d",
              (long long)M, K, N);
  TNArgs a{};
  a.A = A; a.G = G; a.ws = workspace; a.M = M; a.K = K; a.N = N;
  KwsProfScope prof("gemm_tn", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)M * N + (double)K * N), (hipStream_t)stream);
  return launch_tn<false>(a, dW, (hipStream_t)stream);
}

// internal (net.hip): the weight-gradient GEMM WITHOUT its slab sum - workspace (kws_gemm_tn_workspace_floats) receives *S slabs
// of [K, N]; kws_reduce_slabs_batch sums the slabs of several such calls in one launch
int kws_gemm_tn_slabs_f32(const float* A, const float* G, int64_t M, int K, int N, float* workspace, int* S, hipStream_t stream) {
  KWS_REQUIRE(A && G && workspace && S, "gemm_tn_slabs: NULL pointer");
  KWS_REQUIRE(M > 0 && K > 0 && N > 0 && K % 4 == 0 && N % 4 == 0, "gemm_tn_slabs: M=%lld K=%d N=%d", (long long)M, K, N);
  TNArgs a{};
  a.A = A; a.G = G; a.ws = workspace; a.M = M; a.K = K; a.N = N;
  KwsProfScope prof("gemm_tn", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)M * N + (double)K * N), stream);
  return launch_tn<false>(a, nullptr, stream, S);
}

// internal (net.hip): dZ[M, cin] = dY[M, cout] * WT[cout, cin] and the slabs of dW[cin, cout] = Z^T dY in ONE launch
// (gemm_dgrad_wgrad_kernel); workspace / *S as for kws_gemm_tn_slabs_f32.  Returns 1 (nothing launched) when the shapes are not
// eligible for the fused kernel: the caller then makes the two calls.
int kws_gemm_dgrad_wgrad_f32(const float* dY, const float* WT, float* dZ, const float* Z, int64_t M, int cin, int cout,
                             float* workspace, int* S, hipStream_t stream) {
  KWS_REQUIRE(dY && WT && dZ && Z && workspace && S, "gemm_dgrad_wgrad: NULL pointer");
  KWS_REQUIRE(M > 0 && cin > 0 && cout > 0 && cin % 4 == 0 && cout % 4 == 0, "gemm_dgrad_wgrad: M=%lld cin=%d cout=%d", (long long)M, cin, cout);
  NNArgs a{};
  a.A = dY; a.W = WT; a.C = dZ; a.M = M; a.K = cout; a.N = cin; a.stats = nullptr;
  TNArgs t{};
  t.A = Z; t.G = dY; t.ws = workspace; t.M = M; t.K = cin; t.N = cout;
  NNPlan np;
  TNPlan tp;
  int64_t tn_grid = 0;
  if (!dgrad_wgrad_eligible(a, t, &np, &tp, &tn_grid)) return 1;   // before the profiler's scope: nothing is booked for a launch that does not happen
  const double fl = 2.0 * M * cin * cout;
  KwsProfScope prof("gemm_bwd_pair", 2.0 * fl, 4.0 * (3.0 * M * cout + 2.0 * M * cin + 2.0 * (double)cin * cout) , stream);
  return launch_dgrad_wgrad(a, t, np, tp, tn_grid, stream, S);
}

// internal (net_logmfcc.hip, round 6): a gather that is nothing but every stride_t / cin-th ROW of a row-major matrix - a 1 x 1 convolution
// with stride s over inputs whose length is a multiple of s (the shortcut convolutions of the residual nets: row (b, t) reads input row
// s (b L_out + t)) - is a plain GEMM with a row pitch.  *lda = that pitch in floats.
bool kws_gather_strided_rows(const kws_gather_t* g, int* lda) {
  if (!g || g->taps != 1 || g->base_off != 0 || g->cin <= 0 || g->stride_t < g->cin || g->stride_t % 4 != 0) return false;
  if (g->x_batch_stride != (int64_t)g->L_out * g->stride_t) return false;          // uniform row pitch across clip borders
  if ((int64_t)(g->L_out - 1) * g->stride_t + g->cin > g->x_len) return false;     // every row inside its clip
  *lda = g->stride_t;
  return true;
}
// C[M, N] = A'[M, K] W[K, N] with A' = rows of pitch lda (the wave-specialised kernel; BN statistics rows as kws_gemm_nn_stats_rows).
// Returns 1 (nothing launched) when the shape does not take that kernel: the caller then makes the gathered call.
int kws_gemm_nn_strided_f32(const float* A, int lda, const float* W, float* C, int64_t M, int K, int N, float* stats_part, hipStream_t stream) {
  KWS_REQUIRE(A && W && C, "gemm_nn_strided: NULL pointer");
  KWS_REQUIRE(M > 0 && K > 0 && N > 0 && K % 4 == 0 && N % 4 == 0 && lda >= K && lda % 4 == 0, "gemm_nn_strided: M=%lld K=%d N=%d lda=%d",
              (long long)M, K, N, lda);
  const NNPlan pl = nn_plan(M, K, N, false);
  if (!pl.ws || 128ll * lda * 4 >= (1ll << 31)) return 1;
  NNArgs a{};
  a.A = A; a.W = W; a.C = C; a.M = M; a.K = K; a.N = N; a.stats = stats_part; a.lda = lda;
  KwsProfScope prof("gemm_nn", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)K * N + (double)M * N), stream);
  return launch_nn<false>(a, stream);
}
// the slabs of dW[K, N] = A'^T G with A' = rows of pitch lda (the wave-specialised kernel; workspace / *S as kws_gemm_tn_slabs_f32).
// Returns 1 (nothing launched) when the shape does not take that kernel.
int kws_gemm_tn_slabs_strided_f32(const float* A, int lda, const float* G, int64_t M, int K, int N, float* workspace, int* S, hipStream_t stream) {
  KWS_REQUIRE(A && G && workspace && S, "gemm_tn_slabs_strided: NULL pointer");
  KWS_REQUIRE(M > 0 && K > 0 && N > 0 && K % 4 == 0 && N % 4 == 0 && lda >= K && lda % 4 == 0, "gemm_tn_slabs_strided: M=%lld K=%d N=%d lda=%d",
              (long long)M, K, N, lda);
  if (!tn_ws_eligible(K, N, false)) return 1;
  const TNPlan pl = tn_plan(M, K, N, true);
  if (pl.chunk * (int64_t)(lda > N ? lda : N) * 4 >= (1ll << 31)) return 1;       // 32-bit offsets inside a split
  TNArgs a{};
  a.A = A; a.G = G; a.ws = workspace; a.M = M; a.K = K; a.N = N; a.lda = lda;
  KwsProfScope prof("gemm_tn", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)M * N + (double)K * N), stream);
  return launch_tn<false>(a, nullptr, stream, S);
}

int kws_slab_batch_fill(SlabBatch* b, const float* const* ws, float* const* out, const int64_t* n, const int* S, int count, int* blocks_out,
                        double* bytes_out) {
  KWS_REQUIRE(b && ws && out && n && S && count > 0 && count <= KWS_SLAB_BATCH, "reduce_slabs_batch: bad arguments (count=%d)", count);
  int blocks = 0;
  double bytes = 0;
  for (int i = 0; i < count; ++i) {
    KWS_REQUIRE(ws[i] && out[i] && n[i] > 0 && n[i] % 4 == 0 && S[i] != 0, "reduce_slabs_batch: bad entry %d", i);
    b->ws[i] = ws[i]; b->out[i] = out[i]; b->n4[i] = n[i] / 4; b->S[i] = S[i] < 0 ? -S[i] : S[i];
    b->order[i] = S[i] < 0 ? 1 : 0;
    blocks += (int)ceil_div64(n[i] / 4, 64);
    b->blk_end[i] = blocks;
    bytes += 4.0 * n[i] * (b->S[i] + 1);
  }
  b->n = count;
  *blocks_out = blocks;
  *bytes_out = bytes;
  return KWS_OK;
}

int kws_reduce_slabs_batch(const float* const* ws, float* const* out, const int64_t* n, const int* S, int count, hipStream_t stream) {
  SlabBatch b;
  int blocks = 0;
  double bytes = 0;
  KWS_TRY(kws_slab_batch_fill(&b, ws, out, n, S, count, &blocks, &bytes));
  KwsProfScope prof("slab_sum", 0.0, bytes, stream);
  hipLaunchKernelGGL(reduce_slabs_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b);
  KWS_LAUNCH_CHECK("reduce_slabs_batch_kernel");
  return KWS_OK;
}

int kws_gemm_tn_gather_f32(const float* X, const kws_gather_t* g, const float* G, float* dW, int B, int N,
                           float* workspace, void* stream) {
  KWS_REQUIRE(X && G && dW && workspace, "gemm_tn_gather: NULL pointer");
  KWS_TRY(check_gather(g, B, N));
  TNArgs a{};
  a.A = X; a.G = G; a.ws = workspace; a.M = (int64_t)B * g->L_out; a.K = g->taps * g->cin; a.N = N;
  a.g = *g;
  KwsProfScope prof("gemm_tn", 2.0 * a.M * a.K * N, 4.0 * ((double)B * g->x_len + (double)a.M * N + (double)a.K * N), (hipStream_t)stream);
  return launch_tn<true>(a, dW, (hipStream_t)stream);
}

// internal (net_logmfcc.hip, round 5): the gathered weight-gradient GEMM WITHOUT its slab sum - *S slabs of [taps * cin, N] into
// workspace (kws_gemm_tn_workspace_floats of (B * L_out, taps * cin, N)); the caller queues them for kws_reduce_slabs_batch with
// a NEGATIVE count (the summation order of this kernel's own reduce_slabs_kernel launch: bit-identical to kws_gemm_tn_gather_f32)
int kws_gemm_tn_gather_slabs_f32(const float* X, const kws_gather_t* g, const float* G, int B, int N, float* workspace, int* S,
                                 hipStream_t stream) {
  KWS_REQUIRE(X && G && workspace && S, "gemm_tn_gather_slabs: NULL pointer");
  KWS_TRY(check_gather(g, B, N));
  TNArgs a{};
  a.A = X; a.G = G; a.ws = workspace; a.M = (int64_t)B * g->L_out; a.K = g->taps * g->cin; a.N = N;
  a.g = *g;
  KwsProfScope prof("gemm_tn", 2.0 * a.M * a.K * N, 4.0 * ((double)B * g->x_len + (double)a.M * N + (double)a.K * N), stream);
  return launch_tn<true>(a, nullptr, stream, S);
}

int kws_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
  KWS_REQUIRE(in && out && rows > 0 && cols > 0, "transpose: bad arguments");
  KwsProfScope prof("transpose", 0.0, 8.0 * rows * cols, (hipStream_t)stream);
  dim3 g((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  hipLaunchKernelGGL(transpose_kernel, g, dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
  KWS_LAUNCH_CHECK("transpose_kernel");
  return KWS_OK;
}

int kws_reduce_slab_groups_f32(float* ws, int64_t n, int S, int per_group, hipStream_t st) {
  KWS_REQUIRE(ws && n > 0 && n % 4 == 0 && S > 0 && per_group > 0, "reduce_slab_groups: bad arguments");
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div64(n4, 64), (unsigned)ceil_div(S, per_group)), dim3(256), 0, st,
                     ws, ws, n4, S, n4, per_group, (int64_t)per_group * n4);
  KWS_LAUNCH_CHECK("reduce_slabs_kernel");
  return KWS_OK;
}

int kws_reduce_slabs_f32(const float* ws, float* out, int64_t n, int S, hipStream_t st) {
  KWS_REQUIRE(ws && out && n > 0 && n % 4 == 0 && S > 0, "reduce_slabs: bad arguments");
  const int64_t n4 = n / 4;
  const unsigned gx = (unsigned)ceil_div64(n4, 64);
  if (S > 128 && gx < 256) {
    // few columns, many slabs (the first convolution: 40 column groups x 768 slabs): one workgroup per column group cannot
    // stream them - sum groups of slabs in place first (each group's result over its first slab), then the group sums
    const int per_group = 32, groups = ceil_div(S, per_group);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, (unsigned)groups), dim3(256), 0, st, ws, const_cast<float*>(ws), n4, S,
                       n4, per_group, (int64_t)per_group * n4);
    KWS_LAUNCH_CHECK("reduce_slabs_kernel");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx), dim3(256), 0, st, ws, out, n4, groups, (int64_t)per_group * n4, 0,
                       (int64_t)0);
  } else {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx), dim3(256), 0, st, ws, out, n4, S, (int64_t)-1, 0, (int64_t)0);
  }
  KWS_LAUNCH_CHECK("reduce_slabs_kernel");
  return KWS_OK;
}

// internal (net.hip): n <= KWS_TRANSPOSE_BATCH row-major [rows[i], cols[i]] matrices -> their transposes, one launch
int kws_transpose_batch_f32(const float* const* in, float* const* out, const int* rows, const int* cols, int n,
                            hipStream_t stream) {
  KWS_REQUIRE(in && out && rows && cols && n > 0 && n <= KWS_TRANSPOSE_BATCH, "transpose_batch: bad arguments (n=%d)", n);
  TransposeBatch b;
  int tiles = 0;
  double bytes = 0;
  for (int i = 0; i < n; ++i) {
    KWS_REQUIRE(in[i] && out[i] && rows[i] > 0 && cols[i] > 0, "transpose_batch: bad matrix %d", i);
    b.in[i] = in[i]; b.out[i] = out[i]; b.rows[i] = rows[i]; b.cols[i] = cols[i];
    tiles += ceil_div(rows[i], 32) * ceil_div(cols[i], 32);
    b.tile_end[i] = tiles;
    bytes += 8.0 * rows[i] * cols[i];
  }
  b.n = n;
  KwsProfScope prof("transpose", 0.0, bytes, stream);
  hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, b);
  KWS_LAUNCH_CHECK("transpose_batch_kernel");
  return KWS_OK;
}

}  // extern "C"
