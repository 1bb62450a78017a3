// f32 MFMA GEMM family for the pointwise / first convolutions.
//
//   NN : C[M,N]  = A[M,K] * W[K,N]          forward (A = activations, optionally gathered) and
//                                            dgrad (A = dY, W = W^T)
//   TN : dW[K,N] = A^T[K,M] * G[M,N]        wgrad, split over M, partial slabs reduced in a fixed
//                                            order (bit-reproducible, no float atomics)
//
// gfx950 mapping: v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles/SIMD), 256-thread workgroups =
// 4 waves, each wave owns TMxTN 32x32 accumulator tiles.  Operands are staged global -> VGPR ->
// LDS with a register prefetch of the next K-tile (one barrier per K-tile).  The A tile is kept
// row-major with a 4-float row pad so that one ds_read_b128 per lane fetches 4 consecutive k of
// its row conflict-free; the k order inside an 8-wide group is permuted consistently for A and B
// (lane half h owns k = 8q+4h+r), which the MFMA sum does not care about.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef KWS_GEMM_STAMP
__device__ unsigned long long g_stamps[8192][8];
extern "C" int kws_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#define STAMP(i) do { if (tid == 0 && bid < 8192) g_stamps[bid][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i)
#endif

namespace {

#ifndef KWS_GEMM_BK
#define KWS_GEMM_BK 16
#endif
constexpr int BK = KWS_GEMM_BK;  // K-tile (16: 36 KB LDS per workgroup -> 4 workgroups per CU, out-of-phase overlap)
constexpr int BK4 = BK / 4;
constexpr int LDA = BK + 4;   // padded A-tile row (floats); 144 B keeps 16-B alignment
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

template <int BM, int BN, int WM, int WN, bool GATHER, bool STATS>
__global__ __launch_bounds__(256, 4) void gemm_nn_kernel(NNArgs p) {
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int A_F4 = BM * BK / 4 / 256;
  constexpr int B_F4 = BK * BN / 4 / 256;
  constexpr int BN4 = BN / 4;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM * LDA + BK * BN)];
  constexpr int STAGE = BM * LDA + BK * BN;  // floats per pipeline stage: A tile then B tile

  // XCD-aware tile order: blocks b and b+8 share an XCD (observed round-robin dispatch), so the
  // n-tiles of one row panel are given to consecutive slots of ONE XCD and hit its L2.
  const int bid = blockIdx.x;
  const int xcd = bid % NXCD;
  const int slot = bid / NXCD;
  const int tile_n = slot % p.n_tiles;
  const int tile_m = (slot / p.n_tiles) * NXCD + xcd;
  if (tile_m >= p.m_tiles) return;  // whole workgroup leaves together
#ifdef KWS_GEMM_STAGGER
  // All workgroups of the first residency round start together and, doing identical work, stay in
  // lockstep: their store-bound epilogues then coincide instead of hiding under another workgroup's MFMA
  // loop.  A one-off pseudo-random start delay (first round only) de-phases the workgroups that share a
  // CU; later rounds inherit the spread.  Speed only - never correctness.
  if (bid < 256 * 4) {
    const unsigned d = ((unsigned)bid * 2654435761u >> 29) & 3u;
    for (unsigned q = 0; q < d * KWS_GEMM_STAGGER; ++q) __builtin_amdgcn_s_sleep(127);
  }
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;

  // per-thread A row bookkeeping (rows do not change across K-tiles)
  const float* a_row[A_F4];
  bool a_ok[A_F4];
  int a_t[A_F4];
#pragma unroll
  for (int r = 0; r < A_F4; ++r) {
    const int idx = tid + r * 256;
    const int row = idx / BK4;
    const int64_t gm = m0 + row;
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

  float4 ra[A_F4], rb[B_F4];
  auto load_global = [&](int k0) {
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256;
      const int gk = k0 + (idx % BK4) * 4;
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
      const int row = idx / BN4, c4 = idx % BN4;
      const int gk = k0 + row, gn = n0 + c4 * 4;
      rb[r] = ld4_or_zero(p.W + (int64_t)gk * N + gn, gk < K && gn < N);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + (idx / BK4) * LDA + (idx % BK4) * 4]) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < B_F4; ++r) {
      const int idx = tid + r * 256;
      *reinterpret_cast<float4*>(&smem[buf * STAGE + BM * LDA + (idx / BN4) * BN + (idx % BN4) * 4]) = rb[r];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int nk = (K + BK - 1) / BK;
  STAMP(0);
  load_global(0);
  store_lds(0);
  __syncthreads();
  STAMP(1);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_global((kt + 1) * BK);
    const float* cA = smem + cur * STAGE + (wm * TM * 32 + li) * LDA + lh * 4;
    const float* cB = smem + cur * STAGE + BM * LDA + (lh * 4) * BN + wn * TN * 32 + li;
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float4 a[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(cA + i * 32 * LDA + q * 8);
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
    __syncthreads();
  }

  STAMP(2);
  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + li;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t row = m0 + wm * TM * 32 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
        if (row < M && col < N) p.C[row * N + col] = acc[i][j][v];
      }
    }
  }

  STAMP(3);
  if (STATS) {
    // BatchNorm partial sums of this row tile (rows >= M are exact zeros and add nothing).
    float* red = smem;  // [2][WM][BN]
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
        const int c = wn * TN * 32 + j * 32 + li;
        red[(0 * WM + wm) * BN + c] = s;
        red[(1 * WM + wm) * BN + c] = ss;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += red[(0 * WM + w) * BN + tid];
        ss += red[(1 * WM + w) * BN + tid];
      }
      p.stats[((int64_t)tile_m * 2 + 0) * N + n0 + tid] = s;
      p.stats[((int64_t)tile_m * 2 + 1) * N + n0 + tid] = ss;
    }
  }
  STAMP(4);
}

// ------------------------------------------------------------------------------------------------
// Persistent NN kernel (default).  Same tile math as gemm_nn_kernel, restructured around what the
// in-kernel stamps of the one-tile-per-workgroup version showed (6.1 k cycles exposed first-load
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
#ifdef KWS_GEMM_STAMP
  unsigned long long t_load = 0, t_comp = 0, t_store = 0, t_sync = 0, t_epi = 0, t_mark = 0, n_tiles_done = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#define PT0() t_mark = __builtin_amdgcn_s_memtime()
#define PT(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - t_mark; t_mark = now_; } while (0)
#else
#define PT0()
#define PT(acc_)
#endif
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
      PT0();
      if (kt + 1 < nk) {
        load_global((kt + 1) * PBK);
      } else if (has_next) {
        setup(next);        // first K-slab of the NEXT tile flies while this tile's last slab is computed
        load_global(0);
      }
      PT(t_load);
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
      PT(t_comp);
      if (kt + 1 < nk) store_lds(cur ^ 1);
      PT(t_store);
      __syncthreads();   // after the last slab: every wave is done reading the pipeline buffers
      PT(t_sync);
    }
    PT0();

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
#ifdef KWS_GEMM_STAMP
    n_tiles_done++;
#endif
    if (!has_next) break;
    __syncthreads();   // staging / stats scratch fully consumed before the pipeline buffers are refilled
    store_lds(0);
    __syncthreads();
    PT(t_epi);
    local = next;
  }
#ifdef KWS_GEMM_STAMP
  if (tid == 0 && blockIdx.x < 8192) {
    g_stamps[blockIdx.x][0] = t_load; g_stamps[blockIdx.x][1] = t_comp; g_stamps[blockIdx.x][2] = t_store;
    g_stamps[blockIdx.x][3] = t_sync; g_stamps[blockIdx.x][4] = t_epi; g_stamps[blockIdx.x][5] = n_tiles_done;
    g_stamps[blockIdx.x][6] = __builtin_amdgcn_s_memtime() - t_begin;
  }
#endif
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
  auto load_global = [&](int64_t mb) {
#pragma unroll
    for (int r = 0; r < Z_F4; ++r) {
      const int idx = tid + r * 256;
      const int row = idx / BKO4, c4 = idx % BKO4;
      const int64_t gm = mb + row;
      const int gk = k0 + c4 * 4;
      const bool ok = gm < m_end && gk < K;
      if (GATHER) {
        if (ok) {
          const int64_t b = gm / p.g.L_out;
          const int t = (int)(gm - b * p.g.L_out);
          const int j = gk / p.g.cin;
          const int c = gk - j * p.g.cin;
          rz[r] = gather4(p.A + b * p.g.x_batch_stride,
                          t * p.g.stride_t + j * p.g.stride_j + c + p.g.base_off, p.g.x_len);
        } else {
          rz[r] = make_float4(0.f, 0.f, 0.f, 0.f);
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

// out[i] = sum_k ws[k][i], k ascending within 4 interleaved groups that are combined in a fixed order
// (bit-reproducible).  64 float4 columns x 4 slab groups per workgroup so that the S slabs of the small
// K x N outputs are read by S/4-deep loops on many workgroups instead of S-deep loops on a few.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* ws, float* out, int64_t n4, int S) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + col;
  const float4* w = reinterpret_cast<const float4*>(ws);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
  if (i < n4) {
    int k = grp;
    for (; k + 4 < S; k += 8) {
      const float4 v = w[(int64_t)k * n4 + i];
      const float4 u = w[(int64_t)(k + 4) * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
    }
    if (k < S) {
      const float4 v = w[(int64_t)k * n4 + i];
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

// split heuristic of the TN kernel: enough workgroups to fill 256 CUs, slabs no larger than needed
struct TNPlan {
  int bko;  // 128 or 64 (square tiles)
  int k_tiles, n_tiles, S;
  int64_t chunk;
};
TNPlan tn_plan(int64_t M, int K, int N) {
  TNPlan pl;
  pl.bko = (K % 128 == 0 && N % 128 == 0) ? 128 : 64;
  pl.k_tiles = ceil_div(K, pl.bko);
  pl.n_tiles = ceil_div(N, pl.bko);
  const int tiles = pl.k_tiles * pl.n_tiles;
  int64_t S = ceil_div64(768, tiles);   // ~3 workgroups per CU in flight
  if (S > 256) S = 256;                 // bounds the partial-slab traffic (S * K * N floats)
  const int64_t maxS = M / 256 > 1 ? M / 256 : 1;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int64_t chunk = ceil_div64(ceil_div64(M, S), MS) * MS;
  if (chunk < MS) chunk = MS;
  pl.chunk = chunk;
  pl.S = (int)ceil_div64(M > 0 ? M : 1, chunk);
  return pl;
}

template <bool GATHER>
int launch_nn(const NNArgs& a0, hipStream_t st) {
  NNArgs a = a0;
  const bool wide = (a.N % 128 == 0);
  const int BN = wide ? 128 : 64;
  a.m_tiles = (int)ceil_div64(a.M, 128);
  a.n_tiles = ceil_div(a.N, BN);
  const int64_t slots = ceil_div64(a.m_tiles, NXCD) * a.n_tiles;
  const int64_t grid = slots * NXCD;
  if (grid <= 0 || grid > 0x7FFFFFFF) {
    kws_set_error("gemm_nn: grid %lld out of range", (long long)grid);
    return KWS_E_INVALID;
  }
  const bool stats = a.stats != nullptr;
  static const bool use_v1 = getenv("KWS_GEMM_V1") != nullptr;   // one-tile-per-workgroup kernel, A/B only
  if (!use_v1) {
    // persistent: 2 workgroups per CU (69.6 KB LDS each), 32 CUs per XCD
    int per_xcd = (int)(slots < 64 ? slots : 64);
    if (const char* e = getenv("KWS_GEMM_WGS_PER_XCD")) per_xcd = atoi(e) > 0 && atoi(e) < slots ? atoi(e) : per_xcd;
    dim3 gp((unsigned)(per_xcd * NXCD)), bp(256);
    if (wide) {
      if (stats) hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 128, 2, 2, GATHER, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 128, 2, 2, GATHER, false>), gp, bp, 0, st, a);
    } else {
      if (stats) hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 64, 2, 2, GATHER, true>), gp, bp, 0, st, a);
      else hipLaunchKernelGGL((gemm_nn_persist_kernel<128, 64, 2, 2, GATHER, false>), gp, bp, 0, st, a);
    }
    KWS_LAUNCH_CHECK("gemm_nn_persist_kernel");
    return KWS_OK;
  }
  dim3 g((unsigned)grid), b(256);
  if (wide) {
    if (stats) hipLaunchKernelGGL((gemm_nn_kernel<128, 128, 2, 2, GATHER, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((gemm_nn_kernel<128, 128, 2, 2, GATHER, false>), g, b, 0, st, a);
  } else {
    if (stats) hipLaunchKernelGGL((gemm_nn_kernel<128, 64, 2, 2, GATHER, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((gemm_nn_kernel<128, 64, 2, 2, GATHER, false>), g, b, 0, st, a);
  }
  KWS_LAUNCH_CHECK("gemm_nn_kernel");
  return KWS_OK;
}

template <bool GATHER>
int launch_tn(TNArgs a, float* dW, hipStream_t st) {
  const TNPlan pl = tn_plan(a.M, a.K, a.N);
  a.chunk = pl.chunk;
  a.k_tiles = pl.k_tiles;
  a.n_tiles = pl.n_tiles;
  a.S = pl.S;
  dim3 g((unsigned)(pl.k_tiles * pl.n_tiles * ceil_div(pl.S, NXCD) * NXCD)), b(256);
  if (pl.bko == 128) hipLaunchKernelGGL((gemm_tn_kernel<128, 128, GATHER>), g, b, 0, st, a);
  else hipLaunchKernelGGL((gemm_tn_kernel<64, 64, GATHER>), g, b, 0, st, a);
  KWS_LAUNCH_CHECK("gemm_tn_kernel");
  const int64_t n4 = (int64_t)a.K * a.N / 4;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div64(n4, 64)), dim3(256), 0, st, a.ws, dW, n4, pl.S);
  KWS_LAUNCH_CHECK("reduce_slabs_kernel");
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

int kws_gemm_num_row_tiles(int64_t M) { return (int)ceil_div64(M, 128); }

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
  const TNPlan pl = tn_plan(M, K, N);
  return (int64_t)pl.S * K * N;
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

int kws_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
  KWS_REQUIRE(in && out && rows > 0 && cols > 0, "transpose: bad arguments");
  KwsProfScope prof("transpose", 0.0, 8.0 * rows * cols, (hipStream_t)stream);
  dim3 g((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  hipLaunchKernelGGL(transpose_kernel, g, dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
  KWS_LAUNCH_CHECK("transpose_kernel");
  return KWS_OK;
}

}  // extern "C"
