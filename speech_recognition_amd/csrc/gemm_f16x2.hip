// A/B arm, off by default (kws_net_set_gemm_mode(net, 2); bench.py's ab_gemm_f16x2 leg): the pointwise GEMMs with every f32
// operand scaled by a power of two and split into TWO fp16 parts (x s = h1 + h2: 22 - 24 significand bits, the residual
// is exact in f32) and THREE f16 MFMA products accumulated in f32 (h2.h1, h1.h2, h1.h1; h2.h2 is below the f32 rounding
// of the sum) - closer to float64 than the f32 matrix pipe on every case of tests/test_f16x2_gpu.py.  (Round 2 also carried a
// bf16 x 3 form - three parts, six products: same accuracy, power-limited at 1.4 - 1.5 GHz, slower on every layer; removed.)
// The scale of an operand comes from its |x| maximum (common.h: kws_absmax_commit / kws_absmax_scale): the largest
// magnitude lands in [2^14, 2^15), below fp16's 65504; h1 stays normal for elements down to 2^-28 of the maximum, h2 down
// to 2^-16 of it.  The maxima stay on the device: the kernels that produce an operand (dwconv.hip, bn.hip, absmax_kernel
// here) leave them in 16 words per tensor, the GEMM workgroups read them in their prologue; results are multiplied by
// the two inverse scales on the way out (exact).
//
//   kws_absmax_batch_f32     |x| maxima of arbitrary tensors into slot groups
//   kws_f16x2_split_batch    f32 matrices -> scaled fp16 planes [2][rows][cols] (the pointwise kernels, once per step)
//   kws_gemm_nn_f16x2_f32    C[M,N] = A[M,K] . B, B as planes of [N][K]  (+ BatchNorm column sums per 128-row tile)
//   kws_gemm_tn_f16x2_f32    dW[K,N] = Z[M,K]^T . G[M,N]
// 256 threads = 2 x 2 waves, 32-deep slabs, an unpadded [plane][row][32] fp16 LDS image whose 16-byte chunks are XOR-swizzled
// with (row >> 2) & 3, three workgroups per CU, XCD-contiguous tiles; DESIGN.md section 5 has the measurements.
#include "internal.h"
#include <algorithm>
#include <cstring>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int XBM = 128, XBN = 128, XBK = 32;

// x s -> (h1, h2) per component of a float4, each plane as 4 packed fp16
__device__ __forceinline__ void split4h(const float4 v, const float s, f16x4& h1, f16x4& h2) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 a = (_Float16)x[i];
    h1[i] = a;
    h2[i] = (_Float16)(x[i] - (float)a);              // the residual is exact in f32
  }
}

// -DKWS_X3_STAMP builds (scripts/build_variant.sh h2stamp "-DKWS_X3_STAMP" gemm_f16x2; scripts/stamps_x3.py f16x2): wave 0 of
// the first 2048 tiles accumulates s_memtime deltas: [0] entry -> first slab staged, [1] products, [2] barrier after the
// products, [3] wait for the next slab + split + store, [4] barrier after the store, [5] epilogue, [6] cycles, [7] 100 MHz ticks
#ifdef KWS_X3_STAMP
__device__ unsigned long long g_h2_stamps[2048][8];
extern "C" __attribute__((visibility("default"))) int kws_debug_read_h2_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2_stamps), sizeof(g_h2_stamps));
}
#define X3_DECL unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_mark = __builtin_amdgcn_s_memtime(); \
  const unsigned long long st_t0 = st_mark, st_r0 = __builtin_amdgcn_s_memrealtime();
#define X3_ST(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
  st_acc[i] += n_ - st_mark; st_mark = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define X3_DECL
#define X3_ST(i)
#endif

struct H2Args {
  const float* A;
  const _Float16* Bp;  // [2][N][K], scaled by the power of two its slots give
  const unsigned* a_slots;   // |A| and |B| maxima (KWS_ABSMAX_WORDS words each): A is scaled on the way in,
  const unsigned* b_slots;   // C by the two inverse scales on the way out
  float* C;
  float* stats;
  int64_t M;
  int K, N, n_tiles;
  int64_t tiles, per_xcd, plane_stride;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * XBK + ((chunk ^ ((row >> 2) & 3)) << 3); }

template <bool STATS>
__global__ __launch_bounds__(256, 3) void gemm_nn_f16x2_kernel(H2Args p) {
  __shared__ __attribute__((aligned(16))) _Float16 smem[4 * XBM * XBK];   // A planes, B planes: 32,768 B
  constexpr int PL = XBM * XBK;
  _Float16* sA = smem;
  _Float16* sB = smem + 2 * PL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int K = p.K, N = p.N;
  const int64_t M = p.M;
  const int64_t q = (int64_t)(blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (q >= p.tiles) return;
  float inv_a, inv_b;
  const float s_a = kws_absmax_scale(p.a_slots, inv_a);
  (void)kws_absmax_scale(p.b_slots, inv_b);
  const int64_t tile_m = q / p.n_tiles;
  const int tile_n = (int)(q - tile_m * p.n_tiles);
  const int64_t m0 = tile_m * XBM;
  const int n0 = tile_n * XBN;
  // A loader role: thread t moves float4 t % 8 of rows t / 8 + 32 i (eight lanes = one 128-byte line); B loader role:
  // 16-byte chunk t % 4 of rows t / 4 + 64 i of both planes.  Range-checked buffer loads: ONE offset register per
  // operand, the row / plane / slab steps are scalar offsets; rows past the end read zeros (their results are not stored)
  const int lrow = tid >> 3, lc4 = tid & 7;
  const int brow = tid >> 2, bch = tid & 3;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)(M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Bp), 0, (int)(p.plane_stride * 4), 0x00020000);
  const unsigned voA = ((unsigned)(m0 + lrow) * (unsigned)K + 4u * lc4) * 4u;
  const unsigned voB = ((unsigned)(n0 + brow) * (unsigned)K + 8u * bch) * 2u;
  const int a_off = swz(lrow, lc4 >> 1) + 4 * (lc4 & 1);     // + 32 i rows: (row >> 2) & 3 does not change
  const int b_off = swz(brow, bch);
  float4 ra4[2][4];                                  // TWO slabs of A in flight (HBM latency x bandwidth needs the bytes)
  f16x8 rb8[2][2];                                   // one slab of B planes (L2)
  auto ga_load = [&](const int slot, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, voA, (32 * i * K + k0) * 4, 0);
      ra4[slot][i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto gb_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, voB, (int)(pl * p.plane_stride + 64 * i * K + k0) * 2, 0);
        rb8[i][pl] = __builtin_bit_cast(f16x8, v);
      }
  };
  auto s_store = [&](const int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x4 h1, h2;
      split4h(ra4[slot][i], s_a, h1, h2);
      const int off = a_off + 32 * i * XBK;
      *reinterpret_cast<f16x4*>(sA + off) = h1;
      *reinterpret_cast<f16x4*>(sA + PL + off) = h2;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<f16x8*>(sB + pl * PL + b_off + 64 * i * XBK) = rb8[i][pl];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  // fragment offsets: row = w * 64 + 32 i + r, chunk = 2 s + h
  const int fa = (wm * 64 + r) * XBK, fb = (wn * 64 + r) * XBK;
  const int sw = (r >> 2) & 3;
  const int c0 = ((0 + h) ^ sw) << 3, c1 = ((2 + h) ^ sw) << 3;

  const int G = K / XBK;
  X3_DECL
  ga_load(0, 0);
  gb_load(0);
  ga_load(1, XBK);                                   // G >= 2: K >= 64 is required
  s_store(0);
  __syncthreads();
  X3_ST(0);
  auto products = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cs = s ? c1 : c0;
      f16x8 b[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[j][pl] = *reinterpret_cast<const f16x8*>(sB + pl * PL + fb + 32 * j * XBK + cs);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f16x8 a[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const f16x8*>(sA + pl * PL + fa + 32 * i * XBK + cs);
        // small products first: h2.h1, h1.h2, then h1.h1
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][0], acc[i][j], 0, 0, 0);
        }
      }
    }
#ifdef KWS_X3_STAMP
    asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][1][15]));
#endif
    X3_ST(1);
  };
  // slots and the "more slabs follow" flags are compile-time constants at the call sites: a load under a runtime condition
  // makes the compiler wait for EVERY outstanding load at the next use (its counter model merges the two paths).  B first,
  // then A: the store below waits for B (the younger A requests of slab g + 2 stay in flight).
  auto slab = [&](int g, const int slot_cur, const int slot_next, const bool more1, const bool more2) {
    if (more1) gb_load((g + 1) * XBK);
    if (more2) ga_load(slot_cur, (g + 2) * XBK);     // into the registers slab g has just left
    __builtin_amdgcn_sched_barrier(0);               // the requests go out HERE, not where the scheduler finds room
    products();
    if (more1) {
      __syncthreads();                               // every wave has read this slab's planes
      X3_ST(2);
      s_store(slot_next);
      X3_ST(3);
      __syncthreads();
      X3_ST(4);
    }
  };
  for (int g = 0; g < G - 2; g += 2) {               // G is even (K % 64 == 0)
    slab(g, 0, 1, true, true);
    slab(g + 1, 1, 0, true, true);
  }
  slab(G - 2, 0, 1, true, false);
  slab(G - 1, 1, 0, false, false);
  // Epilogue.  The accumulators hold a column per lane (row = (v & 3) + 8 (v >> 2) + 4 h): stored as they are that is 64
  // four-byte store instructions per wave, and the store queue, not the bandwidth, paces the tile (36 % of a workgroup's
  // life at K = 128 by the stamps).  Instead each wave turns its tile through its own 8 KB of the (now dead) slab image,
  // 32 rows at a time, and writes rows as 16-byte pieces: 16 store instructions of four 256-byte row segments each.
  float cs[2] = {0.f, 0.f}, css[2] = {0.f, 0.f};
  __syncthreads();                                   // every wave has read the last slab's planes
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 64);
  const float unscale = inv_a * inv_b;               // exact: powers of two
  const int sr = lane >> 4, sc4 = lane & 15;         // read-back role: row sr + 4 t, columns 4 sc4 .. 4 sc4 + 3
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float c = acc[i][j][v] * unscale;
        stage[((v & 3) + 8 * (v >> 2) + 4 * h) * 64 + 32 * j + r] = c;
        if (STATS && m0 + wm * 64 + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * h < M) {   // BatchNorm column sums in a fixed
          cs[j] += c;                                // order: rows of the lane, then the two lane halves, then the two row waves
          css[j] = fmaf(c, c, css[j]);
        }
      }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 o = *reinterpret_cast<const float4*>(stage + (sr + 4 * t) * 64 + 4 * sc4);
      const int64_t row = m0 + wm * 64 + 32 * i + sr + 4 * t;
      const int col = n0 + wn * 64 + 4 * sc4;
      if (row < M && col < N) *reinterpret_cast<float4*>(p.C + row * N + col) = o;
    }
  }
  if (STATS) __syncthreads();                        // the statistics rows below reuse the first 2 KB of the image
  if (STATS) {
    float* red = reinterpret_cast<float*>(smem);     // [2 wm][2 q][128]: the planes are dead after the last barrier
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float s2 = cs[j] + __shfl_xor(cs[j], 32), q2 = css[j] + __shfl_xor(css[j], 32);
      if (h == 0) {
        red[(wm * 2 + 0) * 128 + wn * 64 + 32 * j + r] = s2;
        red[(wm * 2 + 1) * 128 + wn * 64 + 32 * j + r] = q2;
      }
    }
    __syncthreads();
    const int qq = tid >> 7, col = tid & 127;
    if (n0 + col < N) p.stats[((int64_t)tile_m * 2 + qq) * N + n0 + col] = red[(0 * 2 + qq) * 128 + col] + red[(1 * 2 + qq) * 128 + col];
  }
#ifdef KWS_X3_STAMP
  X3_ST(5);
  if (tid == 0 && q < 2048) {
    for (int i = 0; i < 6; ++i) g_h2_stamps[q][i] = st_acc[i];
    g_h2_stamps[q][6] = __builtin_amdgcn_s_memtime() - st_t0;
    g_h2_stamps[q][7] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
}


constexpr int KWS_SPLIT_BATCH = 24;                  // matrices / tensors per batched launch
// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient in the same arithmetic: dW[K,N] = Z[M,K]^T . G[M,N], the reduction runs over the rows.  Both operands are
// activations, so both are scaled and split on the way into LDS, and both are staged TRANSPOSED ([k or n][32 m], the image and
// swizzle of the forward kernel): a thread loads a 4 (m) x 4 (k) block - eight lanes cover a 128-byte line of one row -
// and writes, per k, the four m values of each plane as one 8-byte store.  Workgroup tile 64 IC x 64 JC (2 x 2 waves),
// one (tile, row chunk) item per workgroup, partial tiles into [S][K][N] slabs summed by kws_reduce_slabs_f32 in a fixed
// order.  Rows past M read as zero (range-checked buffer loads).
struct U2Args {
  const float* Z;
  const float* G;
  float* ws;
  const unsigned* z_slots;
  const unsigned* g_slots;
  int64_t M, chunk, items, per_xcd;
  int K, N, n_tiles, tiles;
};

template <int IC, int JC>
__global__ __launch_bounds__(256, (IC * JC == 4) ? 3 : 4) void gemm_tn_f16x2_kernel(U2Args p) {
  constexpr int BKO = 64 * IC, BNO = 64 * JC;
  constexpr int PLA = BKO * XBK, PLB = BNO * XBK;
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * PLA + 2 * PLB];
  _Float16* sA = smem;
  _Float16* sB = smem + 2 * PLA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int K = p.K, N = p.N;
  const int64_t q = (int64_t)(blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (q >= p.items) return;
  float inv_z, inv_g;
  const float s_z = kws_absmax_scale(p.z_slots, inv_z), s_g = kws_absmax_scale(p.g_slots, inv_g);
  const int64_t sp = q / p.tiles;                    // consecutive items = the tiles of one row chunk: Z and G rows are
  const int tile = (int)(q - sp * p.tiles);          // fetched from HBM once per XCD
  const int k0 = (tile / p.n_tiles) * BKO, n0 = (tile % p.n_tiles) * BNO;
  const int64_t m_begin = sp * p.chunk;
  const int64_t m_end = m_begin + p.chunk < p.M ? m_begin + p.chunk : p.M;
  const int slabs = (int)((m_end - m_begin + XBK - 1) / XBK);
  // loader roles: lane -> (column quad lane & 7, row quad lane >> 3), wave -> column octet; 64-wide operands take two
  // waves (Z waves 0-1, G the other pair when both are narrow).  An idle role loads from beyond the buffer: zeros, no
  // traffic, and no load sits under a condition (the compiler would wait for ALL outstanding loads at the next use)
  const int mg = lane >> 3;
  const bool z_act = IC == 2 || wave < 2;
  const bool g_act = JC == 2 || (IC == 2 ? wave < 2 : wave >= 2);
  const int zq = 8 * (wave & (IC == 2 ? 3 : 1)) + (lane & 7);
  const int gq = 8 * (wave & (JC == 2 ? 3 : 1)) + (lane & 7);
  const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Z), 0, (int)(p.M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.G), 0, (int)(p.M * N * 4), 0x00020000);
  const unsigned row0 = (unsigned)(m_begin + 4 * mg);
  unsigned zoff = z_act ? (row0 * (unsigned)K + (unsigned)(k0 + 4 * zq)) * 4u : 0x80000000u;
  unsigned goff = g_act ? (row0 * (unsigned)N + (unsigned)(n0 + 4 * gq)) * 4u : 0x80000000u;
  const int z_st = (4 * zq) * XBK + 4 * (mg & 1), g_st = (4 * gq) * XBK + 4 * (mg & 1);   // + row j, + swizzled chunk
  const int z_ch = ((mg >> 1) ^ (zq & 3)) << 3, g_ch = ((mg >> 1) ^ (gq & 3)) << 3;      // (row >> 2) & 3 = quad & 3
  float4 rz[4], rg[4];
  auto g_load = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsZ, zoff + (unsigned)(i * K * 4), 0, 0);
      rz[i] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsG, goff + (unsigned)(i * N * 4), 0, 0);
      rg[i] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    }
    zoff += (unsigned)(XBK * K * 4);
    goff += (unsigned)(XBK * N * 4);
  };
  auto t_store = [&](const float4* rr, const float sc, _Float16* sX, const int PL, const int st, const int ch) {
    const float v[4][4] = {{rr[0].x, rr[0].y, rr[0].z, rr[0].w}, {rr[1].x, rr[1].y, rr[1].z, rr[1].w},
                           {rr[2].x, rr[2].y, rr[2].z, rr[2].w}, {rr[3].x, rr[3].y, rr[3].z, rr[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {                    // column j of the block: its four rows are consecutive m
      f16x4 h1, h2;
      split4h(make_float4(v[0][j], v[1][j], v[2][j], v[3][j]), sc, h1, h2);
      const int off = st + j * XBK + ch;
      *reinterpret_cast<f16x4*>(sX + off) = h1;
      *reinterpret_cast<f16x4*>(sX + PL + off) = h2;
    }
  };
  auto s_store = [&]() {
    if (z_act) t_store(rz, s_z, sA, PLA, z_st, z_ch);
    if (g_act) t_store(rg, s_g, sB, PLB, g_st, g_ch);
  };
  f32x16 acc[IC][JC];
#pragma unroll
  for (int i = 0; i < IC; ++i)
#pragma unroll
    for (int j = 0; j < JC; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  const int fa = (wm * 32 * IC + r) * XBK, fb = (wn * 32 * JC + r) * XBK;
  const int sw = (r >> 2) & 3;
  const int c0 = ((0 + h) ^ sw) << 3, c1 = ((2 + h) ^ sw) << 3;
  auto products = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cs = s ? c1 : c0;
      f16x8 b[JC][2];
#pragma unroll
      for (int j = 0; j < JC; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[j][pl] = *reinterpret_cast<const f16x8*>(sB + pl * PLB + fb + 32 * j * XBK + cs);
#pragma unroll
      for (int i = 0; i < IC; ++i) {
        f16x8 a[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[pl] = *reinterpret_cast<const f16x8*>(sA + pl * PLA + fa + 32 * i * XBK + cs);
#pragma unroll
        for (int j = 0; j < JC; ++j) {               // small products first, as in the forward kernel
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][0], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  g_load();
  s_store();
  __syncthreads();
  for (int g = 0; g < slabs - 1; ++g) {
    g_load();                                        // slab g + 1
    products();
    __syncthreads();                                 // every wave has read slab g
    s_store();
    __syncthreads();
  }
  products();
  float* out = p.ws + sp * ((int64_t)K * N);
#pragma unroll
  for (int i = 0; i < IC; ++i)
#pragma unroll
    for (int j = 0; j < JC; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = k0 + wm * 32 * IC + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * h;
        const int col = n0 + wn * 32 * JC + 32 * j + r;
        out[(int64_t)row * N + col] = acc[i][j][v] * inv_z * inv_g;      // exact: powers of two
      }
}

struct U2Plan {
  int ic, jc, tiles, n_tiles, S;
  int64_t chunk;
};
U2Plan u2_plan(int64_t M, int K, int N) {
  U2Plan pl;
  pl.ic = K % 128 == 0 ? 2 : 1;
  pl.jc = N % 128 == 0 ? 2 : 1;
  pl.n_tiles = N / (64 * pl.jc);
  pl.tiles = (K / (64 * pl.ic)) * pl.n_tiles;
  const int slots = (pl.ic * pl.jc == 4) ? 768 : 1024;   // three / four workgroups per CU
  int64_t S = slots / pl.tiles;
  const int64_t maxS = M / 64 > 1 ? M / 64 : 1;          // at least two 32-row slabs per item
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  pl.chunk = ceil_div64(ceil_div64(M, S), XBK) * XBK;
  pl.S = (int)ceil_div64(M, pl.chunk);
  return pl;
}

// |x| maxima of up to KWS_SPLIT_BATCH tensors into consecutive slot groups (KWS_ABSMAX_WORDS words each, zeroed first)
struct AbsmaxBatch {
  const float* in[KWS_SPLIT_BATCH];
  int64_t n[KWS_SPLIT_BATCH];
  unsigned* slots;
};
__global__ __launch_bounds__(256) void absmax_kernel(AbsmaxBatch b) {
  const int e = blockIdx.y;
  const float* in = b.in[e];
  const int64_t n = b.n[e];
  float m = 0.f;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(in[o]));
  kws_absmax_commit(b.slots + (int64_t)e * KWS_ABSMAX_WORDS, m);
}

// f32 matrices [rows][cols], scaled by the power of two their slots give -> fp16 planes [2][rows][cols] (or of the transpose)
struct SplitBatchH {
  const float* in[KWS_SPLIT_BATCH];
  _Float16* out[KWS_SPLIT_BATCH];
  int rows[KWS_SPLIT_BATCH], cols[KWS_SPLIT_BATCH], transpose[KWS_SPLIT_BATCH];
  const unsigned* slots[KWS_SPLIT_BATCH];
};

__global__ __launch_bounds__(256) void split_planes_h_kernel(SplitBatchH b) {
  const int e = blockIdx.y;
  const int R = b.rows[e], Cn = b.cols[e];
  const int64_t n = (int64_t)R * Cn;
  const float* in = b.in[e];
  _Float16* out = b.out[e];
  const bool tr = b.transpose[e] != 0;
  float inv;
  const float s = kws_absmax_scale(b.slots[e], inv);
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
    float x;
    if (tr) {
      const int64_t oc = o / R, orow = o - oc * R;
      x = in[orow * Cn + oc];
    } else {
      x = in[o];
    }
    x *= s;
    const _Float16 h1 = (_Float16)x;
    out[o] = h1;
    out[n + o] = (_Float16)(x - (float)h1);
  }
}

}  // namespace

extern "C" int kws_f16x2_split_batch(const float* const* in, void* const* out, const int* rows, const int* cols,
                                     const int* transpose, const unsigned* const* slots, int count, void* stream) {
  KWS_REQUIRE(in && out && rows && cols && transpose && slots && count > 0 && count <= KWS_SPLIT_BATCH, "f16x2_split_batch: bad arguments (count=%d)", count);
  SplitBatchH b;
  memset(&b, 0, sizeof(b));
  int64_t biggest = 0;
  for (int i = 0; i < count; ++i) {
    KWS_REQUIRE(in[i] && out[i] && slots[i] && rows[i] > 0 && cols[i] > 0, "f16x2_split_batch: bad matrix %d", i);
    b.in[i] = in[i]; b.out[i] = (_Float16*)out[i]; b.rows[i] = rows[i]; b.cols[i] = cols[i]; b.transpose[i] = transpose[i];
    b.slots[i] = slots[i];
    biggest = std::max<int64_t>(biggest, (int64_t)rows[i] * cols[i]);
  }
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(biggest, 256), 256);
  hipLaunchKernelGGL(split_planes_h_kernel, dim3(gx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, b);
  KWS_LAUNCH_CHECK("split_planes_h_kernel");
  return KWS_OK;
}

// shapes the arm's kernels take; the network programs run the f32 kernels for the rest (K granule of the two-slab pipeline,
// 32-bit byte offsets inside one buffer view: a per-GPU batch of ~8 k clips at block 0 is past it)
extern "C" int kws_gemm_nn_f16x2_supported(int64_t M, int K, int N) {
  return M > 0 && K >= 2 * XBK && K % (2 * XBK) == 0 && N > 0 && N % 4 == 0 && (M + XBM) * (int64_t)K * 4 < (1ll << 31) &&
         (int64_t)(N + XBN) * K * 4 < (1ll << 31);
}
extern "C" int kws_gemm_tn_f16x2_supported(int64_t M, int K, int N) {
  return M > 0 && K > 0 && N > 0 && K % 64 == 0 && N % 64 == 0 && M * (int64_t)std::max(K, N) * 4 < (1ll << 31);
}
extern "C" int kws_gemm_nn_f16x2_stats_rows(int64_t M) { return (int)ceil_div64(M, XBM); }   // one per 128-row tile

// C[M,N] = A[M,K] . B with B given as the fp16 planes of (scaled) B stored [N][K]; a_slots / b_slots: the |A| and |B|
// maxima (kws_absmax_batch_f32, or the network's producing kernels)
extern "C" int kws_gemm_nn_f16x2_f32(const float* A, const void* Bp, float* C, int64_t M, int K, int N,
                                     const unsigned* a_slots, const unsigned* b_slots, float* stats_part, void* stream) {
  KWS_REQUIRE(A && Bp && C && a_slots && b_slots && M > 0, "gemm_nn_f16x2: bad arguments");
  KWS_REQUIRE(kws_gemm_nn_f16x2_supported(M, K, N), "gemm_nn_f16x2: unsupported shape M=%lld K=%d N=%d (K a multiple of %d, N of 4, operands within the 2 GB buffer view: kws_gemm_nn_f16x2_supported)", (long long)M, K, N, 2 * XBK);
  H2Args p;
  p.A = A; p.Bp = (const _Float16*)Bp; p.C = C; p.stats = stats_part; p.M = M; p.K = K; p.N = N;
  p.a_slots = a_slots; p.b_slots = b_slots;
  p.n_tiles = (N + XBN - 1) / XBN;
  p.tiles = ceil_div64(M, XBM) * p.n_tiles;
  p.per_xcd = ceil_div64(p.tiles, 8);
  p.plane_stride = (int64_t)N * K;
  const int64_t grid = p.per_xcd * 8;
  KWS_REQUIRE(grid <= 0x7FFFFFFF, "gemm_nn_f16x2: grid out of range");
  KwsProfScope prof("gemm_nn_f16x2", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)K * N + (double)M * N), (hipStream_t)stream);
  if (stats_part) hipLaunchKernelGGL(gemm_nn_f16x2_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_nn_f16x2_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  KWS_LAUNCH_CHECK("gemm_nn_f16x2_kernel");
  return KWS_OK;
}

// |x| maxima of count <= 24 tensors: slots = count consecutive groups of KWS_ABSMAX_WORDS (256) words; zeroed here
extern "C" int kws_absmax_batch_f32(const float* const* in, const int64_t* n, unsigned* slots, int count, void* stream) {
  KWS_REQUIRE(in && n && slots && count > 0 && count <= KWS_SPLIT_BATCH, "absmax_batch: bad arguments (count=%d)", count);
  AbsmaxBatch b;
  memset(&b, 0, sizeof(b));
  int64_t biggest = 0;
  for (int i = 0; i < count; ++i) {
    KWS_REQUIRE(in[i] && n[i] > 0, "absmax_batch: bad tensor %d", i);
    b.in[i] = in[i]; b.n[i] = n[i];
    biggest = std::max(biggest, n[i]);
  }
  b.slots = slots;
  hipStream_t st = (hipStream_t)stream;
  KWS_HIP(hipMemsetAsync(slots, 0, (size_t)count * KWS_ABSMAX_WORDS * sizeof(unsigned), st));
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(biggest, 1024), 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(gx, (unsigned)count), dim3(256), 0, st, b);
  KWS_LAUNCH_CHECK("absmax_kernel");
  return KWS_OK;
}

// dW[K,N] = Z[M,K]^T . G[M,N] in the fp16 x 2 arithmetic (K, N multiples of 64); z_slots / g_slots: the operands' maxima
extern "C" int64_t kws_gemm_tn_f16x2_workspace_floats(int64_t M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || K % 64 || N % 64) return 0;
  return (int64_t)u2_plan(M, K, N).S * K * N;
}

extern "C" int kws_gemm_tn_f16x2_f32(const float* Z, const float* G, float* dW, int64_t M, int K, int N,
                                     const unsigned* z_slots, const unsigned* g_slots, float* workspace, void* stream) {
  KWS_REQUIRE(Z && G && dW && z_slots && g_slots && workspace && M > 0, "gemm_tn_f16x2: bad arguments");
  KWS_REQUIRE(kws_gemm_tn_f16x2_supported(M, K, N), "gemm_tn_f16x2: unsupported shape M=%lld K=%d N=%d (K, N multiples of 64, operands within the 2 GB buffer view: kws_gemm_tn_f16x2_supported)", (long long)M, K, N);
  const U2Plan pl = u2_plan(M, K, N);
  U2Args p;
  p.Z = Z; p.G = G; p.ws = workspace; p.z_slots = z_slots; p.g_slots = g_slots; p.M = M; p.chunk = pl.chunk; p.K = K; p.N = N;
  p.n_tiles = pl.n_tiles; p.tiles = pl.tiles;
  p.items = (int64_t)pl.tiles * pl.S;
  p.per_xcd = ceil_div64(p.items, 8);
  const dim3 grid((unsigned)(p.per_xcd * 8)), block(256);
  hipStream_t st = (hipStream_t)stream;
  KwsProfScope prof("gemm_tn_f16x2", 2.0 * M * K * N, 4.0 * ((double)M * K + (double)M * N + (double)K * N), st);
  if (pl.ic == 2 && pl.jc == 2) hipLaunchKernelGGL((gemm_tn_f16x2_kernel<2, 2>), grid, block, 0, st, p);
  else if (pl.ic == 2) hipLaunchKernelGGL((gemm_tn_f16x2_kernel<2, 1>), grid, block, 0, st, p);
  else if (pl.jc == 2) hipLaunchKernelGGL((gemm_tn_f16x2_kernel<1, 2>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_tn_f16x2_kernel<1, 1>), grid, block, 0, st, p);
  KWS_LAUNCH_CHECK("gemm_tn_f16x2_kernel");
  return kws_reduce_slabs_f32(workspace, dW, (int64_t)K * N, pl.S, st);
}
