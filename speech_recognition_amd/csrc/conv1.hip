// First convolution of the raw-waveform net as a Toeplitz GEMM (reference model.py:805-807: overlapping_time_slice_stack
// (40, 20) + Conv1D(128, 3, strides=2); net.hip folds the three overlapping taps into ONE tap of KF = 80 samples).
//
// A[(b, t), k] = x[b, 40 t - 10 + k] is not a matrix in memory: consecutive rows overlap by half and the first row of
// every clip hangs over its start.  The generic gathered GEMMs (gemm.hip) pay for that with per-element address
// arithmetic inside a K loop that is only 2.5 slabs long (forward 160 us = 52 TFLOP/s, weight gradient 212 us = 40 TFLOP/s
// at batch 1024).  These two kernels are shaped for this one operand instead:
//   * the whole K extent (80) of a row tile is staged at once, two threads per row reading 8-byte pairs (the rows are
//     8-byte aligned: stride 40, offset -10), zero filled outside the clip; no K loop, no per-slab barrier;
//   * forward: persistent workgroups hold the folded [80, 128] kernel in REGISTERS for all their tiles (see the kernel),
//     next tile's rows in flight in registers while the current one is multiplied; one BN statistics row per workgroup;
//   * weight gradient: dW[80, 128] = A^T G split over M into one slab per workgroup (fixed-order slab sum afterwards),
//     32-row units; wave w owns output columns [32 w, 32 w + 32) x 80 rows (ten 16 x 16 accumulators, v_mfma_f32_16x16x4_f32).
//     Measured alone at batch 1024: 64-row units with A and G in LDS and one workgroup per CU 157 us, 32-row units two
//     per CU 127 us, G straight into registers and three per CU 111 us (the generic gathered kernel: 156 us) - those with
//     three 32 x 32 blocks per wave whose rows 80..95 multiplied LDS zeros; round 3 see the kernel.
// Three things the compiler had to be told (each visible in the ISA): selecting between a row pointer and a `const`
// zero buffer makes the loads FLAT loads (the zero buffer is a plain __device__ array); the next tile's global loads
// sink below the MFMA loop unless a memory clobber pins them; the LDS operand reads are issued one pair at a time with
// lgkmcnt(0) in front of every two MFMAs unless the next K group's reads are written out and fenced with
// __builtin_amdgcn_sched_barrier (168 -> 157 us).
#include "internal.h"

// -DKWS_C1_WS=1 builds the forward kernel with the pointwise GEMMs' wave roles (round 6's experiment, conv1_fwd_ws_kernel below: bit-identical
// outputs, measured SLOWER - profiles/r06_conv1_ws.txt - so the shipped library keeps the four-wave kernel)
#ifndef KWS_C1_WS
#define KWS_C1_WS 0
#endif
// -DKWS_C1_ABL=<bits> builds (timing only, wrong results) of the forward kernels: without 1 the output stores, 2 the BN statistics
// sums, 4 the MFMAs, 8 the global row loads, 16 (wave-role kernel) the MFMA waves' staging writes
#ifndef KWS_C1_ABL
#define KWS_C1_ABL 0
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NOUT = 128;   // output channels (filter_mult 1)
constexpr int FM = 64;      // forward row tile
constexpr int UM = 32;      // weight-gradient row unit
#ifndef KWS_C1W_OCC
#define KWS_C1W_OCC 3      // weight-gradient workgroups per CU (one round: chunk = M / (256 x this))
#endif
#ifndef KWS_C1W_ABL
#define KWS_C1W_ABL 0      // timing-only ablation (wrong results): 1 no MFMAs, 2 no global loads
#endif

struct Conv1Args {
  const float* x;
  const float* W;      // forward: folded kernel [KF, 128]
  float* y;            // forward: [M, 128]
  float* stats;        // forward: [m_tiles][2][128] or NULL
  const float* G;      // wgrad: dy [M, 128]
  float* ws;           // wgrad: [S][KF][128]
  kws_gather_t g;
  int taps, cin, hop;  // the unfolded kernel W[taps][cin][128]: Weff[s] = sum_j W[j][s - hop j] (taps = 1: W is folded)
  int B, m_tiles, S;
  int64_t M, chunk;
};

// Rows that lie fully inside their clip (all but the first row of a clip) are read with unconditional 8-byte loads; the
// others - and the rows past M - read this zero buffer instead, and only a row that straddles the clip start is patched
// element by element.  (A per-load `inside ? load : 0` compiles to a branch and a vmcnt(0) per load.)
__device__ __attribute__((aligned(16))) float g_zero64[128] = {0.f};   // (>= one whole 80-sample row) not const: a constant-address-space operand turns the selected loads into flat loads

__device__ __forceinline__ float2 load2_or_zero(const float* xb, int pos, int x_len) {
  if (pos >= 0 && pos + 1 < x_len) return *reinterpret_cast<const float2*>(xb + pos);
  float2 v = make_float2(0.f, 0.f);
  if (pos >= 0 && pos < x_len) v.x = xb[pos];
  if (pos + 1 >= 0 && pos + 1 < x_len) v.y = xb[pos + 1];
  return v;
}

// Forward.  64-row tiles, wave (wr, wc) = (wave / 2, wave % 2) computes rows [32 wr, 32 wr + 32) x columns [64 wc, 64 wc + 64).
// The folded kernel never touches LDS: lane (li, lh) needs W[8 q + 4 lh + r][64 wc + 32 jj + li] for every K group q -
// 80 values that stay in registers for all tiles of the persistent workgroup, so the MFMA loop reads ONE 16-byte LDS
// operand per 8 MFMAs.  43 KB of LDS (double-buffered rows) -> three workgroups per CU hide each other's epilogues and
// loads.  BN statistics: every wave keeps running column sums over all its tiles; ONE row per workgroup at the end.
template <int KF, bool STATS>
__global__ __launch_bounds__(256, 3) void conv1_fwd_kernel(Conv1Args p) {
  constexpr int PA = KF + 4;        // LDS row pitch of the staged rows (16-byte aligned fragments)
  constexpr int QF = KF / 4;        // floats per staging thread (4 threads per row)
  constexpr int NL = QF / 2;        // 8-byte loads per staging thread
  constexpr int NQ = KF / 8;
  __shared__ float sA[2][FM * PA];
  __shared__ float sRed[2][2][NOUT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  // Fold on load.  Every tap's candidate is requested unconditionally (the address of a tap that does not reach sample s is
  // that of a zero buffer) and the taps are added in tap order: the 24 loads of a K group are in flight together.  Written as
  // `for (j < taps) if (inside) w += W[..]` the compiler made 80 loops of load -> s_waitcnt vmcnt(0) -> add: ~120 DEPENDENT
  // L2 round trips per lane before the first MFMA of every one of the 768 workgroups (round 3: 126 -> see DESIGN.md section 5).
  float wreg[NQ][4][2];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float wv[3][4][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int sidx = q * 8 + lh * 4 + r, n = wc * 64 + jj * 32 + li;
          const int c = sidx - p.hop * j;
          const bool ok = j < p.taps && c >= 0 && c < p.cin;
          const float* src = ok ? p.W + ((int64_t)j * p.cin + c) * NOUT + n : g_zero64;
          wv[j][r][jj] = *src;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        wreg[q][r][jj] = ((0.f + wv[0][r][jj]) + wv[1][r][jj]) + wv[2][r][jj];   // = fold_taps_kernel's order
        asm volatile("" : "+v"(wreg[q][r][jj]));    // folded HERE: one K group's 24 candidates live at a time (hoisted across
      }                                             // the groups, the 240 loads spilled)
    __builtin_amdgcn_sched_barrier(0);
  }
  const int srow = tid >> 2, sq = tid & 3;
  float2 ra[NL];
  auto load_rows = [&](int tile) {
    const int64_t m = (int64_t)tile * FM + srow;
    const bool row_ok = m < p.M;
    const int64_t b = row_ok ? m / p.g.L_out : 0;
    const int t = row_ok ? (int)(m - b * p.g.L_out) : 0;
    const float* xb = p.x + b * p.g.x_batch_stride;
    const int e0 = t * p.g.stride_t + p.g.base_off + sq * QF;
    const bool inside = row_ok && e0 >= 0 && e0 + QF <= p.g.x_len;
    const float* src = inside ? xb + e0 : g_zero64;
#pragma unroll
    for (int i = 0; i < NL; ++i) ra[i] = *reinterpret_cast<const float2*>(src + 2 * i);
    if (row_ok && !inside) {
#pragma unroll
      for (int i = 0; i < NL; ++i) ra[i] = load2_or_zero(xb, e0 + 2 * i, p.g.x_len);
    }
  };
  auto store_rows = [&](int buf) {
    float* dst = &sA[buf][srow * PA + sq * QF];
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<float2*>(dst + 2 * i) = ra[i];
  };
  float st_s[2] = {0.f, 0.f}, st_ss[2] = {0.f, 0.f};
  int tile = blockIdx.x;
  if (tile < p.m_tiles) {
    load_rows(tile);
    store_rows(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < p.m_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool has_next = next < p.m_tiles;
    if (has_next && !(KWS_C1_ABL & 8)) load_rows(next);
    asm volatile("" ::: "memory");   // the loads are issued HERE (the scheduler otherwise sinks them below the MFMA loop)
    f32x16 acc[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[jj][v] = 0.f;
    const float* cA = &sA[buf][(wr * 32 + li) * PA + lh * 4];
    // operand reads run two K groups ahead of the MFMAs (pinned: left alone the compiler reads one operand, waits, issues)
    float4 av[3];
    av[0] = *reinterpret_cast<const float4*>(cA);
    av[1] = *reinterpret_cast<const float4*>(cA + 8);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + 2 < NQ) av[(q + 2) % 3] = *reinterpret_cast<const float4*>(cA + (q + 2) * 8);
      __builtin_amdgcn_sched_barrier(0);
      const float4 ac = av[q % 3];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = r == 0 ? ac.x : (r == 1 ? ac.y : (r == 2 ? ac.z : ac.w));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (KWS_C1_ABL & 4) { acc[jj][(q * 4 + r) & 15] += a * wreg[q][r][jj]; continue; }
          acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[q][r][jj], acc[jj], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: a store instruction covers 2 rows x 32 consecutive columns (two 128-byte segments)
    const int64_t m0 = (int64_t)tile * FM + wr * 32 + 4 * lh;
    if (KWS_C1_ABL & 1) {
      if (acc[0][0] == 123.456f) p.y[m0] = acc[1][5];
    } else if ((int64_t)(tile + 1) * FM <= p.M) {   // whole tile inside: no per-row test
      float* y0 = p.y + m0 * NOUT + wc * 64 + li;
#pragma unroll
      for (int v = 0; v < 16; ++v)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) y0[((v & 3) + 8 * (v >> 2)) * NOUT + jj * 32] = acc[jj][v];
    } else {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t m = m0 + (v & 3) + 8 * (v >> 2);
        if (m < p.M) {
          float* yr = p.y + m * NOUT + wc * 64 + li;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) yr[jj * 32] = acc[jj][v];
        }
      }
    }
    if (STATS && !(KWS_C1_ABL & 2)) {   // rows past M were staged as zeros: they add nothing
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          st_s[jj] += acc[jj][v];
          st_ss[jj] = fmaf(acc[jj][v], acc[jj][v], st_ss[jj]);
        }
    }
    if (has_next) store_rows(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (STATS) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const float s = st_s[jj] + __shfl_xor(st_s[jj], 32), ss = st_ss[jj] + __shfl_xor(st_ss[jj], 32);
      if (lh == 0) {
        sRed[0][wr][wc * 64 + jj * 32 + li] = s;
        sRed[1][wr][wc * 64 + jj * 32 + li] = ss;
      }
    }
    __syncthreads();
    const int q = tid >> 7, c = tid & 127;
    p.stats[((int64_t)blockIdx.x * 2 + q) * NOUT + c] = sRed[q][0][c] + sRed[q][1][c];
  }
}

#if KWS_C1_WS
// Round 6 EXPERIMENT (variant builds only, -DKWS_C1_WS=1; measured a LOSS: 104.5 - 106 us against the 92 - 94 us of conv1_fwd_kernel,
// profiles/r06_conv1_ws.txt): the forward kernel with the pointwise GEMMs' WAVE ROLES (gemm.hip gemm_nn_ws_kernel).  Round 5's ablations of the
// kernel above (profiles/r05_conv1_fwd_ablation.txt): the MFMA loop with its LDS reads alone 75.3 us, the shipped kernel 94.5 - every
// non-MFMA instruction in an MFMA wave's stream (32 output stores, 64 statistics operations, 10 row loads and LDS writes per tile) costs
// matrix cycles, and software-pipelining them inside the same four waves ran out of registers at three workgroups per CU.  Here ONE
// 8-wave workgroup per CU:
//   waves 0-3  MFMA only: the same (wr, wc) blocks, the same folded kernel in registers, the same k order - with the two MFMA
//              operands SWAPPED (D = W^T-fragment x row-fragment: the products and their order are unchanged, every output element
//              is bit-identical) so that a lane's four consecutive accumulator registers are four consecutive COLUMNS of one output
//              row: a finished tile leaves as 8 ds_write_b128 per lane into an LDS staging tile.  Two accumulator sets: tile t's set
//              is staged after the first K group of tile t + 1 has been issued, so the matrix pipe never drains;
//   waves 4-5  loaders: wave w stages the rows of its tiles t = w (mod 2) - lane = row, 40 8-byte loads, issued two iterations
//              before they are written to LDS;
//   waves 6-7  storers: the staging tile of iteration t - 2 to global memory as 16-byte row stores (a wave instruction = two whole
//              512-byte rows) and the BatchNorm column sums on the way (running sums per thread over all tiles, folded in a fixed
//              order at the end: ONE statistics row per workgroup, as before - the same sums in another fixed order).
// One barrier per tile.  LDS: 2 x 21.5 KB rows + 2 x 33.8 KB staging.
template <int KF, bool STATS>
__global__ __launch_bounds__(512, 1) void conv1_fwd_ws_kernel(Conv1Args p) {
  constexpr int PA = KF + 4;        // LDS row pitch of the staged rows
  constexpr int NQ = KF / 8;
  constexpr int NL = KF / 2;        // 8-byte loads per loader lane (one whole row)
  constexpr int SLD = NOUT + 4;     // staging row pitch: conflict-free b128 writes
  __shared__ __attribute__((aligned(16))) float sA[2][FM * PA];
  __shared__ __attribute__((aligned(16))) float sC[2][FM * SLD];
  __shared__ float sRed[2][4][NOUT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n_my = (p.m_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1: the grid is at most m_tiles
  // iterations it = 0 .. n_my + 1: MFMA waves multiply local tile it (it < n_my) and stage tile it - 1 (1 <= it <= n_my); the
  // storers move tile it - 2 (it >= 2); a loader writes tile it + 1 (its own) and re-issues.  Barriers per wave: 1 + n_my + 2 (+ 1)
  if (wave < 4) {
    // ---------------------------------------------------------------- MFMA waves
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    float wreg[NQ][4][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {                  // fold on load: as conv1_fwd_kernel (same candidates, same order of the adds)
      float wv[3][4][2];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int sidx = q * 8 + lh * 4 + r, n = wc * 64 + jj * 32 + li;
            const int c = sidx - p.hop * j;
            const bool ok = j < p.taps && c >= 0 && c < p.cin;
            const float* src = ok ? p.W + ((int64_t)j * p.cin + c) * NOUT + n : g_zero64;
            wv[j][r][jj] = *src;
          }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          wreg[q][r][jj] = ((0.f + wv[0][r][jj]) + wv[1][r][jj]) + wv[2][r][jj];
          asm volatile("" : "+v"(wreg[q][r][jj]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 accA[2], accB[2];
    // stage a finished set: register group vq of block jj = columns 64 wc + 32 jj + 8 vq + 4 lh .. + 3 of row 32 wr + li
    auto stage = [&](const f32x16 (&acc)[2], int buf) {
      if (KWS_C1_ABL & 16) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][5])); return; }
      float* dst = &sC[buf][(wr * 32 + li) * SLD + wc * 64 + 4 * lh];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int vq = 0; vq < 4; ++vq)
          *reinterpret_cast<float4*>(dst + jj * 32 + 8 * vq) =
              make_float4(acc[jj][4 * vq], acc[jj][4 * vq + 1], acc[jj][4 * vq + 2], acc[jj][4 * vq + 3]);
    };
    auto iteration = [&](int it, f32x16 (&cur)[2], const f32x16 (&prev)[2]) {
      if (it < n_my) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int v = 0; v < 16; ++v) cur[jj][v] = 0.f;
        const float* cA = &sA[it & 1][(wr * 32 + li) * PA + lh * 4];
        float4 av[3];
        av[0] = *reinterpret_cast<const float4*>(cA);
        av[1] = *reinterpret_cast<const float4*>(cA + 8);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q + 2 < NQ) av[(q + 2) % 3] = *reinterpret_cast<const float4*>(cA + (q + 2) * 8);
          __builtin_amdgcn_sched_barrier(0);
          const float4 ac = av[q % 3];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = r == 0 ? ac.x : (r == 1 ? ac.y : (r == 2 ? ac.z : ac.w));
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              if (KWS_C1_ABL & 4) { cur[jj][(q * 4 + r) & 15] += a * wreg[q][r][jj]; continue; }
              cur[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[q][r][jj], a, cur[jj], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (q == 0 && it >= 1) {                  // the previous tile's set, under the MFMAs just issued
            stage(prev, (it - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (it == n_my) {
        stage(prev, (it - 1) & 1);
      }
      __syncthreads();
    };
    __syncthreads();                                // prologue: tile 0 staged by loader 0
    for (int it = 0; it <= n_my + 1; it += 2) {
      iteration(it, accA, accB);
      if (it + 1 <= n_my + 1) iteration(it + 1, accB, accA);
    }
    if (STATS) __syncthreads();
  } else if (wave < 6) {
    // ---------------------------------------------------------------- loader waves: lane = row of my tiles
    // Branch-free: ONE buffer descriptor over the whole input, 40 8-byte buffer loads per row at immediate offsets.  A row that hangs
    // over its clip's start (the first row of every clip: 10 samples) reads the previous clip's tail - or, before the first clip and
    // for the rows past M, nothing: the descriptor's range check returns zeros - and the samples outside the clip are zeroed in LDS
    // afterwards by the one lane that owns the row.  (The element-by-element patch of conv1_fwd_kernel is 80 loads behind branches:
    // a loader wave that runs it holds the whole workgroup at the barrier.)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int lw = wave - 4;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (unsigned)((int64_t)p.B * p.g.x_batch_stride * 4), 0x00020000);
    float2 ra[NL];
    int e0_cur = 0;                                 // first sample of my row inside its clip (of the tile held in ra)
    bool ok_cur = false;
    auto issue = [&](int loc) {                     // local tile loc (>= n_my: nothing to load)
      if (loc >= n_my) return;
      const int tile = (int)blockIdx.x + loc * (int)gridDim.x;
      const unsigned m = (unsigned)tile * FM + lane;
      ok_cur = (int64_t)m < p.M;
      const unsigned b = m / (unsigned)p.g.L_out;
      const int t = (int)(m - b * (unsigned)p.g.L_out);
      e0_cur = t * p.g.stride_t + p.g.base_off;
      // byte offset of the row's first sample; a row past M points past the descriptor's range.  ONE row of the whole batch starts
      // in front of the buffer (the first row of the first clip): a negative offset does not wrap inside the address unit (offset +
      // immediate is range-checked unwrapped: the row would read as all zeros), so that lane alone loads element by element
      const bool front = ok_cur && b == 0 && e0_cur < 0;
      const unsigned voff = ok_cur && !front ? (b * (unsigned)p.g.x_batch_stride + (unsigned)e0_cur) * 4u : 0xFFFFF000u;
      if (KWS_C1_ABL & 8) {
#pragma unroll
        for (int i = 0; i < NL; ++i) ra[i] = make_float2((float)voff, (float)i);
        return;
      }
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xres, voff + 8 * i, 0, 0);
        ra[i] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
      }
      if (front) {
#pragma unroll
        for (int i = 0; i < NL; ++i) ra[i] = load2_or_zero(p.x, e0_cur + 2 * i, p.g.x_len);
      }
    };
    auto write_lds = [&](int loc) {
      if (loc >= n_my) return;
      float* dst = &sA[loc & 1][lane * PA];
#pragma unroll
      for (int i = 0; i < NL; ++i) *reinterpret_cast<float2*>(dst + 2 * i) = ra[i];
      if (ok_cur && (e0_cur < 0 || e0_cur + KF > p.g.x_len)) {     // rare: the samples of this row that lie outside its clip are zeros
        for (int k = 0; k < KF; ++k)
          if (e0_cur + k < 0 || e0_cur + k >= p.g.x_len) dst[k] = 0.f;
      }
    };
    int mine = lw;                                  // my next tile
    issue(mine);
    if (lw == 0) {
      write_lds(0);
      mine += 2;
      issue(mine);
    }
    __syncthreads();
    for (int it = 0; it <= n_my + 1; ++it) {
      if (mine == it + 1) {                         // slot (it + 1) & 1 was last read in iteration it - 1
        write_lds(mine);
        mine += 2;
        issue(mine);
      }
      __syncthreads();
    }
    if (STATS) __syncthreads();
  } else {
    // ---------------------------------------------------------------- storer waves
    const int st = tid - 384;                       // 0 .. 127
    const int c4 = st & 31, rg = st >> 5;           // columns 4 c4 .. 4 c4 + 3; rows rg, rg + 4, ...
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
    __syncthreads();
    for (int it = 0; it <= n_my + 1; ++it) {
      if (it >= 2) {
        const int loc = it - 2;
        const int tile = (int)blockIdx.x + loc * (int)gridDim.x;
        const float* src = &sC[loc & 1][rg * SLD + 4 * c4];
        const int64_t m0 = (int64_t)tile * FM + rg;
        float* dst = p.y + m0 * NOUT + 4 * c4;
        const bool whole = (int64_t)(tile + 1) * FM <= p.M;
#pragma unroll
        for (int i = 0; i < FM / 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(src + 4 * i * SLD);
          if (KWS_C1_ABL & 1) { asm volatile("" :: "v"(v.x)); }
          else if (whole || m0 + 4 * i < p.M) *reinterpret_cast<float4*>(dst + (int64_t)4 * i * NOUT) = v;
          if (STATS && !(KWS_C1_ABL & 2)) {         // rows past M were staged as zeros: they add nothing
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            ss.x = fmaf(v.x, v.x, ss.x); ss.y = fmaf(v.y, v.y, ss.y); ss.z = fmaf(v.z, v.z, ss.z); ss.w = fmaf(v.w, v.w, ss.w);
          }
        }
      }
      __syncthreads();
    }
    if (STATS) {
      *reinterpret_cast<float4*>(&sRed[0][rg][4 * c4]) = s;
      *reinterpret_cast<float4*>(&sRed[1][rg][4 * c4]) = ss;
      __syncthreads();
    }
  }
  if (STATS && tid < 2 * NOUT) {                    // (every wave has passed the barrier above)
    const int q = tid >> 7, c = tid & 127;
    p.stats[((int64_t)blockIdx.x * 2 + q) * NOUT + c] = ((sRed[q][0][c] + sRed[q][1][c]) + sRed[q][2][c]) + sRed[q][3][c];
  }
}
#endif  // KWS_C1_WS

// Weight gradient.  dy goes from global memory STRAIGHT into the MFMA operand registers and only the Toeplitz rows of x pass
// through LDS (20.5 KB double buffered -> three workgroups per CU).  Round 3: v_mfma_f32_16x16x4_f32 instead of 32x32x2 - the
// 80 folded kernel rows are FIVE 16-row blocks exactly, where three 32-row blocks multiplied 16 rows of LDS zeros (one MFMA
// cycle in six); a wave owns 32 output columns = two column blocks x five row blocks = ten accumulators of 16 x 16.
//   A (x rows)   lane (i = lane % 16, k = lane / 16): sA[row 4 s + k][16 rb + i] of K step s - row pitch 80 floats = 16 banks
//                mod 32, so the four rows of a read fall on disjoint bank halves;
//   B (dy)       lane (k, j = lane % 16): dy[m0 + 4 s + k][32 w + 2 j + cb] - ONE 8-byte load per K step feeds both column
//                blocks (block cb = the columns of parity cb: 16 lanes read 128 contiguous bytes of a row);
//   D            register v of block (rb, cb): dW[16 rb + 4 (lane / 16) + v][32 w + 2 j + cb] - the two blocks leave as 8-byte stores.
template <int KF>
__device__ __forceinline__ void conv1_wgrad_body(const Conv1Args& p, float (*sA)[UM * KF], const int bid) {
  constexpr int PA = KF;            // row pitch of the staged rows: 80 = 16 mod 32 banks
  constexpr int TPR = 256 / UM;     // A-staging threads per row
  constexpr int QF = KF / TPR;      // floats per A-staging thread
  constexpr int NS = UM / 4;        // K steps (4 rows each) per unit
  constexpr int RB = KF / 16;       // row blocks of dW
  static_assert(KF % 16 == 0 && KF % (2 * TPR) == 0, "whole 16-row blocks, 8-byte staging loads");
  constexpr int NL = QF / 2;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lk = lane >> 4;
  const int64_t m_begin = (int64_t)bid * p.chunk;
  const int64_t m_end = m_begin + p.chunk < p.M ? m_begin + p.chunk : p.M;
  const int arow = tid / TPR, aq = tid % TPR;
  float2 ra[NL];
  float2 g_cur[NS], g_nxt[NS];
  auto load_unit = [&](int64_t mb) {
    const int64_t m = mb + arow;
    const bool row_ok = m < m_end;
    const int64_t b = row_ok ? m / p.g.L_out : 0;
    const int t = row_ok ? (int)(m - b * p.g.L_out) : 0;
    const float* xb = p.x + b * p.g.x_batch_stride;
    const int e0 = t * p.g.stride_t + p.g.base_off + aq * QF;
    const bool inside = row_ok && e0 >= 0 && e0 + QF <= p.g.x_len;
    const float* src = inside ? xb + e0 : g_zero64;
    if (KWS_C1W_ABL & 2) {
#pragma unroll
      for (int i = 0; i < NL; ++i) ra[i] = make_float2((float)mb, (float)i);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) g_nxt[s2] = make_float2((float)(mb + s2), 1.f);
      return;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) ra[i] = *reinterpret_cast<const float2*>(src + 2 * i);
    if (row_ok && !inside) {
#pragma unroll
      for (int i = 0; i < NL; ++i) ra[i] = load2_or_zero(xb, e0 + 2 * i, p.g.x_len);
    }
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int64_t gm = mb + 4 * s2 + lk;
      const float* gsrc = gm < m_end ? p.G + gm * NOUT + wave * 32 + 2 * l16 : g_zero64;   // address select, not a branch
      g_nxt[s2] = *reinterpret_cast<const float2*>(gsrc);
    }
  };
  auto store_unit = [&](int buf) {
    float* dst = &sA[buf][arow * PA + aq * QF];
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<float2*>(dst + 2 * i) = ra[i];
  };
  auto take_g = [&]() {
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) g_cur[s2] = g_nxt[s2];
  };
  f32x4 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (m_begin < m_end) {
    load_unit(m_begin);
    store_unit(0);
  }
  take_g();
  __syncthreads();
  int buf = 0;
  for (int64_t mb = m_begin; mb < m_end; mb += UM) {
    const bool has_next = mb + UM < m_end;
    if (has_next) load_unit(mb + UM);
    asm volatile("" ::: "memory");   // keep the next unit's loads above the MFMA loop
    const float* cA = &sA[buf][lk * PA + l16];
    float a_cur[RB], a_nxt[RB];       // software-pipelined LDS reads: one K step ahead
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) a_cur[rb] = cA[16 * rb];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      if (s2 + 1 < NS) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a_nxt[rb] = cA[(4 * (s2 + 1)) * PA + 16 * rb];
      }
      __builtin_amdgcn_sched_barrier(0);   // the reads above stay above the MFMAs below
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (KWS_C1W_ABL & 1) { asm volatile("" :: "v"(a_cur[rb]), "v"(g_cur[s2].x), "v"(g_cur[s2].y)); continue; }
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[rb], g_cur[s2].x, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[rb], g_cur[s2].y, acc[rb][1], 0, 0, 0);
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a_cur[rb] = a_nxt[rb];
    }
    if (has_next) {
      store_unit(buf ^ 1);
      take_g();
    }
    __syncthreads();
    buf ^= 1;
  }
  float* slab = p.ws + (int64_t)bid * KF * NOUT;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = 16 * rb + 4 * lk + v;
      *reinterpret_cast<float2*>(slab + k * NOUT + wave * 32 + 2 * l16) = make_float2(acc[rb][0][v], acc[rb][1][v]);
    }
}

template <int KF>
__global__ __launch_bounds__(256, KWS_C1W_OCC) void conv1_wgrad_kernel(Conv1Args p) {
  __shared__ float sA[2][UM * KF];
  conv1_wgrad_body<KF>(p, sA, blockIdx.x);
}

// Round 4: the first convolution's weight gradient AND the batched slab sum of the pointwise weight gradients (independent of each
// other: one MFMA / memory bound, one HBM bound) as ONE grid - blocks [0, n_conv) are the weight-gradient slabs, the blocks behind
// them the slab-sum columns (internal.h kws_reduce_slabs_batch_body); no event between them, the hardware co-schedules the two.
template <int KF>
__global__ __launch_bounds__(256, KWS_C1W_OCC) void conv1_wgrad_slabsum_kernel(Conv1Args p, SlabBatch b, int n_conv) {
  __shared__ float sA[2][UM * KF];
  __shared__ float4 red[4][64];
  if ((int)blockIdx.x < n_conv) conv1_wgrad_body<KF>(p, sA, blockIdx.x);
  else kws_reduce_slabs_batch_body(b, (int)blockIdx.x - n_conv, red);
}

// dW[j][c][n] = sum over slab groups of ws[group][hop j + c][n]: the second stage of the slab sum writes the three taps
// directly (every tap row reads the gradient of the sample it multiplies)
__global__ __launch_bounds__(256) void conv1_unfold_sum_kernel(const float* __restrict__ ws, float* __restrict__ dW,
                                                               int groups, int64_t group_stride, int taps, int cin, int hop) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // float4 index into dW[taps][cin][128]
  if (i >= taps * cin * (NOUT / 4)) return;
  const int n4 = i % (NOUT / 4), jc = i / (NOUT / 4);
  const int j = jc / cin, c = jc - j * cin;
  const float4* src = reinterpret_cast<const float4*>(ws) + (int64_t)(hop * j + c) * (NOUT / 4) + n4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = 0; g < groups; ++g) {
    const float4 v = src[(int64_t)g * group_stride];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(dW)[i] = s;
}

struct WgradPlan {
  int S;
  int64_t chunk;
};
WgradPlan wgrad_plan(int64_t M) {
  WgradPlan pl;
  int64_t chunk = ceil_div64(ceil_div64(M, 256 * KWS_C1W_OCC), UM) * UM;   // KWS_C1W_OCC workgroups per CU, one round
  if (chunk < UM) chunk = UM;
  pl.chunk = chunk;
  pl.S = (int)ceil_div64(M, chunk);
  return pl;
}

}  // namespace

// statistics rows the forward kernel writes = its grid: three persistent 43 KB workgroups per CU (-DKWS_C1_WS=1, round 6's
// wave-role experiment: ONE 8-wave workgroup per CU)
int kws_conv1_stats_rows(int64_t M) {
  const int64_t tiles = ceil_div64(M, FM);
  const int64_t cap = KWS_C1_WS ? 256 : 768;
  return (int)(tiles < cap ? tiles : cap);
}

bool kws_conv1_supported(const kws_gather_t* g, const kws_gather_t* unfolded, int N) {
  // anything else (filter_mult = 2: 256 output channels; more than three taps folded into the 80 samples: the forward
  // kernel's fold-on-load prologue has three tap candidates) takes the generic gathered GEMMs of gemm.hip
  return g && unfolded && g->taps == 1 && g->cin == 80 && N == NOUT && g->stride_t % 2 == 0 && g->base_off % 2 == 0 &&
         g->x_batch_stride % 2 == 0 && g->L_out > 0 && unfolded->taps >= 1 && unfolded->taps <= 3;
}

int kws_conv1_fwd(const float* x, const kws_gather_t* g, const kws_gather_t* unfolded, const float* W, float* y, int B, int N,
                  float* stats, hipStream_t st) {
  KWS_REQUIRE(x && g && unfolded && W && y && B > 0 && kws_conv1_supported(g, unfolded, N), "conv1_fwd: unsupported shape");
  Conv1Args a{};
  a.x = x; a.W = W; a.y = y; a.stats = stats; a.g = *g; a.B = B; a.M = (int64_t)B * g->L_out;
  a.taps = unfolded->taps; a.cin = unfolded->cin; a.hop = unfolded->stride_j;
  a.m_tiles = (int)ceil_div64(a.M, FM);
  KwsProfScope prof("conv1_fwd", 2.0 * a.M * 80 * N, 4.0 * ((double)B * g->x_len + (double)a.M * N + 80.0 * N), st);
  const int grid = kws_conv1_stats_rows(a.M);
#if KWS_C1_WS
  // the wave-role kernel addresses the whole input through ONE buffer descriptor with 32-bit byte offsets; a batch beyond that
  // (> 4 GB of clips) takes the four-wave kernel on the same grid (persistent: any grid; one statistics row per workgroup)
  if ((int64_t)B * g->x_batch_stride * 4 < 0xFFFFF000ll && a.M < (1ll << 31)) {
    if (stats) hipLaunchKernelGGL((conv1_fwd_ws_kernel<80, true>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv1_fwd_ws_kernel<80, false>), dim3(grid), dim3(512), 0, st, a);
  } else if (stats) hipLaunchKernelGGL((conv1_fwd_kernel<80, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv1_fwd_kernel<80, false>), dim3(grid), dim3(256), 0, st, a);
#else
  if (stats) hipLaunchKernelGGL((conv1_fwd_kernel<80, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv1_fwd_kernel<80, false>), dim3(grid), dim3(256), 0, st, a);
#endif
  KWS_LAUNCH_CHECK("conv1_fwd_kernel");
  return KWS_OK;
}

int64_t kws_conv1_wgrad_workspace_floats(int64_t M) { return (int64_t)wgrad_plan(M).S * 80 * NOUT; }

int kws_conv1_wgrad(const float* x, const kws_gather_t* g, const kws_gather_t* unfolded, const float* G, float* dW, int B,
                    int N, float* workspace, hipStream_t st) {
  return kws_conv1_wgrad_slabs(x, g, unfolded, G, dW, B, N, workspace, nullptr, nullptr, nullptr, nullptr, 0, st);
}

// as kws_conv1_wgrad; with n_sl > 0 the same launch also sums the slabs of n_sl pointwise weight-gradient GEMMs (the arguments of
// kws_reduce_slabs_batch) - conv1_wgrad_slabsum_kernel
int kws_conv1_wgrad_slabs(const float* x, const kws_gather_t* g, const kws_gather_t* unfolded, const float* G, float* dW, int B,
                          int N, float* workspace, const float* const* sl_ws, float* const* sl_out, const int64_t* sl_n,
                          const int* sl_S, int n_sl, hipStream_t st) {
  KWS_REQUIRE(x && g && unfolded && G && dW && workspace && B > 0 && kws_conv1_supported(g, unfolded, N),
              "conv1_wgrad: unsupported shape");
  Conv1Args a{};
  a.x = x; a.G = G; a.ws = workspace; a.g = *g; a.B = B; a.M = (int64_t)B * g->L_out;
  const WgradPlan pl = wgrad_plan(a.M);
  a.S = pl.S; a.chunk = pl.chunk;
  if (n_sl > 0) {
    SlabBatch sb;
    int blocks = 0;
    double bytes = 0;
    KWS_TRY(kws_slab_batch_fill(&sb, sl_ws, sl_out, sl_n, sl_S, n_sl, &blocks, &bytes));
    KwsProfScope prof("conv1_wgrad", 2.0 * a.M * 80 * N, 4.0 * ((double)B * g->x_len + (double)a.M * N + 80.0 * N) + bytes, st);
    hipLaunchKernelGGL((conv1_wgrad_slabsum_kernel<80>), dim3(pl.S + blocks), dim3(256), 0, st, a, sb, pl.S);
    KWS_LAUNCH_CHECK("conv1_wgrad_slabsum_kernel");
  } else {
  KwsProfScope prof("conv1_wgrad", 2.0 * a.M * 80 * N, 4.0 * ((double)B * g->x_len + (double)a.M * N + 80.0 * N), st);
  hipLaunchKernelGGL((conv1_wgrad_kernel<80>), dim3(pl.S), dim3(256), 0, st, a);
  KWS_LAUNCH_CHECK("conv1_wgrad_kernel");
  }
  // slab sum in two stages: groups of 32 slabs in place (over each group's first slab), then the group sums straight into
  // the three taps of dW
  const int per_group = 32, groups = ceil_div(pl.S, per_group);
  KWS_TRY(kws_reduce_slab_groups_f32(workspace, (int64_t)80 * N, pl.S, per_group, st));
  const int n4 = unfolded->taps * unfolded->cin * (NOUT / 4);
  hipLaunchKernelGGL(conv1_unfold_sum_kernel, dim3(ceil_div(n4, 256)), dim3(256), 0, st, workspace, dW, groups,
                     (int64_t)per_group * 80 * (NOUT / 4), unfolded->taps, unfolded->cin, unfolded->stride_j);
  KWS_LAUNCH_CHECK("conv1_unfold_sum_kernel");
  return KWS_OK;
}
