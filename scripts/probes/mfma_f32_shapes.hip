// Probe (round 4): which f32 MFMA shape sustains more FLOP/s when the chip holds its clock down under load?
// MI355X_MICROARCH.md ('DVFS give-back' item 7) reports 1.12 - 1.15 x for the 16x16 bf16 shape over the 32x32 one in that
// regime; the f32 GEMMs of this library all use v_mfma_f32_32x32x2_f32.  Bare loops, one wave per SIMD, a 64 x 64 output
// tile per wave in both forms, random operands, operand fragments either held in registers (mode 0) or re-read from LDS every
// k group the way gemm_nn_ws_kernel reads them (mode 1: ds_read_b128 for the row operand, ds_read_b32 for the column operand).
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/probes/mfma_f32_shapes.hip -o variants/mfma_f32_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDA = 36;   // 32 k + 4 pad: conflict-free b128 reads by (row, k half)

// 32x32x2: wave tile 64 x 64 = 2 x 2 blocks; per 8-deep k group 16 MFMAs; operands: 2 b128 (rows) + 8 b32 (columns)
template <int MODE>
__global__ __launch_bounds__(256) void k32(const float* src, float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float sA[128 * LDA], sB[32 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 128 * LDA; i += 256) sA[i] = src[i];
  for (int i = tid; i < 32 * 128; i += 256) sB[i] = src[8192 + i];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  const float* cA = sA + (wm * 32 + li) * LDA + lh * 4;
  const float* cB = sB + (lh * 4) * 128 + wn * 64 + li;
  float4 a[2];
  float b[4][2];
  auto load = [&](int q) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(cA + i * 64 * LDA + q * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j) b[r][j] = cB[(q * 8 + r) * 128 + j * 32];
  };
  load(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 1) {
        __builtin_amdgcn_sched_barrier(0);
        load(q);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float av = r == 0 ? a[i].x : (r == 1 ? a[i].y : (r == 2 ? a[i].z : a[i].w));
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[r][j], av, acc[i][j], 0, 0, 0);
        }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) s += acc[i][j][v];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

// 16x16x4: wave tile 64 x 64 = 4 x 4 blocks; lane l: row / column l % 16, k group l / 16; per 16-deep k group (4 k steps of
// 4) 64 MFMAs; operands: 4 b128 (rows: k = 16 c + 4 g + r) + 16 b32 (columns)
template <int MODE>
__global__ __launch_bounds__(256) void k16(const float* src, float* out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float sA[128 * LDA], sB[32 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 128 * LDA; i += 256) sA[i] = src[i];
  for (int i = tid; i < 32 * 128; i += 256) sB[i] = src[8192 + i];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;
  const float* cA = sA + (wm * 64 + l16) * LDA + g * 4;
  const float* cB = sB + (g * 4) * 128 + wn * 64 + l16;
  float4 a[4];
  float b[4][4];
  auto load = [&](int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4*>(cA + i * 16 * LDA + c * 16);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[r][j] = cB[(c * 16 + r) * 128 + j * 16];
  };
  load(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (MODE == 1) {
        __builtin_amdgcn_sched_barrier(0);
        load(c);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float av = r == 0 ? a[i].x : (r == 1 ? a[i].y : (r == 2 ? a[i].z : a[i].w));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[r][j], av, acc[i][j], 0, 0, 0);
        }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) s += acc[i][j][v];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

int main() {
  const int grid = 256, iters = 40000;   // 32 k per iteration: 64 x 64 x 32 x 2 = 262144 FLOP per wave and iteration
  float *src, *out;
  unsigned long long* stamps;
  CK(hipMalloc(&src, 16384 * 4)); CK(hipMalloc(&out, grid * 256 * 4)); CK(hipMalloc(&stamps, grid * 4 * 2 * 8));
  std::vector<float> h(16384);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"32x32x2 regs", "16x16x4 regs", "32x32x2 lds ", "16x16x4 lds "};
  for (int round = 0; round < 3; ++round)
    for (int k = 0; k < 4; ++k) {
      auto go = [&] {
        if (k == 0) hipLaunchKernelGGL(k32<0>, dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
        if (k == 1) hipLaunchKernelGGL(k16<0>, dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
        if (k == 2) hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
        if (k == 3) hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(256), 0, 0, src, out, stamps, iters);
      };
      go();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) go();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> st(grid * 4 * 2);
      CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> ghz, cyc;
      for (int i = 0; i < grid * 4; ++i) { ghz.push_back((double)st[2 * i] / st[2 * i + 1] * 0.1); cyc.push_back((double)st[2 * i] / iters); }
      std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
      const double fl = 5.0 * grid * 4 * (double)iters * 262144.0;
      printf("round %d %s: %7.2f ms per launch, %6.1f TFLOP/s, %.0f cycles per 32-deep iteration (4096 of matrix time), %.2f GHz\n", round,
             names[k], ms / 5, fl / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2], ghz[ghz.size() / 2]);
    }
  return 0;
}
