// Round 5 probe: what does the per-K-slab s_barrier of the wave-specialised GEMM kernels cost, and would LDS flags
// (loader -> MFMA waves: "slab ready"; MFMA waves -> loader: "slot free") give it back?  The kernels' in-kernel stamps (round 4:
// profiles/r04_nn_per_layer.txt) put 120 - 500 of an iteration's 4,700 - 5,460 cycles at the barrier: whichever of the four MFMA
// waves arrives last, the other three wait.  With flags an MFMA wave waits only for its operands, never for its siblings.
//
// One 6-wave workgroup per CU: 4 MFMA waves (a 64 x 64 block each of a 128 x 128 tile, 64 v_mfma_f32_32x32x2_f32 per 32-deep
// slab, fragments read from LDS exactly as gemm_nn_ws_kernel reads them) + 2 loader waves (whole slabs from global memory by
// 16-byte loads, slab s + 2 in flight while slab s is consumed, written to one of two LDS slots).  No C stores, no epilogue:
// the loop alone.  Variant 0: __syncthreads() per slab.  Variant 1: flags.  Both produce the same accumulators (checked).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/ws_sync.hip -o scripts/probes/build/ws_sync && scripts/probes/build/ws_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KB = 32, PLDA = KB + 4;
constexpr int STAGE = BM * PLDA + KB * BN;          // floats per slot
constexpr int NQ = KB / 8;

struct Args {
  const float* A;      // [slabs][BM][KB] per workgroup stream (contiguous)
  const float* B;      // [slabs][KB][BN]
  float* out;          // [grid][256] checksum lanes
  unsigned long long* cyc;   // [grid][4]: cycles of MFMA wave 0 in the loop, at waits
  int slabs;
};

template <int FLAGS>
__global__ __launch_bounds__(384, 1) void ws_sync_kernel(Args p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  __shared__ unsigned ready[2], done[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2) { ready[tid] = 0; done[tid] = 0; }
  __syncthreads();
  const float* Ag = p.A + (size_t)blockIdx.x * p.slabs * BM * KB;
  const float* Bg = p.B + (size_t)blockIdx.x * p.slabs * KB * BN;
  if (wave < 4) {
    // ---------------- MFMA waves
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    unsigned long long t_wait = 0, t0 = __builtin_amdgcn_s_memtime();
    if (!FLAGS) __syncthreads();                      // slab 0 written
    for (int s = 0; s < p.slabs; ++s) {
      const int slot = s & 1;
      if (FLAGS) {
        const unsigned want = (unsigned)(s / 2 + 1) * 2u;   // two loader waves' halves... (each slab is written by ONE wave: want = gen + 1)
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(&ready[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(s / 2 + 1)) __builtin_amdgcn_s_sleep(1);   // (an LDS read: a generic-pointer volatile read is a FLAT load, which waits behind the wave's global loads)
        t_wait += __builtin_amdgcn_s_memtime() - w0;
        (void)want;
      }
      const float* cA = smem + slot * STAGE + (wm * 32 + li) * PLDA + lh * 4;
      const float* cB = smem + slot * STAGE + BM * PLDA + (lh * 4) * BN + wn * 64 + li;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float4 a[2];
        float b[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(cA + i * 64 * PLDA + q * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 2; ++j) b[r][j] = cB[(q * 8 + r) * BN + j * 32];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float av = r == 0 ? a[i].x : (r == 1 ? a[i].y : (r == 2 ? a[i].z : a[i].w));
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[r][j], av, acc[i][j], 0, 0, 0);
          }
      }
      if (FLAGS) {
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): this wave's fragment reads of the slot are done
        if (lane == 0) atomicAdd(&done[slot], 1u);
      } else {
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        t_wait += __builtin_amdgcn_s_memtime() - w0;
      }
    }
    float cs = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) cs += acc[i][j][v];
    p.out[(size_t)blockIdx.x * 256 + tid] = cs;
    if (lane == 0) {
      p.cyc[(size_t)blockIdx.x * 8 + wave * 2] = __builtin_amdgcn_s_memtime() - t0;
      p.cyc[(size_t)blockIdx.x * 8 + wave * 2 + 1] = t_wait;
    }
  } else {
    // ---------------- loader waves: wave lw takes slabs s = lw (mod 2), a whole slab each
    const int lw = wave - 4;
    constexpr int A_F4 = BM * KB / 4 / 64, B_F4 = KB * BN / 4 / 64;     // 16, 16
    float4 ra[A_F4], rb[B_F4];
    auto issue = [&](int s) {
      if (s >= p.slabs) return;
      const float4* a4 = reinterpret_cast<const float4*>(Ag + (size_t)s * BM * KB);
      const float4* b4 = reinterpret_cast<const float4*>(p.B + (size_t)(s & 7) * KB * BN);   // (weights: a few slabs, L2 resident, as in the real kernel)
#pragma unroll
      for (int r = 0; r < A_F4; ++r) ra[r] = a4[r * 64 + lane];
#pragma unroll
      for (int r = 0; r < B_F4; ++r) rb[r] = b4[r * 64 + lane];
    };
    auto write_lds = [&](int s) {
      float* dst = smem + (s & 1) * STAGE;
#pragma unroll
      for (int r = 0; r < A_F4; ++r) {
        const int f = r * 64 + lane, row = f / (KB / 4), c4 = f % (KB / 4);
        *reinterpret_cast<float4*>(dst + row * PLDA + c4 * 4) = ra[r];
      }
#pragma unroll
      for (int r = 0; r < B_F4; ++r) {
        const int f = r * 64 + lane;
        *reinterpret_cast<float4*>(dst + BM * PLDA + f * 4) = rb[r];
      }
    };
    int s = lw;
    issue(s);
    if (!FLAGS) {
      if (lw == 0) write_lds(0), s += 2, issue(s);
      __syncthreads();                                // slab 0 written
      for (int g = 0; g < p.slabs; ++g) {
        if (s == g + 1) {                             // my slab is the next one: its slot (g + 1) & 1 was read during iteration g - 1
          write_lds(s);
          s += 2;
          issue(s);
        }
        __syncthreads();
      }
    } else {
      for (; s < p.slabs; s += 2) {
        const int slot = s & 1, gen = s / 2;          // slot `slot` held slab s - 2: all four MFMA waves must have released it
        while (__hip_atomic_load(&done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)gen) __builtin_amdgcn_s_sleep(1);
        write_lds(s);
        issue(s + 2);
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the slab is in LDS
        if (lane == 0) atomicAdd(&ready[slot], 1u);
      }
    }
  }
}

int main() {
  const int G = 256, SLABS = 600;
  const size_t na = (size_t)G * SLABS * BM * KB, nb = (size_t)G * SLABS * KB * BN;
  std::vector<float> hA(na), hB(nb);
  unsigned x = 12345u;
  for (size_t i = 0; i < na; ++i) { x = x * 1664525u + 1013904223u; hA[i] = ((x >> 9) & 0xffff) / 65536.f - 0.5f; }
  for (size_t i = 0; i < nb; ++i) { x = x * 1664525u + 1013904223u; hB[i] = ((x >> 9) & 0xffff) / 65536.f - 0.5f; }
  Args a;
  float *dA, *dB, *o0, *o1; unsigned long long* cyc;
  CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dB, nb * 4)); CK(hipMalloc(&o0, G * 256 * 4)); CK(hipMalloc(&o1, G * 256 * 4)); CK(hipMalloc(&cyc, G * 8 * 8));
  CK(hipMemcpy(dA, hA.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), nb * 4, hipMemcpyHostToDevice));
  a.A = dA; a.B = dB; a.cyc = cyc; a.slabs = SLABS;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> hc(G * 8);
  std::vector<float> h0(G * 256), h1(G * 256);
  for (int rep = 0; rep < 3; ++rep)
    for (int flags = 0; flags < 2; ++flags) {
      a.out = flags ? o1 : o0;
      CK(hipMemset(a.out, 0, G * 256 * 4));
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) {
        if (flags) hipLaunchKernelGGL(ws_sync_kernel<1>, dim3(G), dim3(384), 0, 0, a);
        else hipLaunchKernelGGL(ws_sync_kernel<0>, dim3(G), dim3(384), 0, 0, a);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(hc.data(), cyc, G * 8 * 8, hipMemcpyDeviceToHost));
      double loop = 0, wait = 0;
      for (int g = 0; g < G; ++g) for (int w = 0; w < 4; ++w) { loop += hc[g * 8 + w * 2]; wait += hc[g * 8 + w * 2 + 1]; }
      loop /= G * 4.0 * SLABS; wait /= G * 4.0 * SLABS;
      const double fl = 2.0 * G * SLABS * BM * BN * KB * 5;
      printf("%s: %.1f us per launch, %.1f TFLOP/s; per slab and MFMA wave: %.0f cycles in the loop (4096 = the matrix time), %.0f at its wait\n",
             flags ? "LDS flags " : "s_barrier ", ms / 5 * 1e3, fl / (ms * 1e-3) / 1e12, loop, wait);
    }
  CK(hipMemcpy(h0.data(), o0, G * 256 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, G * 256 * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
  printf("accumulator checksums of the two variants: %zu of %zu lanes differ (0 = the flag protocol delivered every slab)\n", bad, h0.size());
  return 0;
}
