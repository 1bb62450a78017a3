// Round 5 probe: what do device-scope 64-bit integer atomics cost when 256 workgroups (one per CU, all 8 XCDs) add their partial
// sums into ONE shared row of accumulators at the end of a kernel - the alternative to "every workgroup writes its own row, a
// finalise launch folds them" (24 such launches sit on the training step's dependency chain; -DKWS_ABL_NO_FIN: 181 us per step).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/atomic_acc.hip -o /tmp/atomic_acc && /tmp/atomic_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void rows_kernel(float* rows, int n) {          // the shipped form: a row per workgroup
  for (int i = threadIdx.x; i < n; i += 512) rows[(size_t)blockIdx.x * n + i] = (float)(blockIdx.x + i);
}
__global__ __launch_bounds__(512) void atomic_kernel(unsigned long long* acc, int n, int limbs) {
  for (int i = threadIdx.x; i < n; i += 512) {
    const float v = (float)(blockIdx.x + i) * 0.37f;
    const float fl = floorf(v);
    atomicAdd(acc + i, (unsigned long long)(long long)fl);
    if (limbs == 2) atomicAdd(acc + n + i, (unsigned long long)(long long)((v - fl) * 4503599627370496.0f));
  }
}
// the same atomics after a streaming phase of realistic length (so that the workgroups do not arrive in lock step)
__global__ __launch_bounds__(512) void stream_then_atomic_kernel(const float4* src, size_t n4, unsigned long long* acc, int n, int limbs, float* sink) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) {
    const float4 v = src[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x == 123.456f) sink[0] = s.y + s.z + s.w;
  if (acc == nullptr) return;
  for (int i = threadIdx.x; i < n; i += 512) {
    const float v = s.x + (float)i;
    const float fl = floorf(v);
    atomicAdd(acc + i, (unsigned long long)(long long)fl);
    if (limbs == 2) atomicAdd(acc + n + i, (unsigned long long)(long long)((v - fl) * 4503599627370496.0f));
  }
}

int main() {
  const int G = 256;
  float* rows; unsigned long long* acc; float4* src; float* sink;
  const size_t n4 = (size_t)200e6 / 16;
  CK(hipMalloc(&rows, (size_t)G * 8192 * 4)); CK(hipMalloc(&acc, 2 * 8192 * 8)); CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&sink, 16));
  CK(hipMemset(acc, 0, 2 * 8192 * 8)); CK(hipMemset(src, 0, n4 * 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto time = [&](auto fn) { fn(); fn(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1e3f; };
  printf("256 workgroups x 512 threads; us per launch\n");
  for (int n : {256, 1024, 2560, 5120}) {   // values per workgroup: BN statistics of N = 128 / 512, depthwise fold of C = 512 / 1024... (5 C)
    const float t_rows = time([&] { hipLaunchKernelGGL(rows_kernel, dim3(G), dim3(512), 0, 0, rows, n); });
    const float t_a1 = time([&] { hipLaunchKernelGGL(atomic_kernel, dim3(G), dim3(512), 0, 0, acc, n, 1); });
    const float t_a2 = time([&] { hipLaunchKernelGGL(atomic_kernel, dim3(G), dim3(512), 0, 0, acc, n, 2); });
    const float t_s0 = time([&] { hipLaunchKernelGGL(stream_then_atomic_kernel, dim3(G), dim3(512), 0, 0, src, n4, (unsigned long long*)nullptr, n, 2, sink); });
    const float t_s1 = time([&] { hipLaunchKernelGGL(stream_then_atomic_kernel, dim3(G), dim3(512), 0, 0, src, n4, acc, n, 1, sink); });
    const float t_s2 = time([&] { hipLaunchKernelGGL(stream_then_atomic_kernel, dim3(G), dim3(512), 0, 0, src, n4, acc, n, 2, sink); });
    printf("n = %4d values per workgroup: rows %.1f | atomics alone: 1 limb %.1f, 2 limbs %.1f | 200 MB stream %.1f, + 1 limb %.1f, + 2 limbs %.1f\n", n, t_rows, t_a1, t_a2, t_s0, t_s1, t_s2);
  }
  return 0;
}
