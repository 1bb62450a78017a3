// Probe: issue cost of the vector instructions the STFT stage is made of, per SIMD, with 1 and 3 waves per SIMD.
// Every kernel runs ITERS trips of 32 independent instructions of one kind (8 destination registers in rotation) and
// reports s_memtime cycles per instruction per wave; with W waves per SIMD the SIMD's cost per instruction is that / W.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/valu_rates.hip -o variants/valu_rates ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PROBE(NAME, DECL, BODY)                                                                              \
  __global__ void NAME(int iters, unsigned long long* out, float* sink) {                                    \
    const int lane = threadIdx.x & 63;                                                                       \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                       \
    DECL                                                                                                     \
    __shared__ float sh[4096]; /* 16 KB: the LDS probes address real memory */                               \
    if (iters < 0) sh[threadIdx.x] = 1.f;                                                                    \
    __syncthreads();                                                                                         \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < iters; ++i) {                                                                        \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) { _Pragma("unroll") for (int u = 0; u < 8; ++u) { BODY } } \
    }                                                                                                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    float res = 0.f;                                                                                         \
    for (int u = 0; u < 8; ++u) res += x[u] + y[u].x + y[u].y + __uint_as_float(h[u]) + z4.x + sh[lane];                       \
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;                                                    \
    if (res == 123.456f) sink[0] = res;                                                                      \
  }

#define DECL0                                                                                    \
  float x[8]; f32x2 y[8]; unsigned h[8]; f32x4 z4 = {0, 0, 0, 0}; const unsigned long long smask = 0x0001000100010001ull; (void)smask; (void)z4;                                                    \
  const float m = 1.0001f, c = 0.001f; const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {0.001f, 0.002f}; \
  for (int u = 0; u < 8; ++u) { x[u] = lane * 0.01f + u + 1.f; y[u] = f32x2{x[u], x[u] + 1.f}; h[u] = lane + u; }

PROBE(p_fma, DECL0, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(m), "v"(c));)
PROBE(p_add, DECL0, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[u]) : "v"(c));)
PROBE(p_pk_fma, DECL0, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[u]) : "v"(m2), "v"(c2));)
PROBE(p_pk_add, DECL0, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[u]) : "v"(c2));)
PROBE(p_pk_mul, DECL0, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[u]) : "v"(m2));)
PROBE(p_pk_mov, DECL0, asm volatile("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(y[u]) : "v"(c2));)
PROBE(p_sqrt, DECL0, asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[u]));)
PROBE(p_log, DECL0, asm volatile("v_log_f32 %0, %0" : "+v"(x[u]));)
PROBE(p_cvt_pkrtz, DECL0, asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[u]) : "v"(x[u]), "v"(x[(u + 1) & 7]));)
PROBE(p_cvt_pk, DECL0, asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[u]) : "v"(x[u]), "v"(x[(u + 1) & 7]));)
PROBE(p_fma_mix, DECL0, asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[u]) : "v"(h[u]), "v"(m));)
PROBE(p_dpp, DECL0, asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(h[u]) : "v"(h[(u + 1) & 7]));)
PROBE(p_cndmask, DECL0, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[u]) : "v"(c));)
PROBE(p_cndmask_s, DECL0, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[u]) : "v"(c), "s"(smask));)
PROBE(p_bfi, DECL0, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(h[u]) : "v"(h[(u + 3) & 7]), "v"(lane));)
PROBE(p_max, DECL0, asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[u]) : "v"(c));)
PROBE(p_ds_b32, DECL0, asm volatile("ds_read_b32 %0, %1" : "=v"(x[u]) : "v"(lane * 4 + u * 256)); if (u == 7) asm volatile("s_waitcnt lgkmcnt(0)");)
PROBE(p_ds_b64, DECL0, asm volatile("ds_read_b64 %0, %1" : "=v"(y[u]) : "v"(lane * 8 + u * 512)); if (u == 7) asm volatile("s_waitcnt lgkmcnt(0)");)
PROBE(p_ds_b128, DECL0, asm volatile("ds_read_b128 %0, %1" : "=v"(z4) : "v"(lane * 16 + u * 1024)); if (u == 7) asm volatile("s_waitcnt lgkmcnt(0)");)
PROBE(p_ds_w32, DECL0, asm volatile("ds_write_b32 %0, %1" :: "v"(lane * 4 + u * 256), "v"(x[u])); if (u == 7) asm volatile("s_waitcnt lgkmcnt(0)");)
PROBE(p_mul_e64, DECL0, asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(x[u]) : "v"(m));)

typedef void (*kern_t)(int, unsigned long long*, float*);

int main() {
  unsigned long long* d; float* s;
  const int blocks = 256, iters = 4000;
  hipMalloc(&d, blocks * 16 * sizeof(unsigned long long)); hipMalloc(&s, 4);
  std::vector<unsigned long long> hbuf(blocks * 16);
  struct { const char* n; kern_t k; } ks[] = {
    {"v_fma_f32", p_fma}, {"v_add_f32", p_add}, {"v_mul_f32_e64", p_mul_e64}, {"v_pk_fma_f32", p_pk_fma}, {"v_pk_add_f32", p_pk_add},
    {"v_pk_mul_f32", p_pk_mul}, {"v_pk_mov_b32", p_pk_mov}, {"v_sqrt_f32", p_sqrt}, {"v_log_f32", p_log},
    {"v_cvt_pkrtz_f16_f32", p_cvt_pkrtz}, {"v_cvt_pk_f16_f32", p_cvt_pk}, {"v_fma_mix_f32", p_fma_mix},
    {"v_mov_b32_dpp row_mirror", p_dpp}, {"v_cndmask_b32 (vcc)", p_cndmask}, {"v_cndmask_b32_e64 (sgpr mask)", p_cndmask_s},
    {"v_bfi_b32", p_bfi}, {"v_max_f32", p_max}, {"ds_read_b32", p_ds_b32}, {"ds_read_b64", p_ds_b64}, {"ds_read_b128", p_ds_b128},
    {"ds_write_b32", p_ds_w32}};
  for (auto& k : ks)
    for (int wps : {1, 2, 3}) {
      const int nw = 4 * wps;
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64 * nw), 0, 0, iters, d, s);
      if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", k.n); return 1; }
      hipMemcpy(hbuf.data(), d, hbuf.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> v;
      for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) v.push_back((double)hbuf[b * 16 + w]);
      std::sort(v.begin(), v.end());
      const double per = v[v.size() / 2] / (32.0 * iters);
      printf("%-26s %d wave(s)/SIMD: %6.2f cycles per instruction and wave = %5.2f per instruction on the SIMD\n", k.n, wps, per, per / wps);
    }
  return 0;
}
