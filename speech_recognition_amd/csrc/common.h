// Shared helpers for libkws_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kws_hip.h"

void kws_set_error(const char* fmt, ...);

#define KWS_REQUIRE(cond, ...)                      \
  do {                                              \
    if (!(cond)) {                                  \
      kws_set_error(__VA_ARGS__);                   \
      return KWS_E_INVALID;                         \
    }                                               \
  } while (0)

#define KWS_HIP(call)                                                          \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      kws_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return KWS_E_HIP;                                                        \
    }                                                                          \
  } while (0)

#define KWS_LAUNCH_CHECK(name)                                                 \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      kws_set_error("launch of %s failed: %s", name, hipGetErrorString(e_));   \
      return KWS_E_HIP;                                                        \
    }                                                                          \
  } while (0)

#define KWS_TRY(call)            \
  do {                           \
    int rc_ = (call);            \
    if (rc_ != KWS_OK) return rc_; \
  } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- |x| maximum of a tensor, for the fp16 x 2 GEMM arm (gemm_f16x2.hip) -----------------------------------------------
// Kept as the bit pattern of a non-negative float in KWS_ABSMAX_SLOTS words 64 bytes apart (zeroed at the start of a
// step): every wave of a producing kernel commits its own maximum with one atomicMax - order-independent, so the value a
// consumer derives its power-of-two scale from is the same in every run.
constexpr int KWS_ABSMAX_SLOTS = 16, KWS_ABSMAX_STRIDE = 16, KWS_ABSMAX_WORDS = KWS_ABSMAX_SLOTS * KWS_ABSMAX_STRIDE;
#ifdef __HIPCC__
__device__ __forceinline__ float kws_abs4max(float m, const float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void kws_absmax_commit(unsigned* slots, float m) {
  const unsigned long long act = __ballot(1);      // lanes that are here (a partial last wave, early exits): only their values count
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(m, o);
    if ((act >> (lane ^ o)) & 1ull) m = fmaxf(m, other);
  }
  if ((threadIdx.x & 63) == 0)
    atomicMax(slots + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % KWS_ABSMAX_SLOTS) * KWS_ABSMAX_STRIDE, __float_as_uint(m));
}
// scale = 2^(14 - floor(log2 max)): the largest magnitude lands in [2^14, 2^15), below fp16's 65504; inv = 1 / scale.
// A zero (or denormal) tensor takes 2^125; an infinite / NaN maximum leaves the infinities in place.
__device__ __forceinline__ float kws_absmax_scale(const unsigned* slots, float& inv) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < KWS_ABSMAX_SLOTS; ++i) {
    const unsigned v = slots[i * KWS_ABSMAX_STRIDE];
    m = v > m ? v : m;
  }
  int e = (int)((m >> 23) & 0xffu);
  if (e < 16) e = 16;
  inv = __uint_as_float((unsigned)(e - 14) << 23);
  return __uint_as_float((unsigned)(268 - e) << 23);
}
#endif

// ---- counter-based dropout RNG (bit-for-bit the oracle's oracle/layers.py) ----------------
__host__ __device__ static inline uint32_t kws_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
static inline uint32_t kws_dropout_key(uint64_t seed, uint32_t step, uint32_t layer_id) {
  uint32_t k = kws_fmix32((uint32_t)(seed & 0xFFFFFFFFu) ^ 0x85EBCA6Bu);
  k = kws_fmix32(k ^ (uint32_t)(seed >> 32));
  k = kws_fmix32(k + step * 0x9E3779B1u);
  k = kws_fmix32(k ^ (layer_id * 0xC2B2AE35u));
  return k;
}
static inline uint32_t kws_dropout_threshold(double keep_prob) {
  double t = keep_prob * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
__device__ static inline bool kws_keep(uint32_t idx, uint32_t key, uint32_t thresh) {
  return kws_fmix32(idx * 0x9E3779B1u + key) < thresh;
}

__device__ static inline float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// ---- optional per-kernel-family profiler (a handle the calling thread attaches: kws_profiler_attach) ----
// While a profiler is attached to the calling thread, every launcher brackets its launch with a hipEvent pair
// on the launch stream and books the algorithmic FLOPs / bytes it was asked to process; kws_profiler_get()
// then reports, per family, the summed device time between the events.  This is the "HIP events on the stream the kernel is
// launched on" measurement of the roofline numbers.
bool kws_prof_on();
void* kws_prof_begin(hipStream_t st);
void kws_prof_end(void* token, const char* name, double flops, double bytes, hipStream_t st);
// KWS_ROCTX=1: every launcher also opens a roctx range under its family name, so the host side of each C-ABI call shows
// up in rocprofv3 --marker-trace timelines (librocprofiler-sdk-roctx / libroctx64 resolved at run time; a no-op otherwise)
bool kws_roctx_on();
void kws_roctx_push(const char* name);
void kws_roctx_pop();
struct KwsProfScope {
  void* tok;
  const char* name;
  double flops, bytes;
  hipStream_t st;
  bool tx;
  KwsProfScope(const char* n, double f, double b, hipStream_t s) : tok(nullptr), name(n), flops(f), bytes(b), st(s), tx(false) {
    if (kws_prof_on()) tok = kws_prof_begin(s);
    if (kws_roctx_on()) {
      kws_roctx_push(n);
      tx = true;
    }
  }
  ~KwsProfScope() {
    if (tok) kws_prof_end(tok, name, flops, bytes, st);
    if (tx) kws_roctx_pop();
  }
};
