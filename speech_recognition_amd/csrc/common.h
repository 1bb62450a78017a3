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

// ---- optional per-kernel-family profiler (off by default; bench.py turns it on for a few steps) ----
// When enabled, every launcher brackets its launch with a hipEvent pair on the launch stream and books
// the algorithmic FLOPs / bytes it was asked to process; kws_profile_get() then reports, per family,
// the summed device time between the events.  This is the "HIP events on the stream the kernel is
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
