// Fused multi-tensor optimizers over ONE flat parameter buffer.
// Keras 2.1.2 semantics, constants pinned by the reference graph_def:
//   RMSprop: a' = rho*a + (1-rho)*g^2 ; p' = p - lr*g / (sqrt(a') + eps)      (model.py:834)
//   SGD    : v' = m*v - lr*g         ; p' = p + v'                            (model.py:96,110)
// g = grad*grad_scale + 2*l2[i]*p folds the kernel_regularizer=l2(1e-5) gradient (model.py:37,807)
// and the data-parallel 1/world averaging into the same pass: 6 streams of n floats, HBM-bound.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                      float* __restrict__ acc, const float* __restrict__ l2, int64_t n,
                                                      float lr, float rho, float eps, float gs) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = i4 * 4;
  if (i + 3 < n) {
    float4 pv = reinterpret_cast<float4*>(p)[i4];
    const float4 gv = reinterpret_cast<const float4*>(grad)[i4];
    float4 av = reinterpret_cast<float4*>(acc)[i4];
    const float4 lv = reinterpret_cast<const float4*>(l2)[i4];
    float g;
#define KWS_RMS(c)                                   \
  g = fmaf(2.0f * lv.c, pv.c, gv.c * gs);            \
  av.c = rho * av.c + (1.0f - rho) * g * g;          \
  pv.c = pv.c - lr * g / (sqrtf(fmaxf(av.c, 0.0f)) + eps);
    KWS_RMS(x) KWS_RMS(y) KWS_RMS(z) KWS_RMS(w)
#undef KWS_RMS
    reinterpret_cast<float4*>(p)[i4] = pv;
    reinterpret_cast<float4*>(acc)[i4] = av;
  } else {
    for (int64_t j = i; j < n; ++j) {
      const float g = fmaf(2.0f * l2[j], p[j], grad[j] * gs);
      const float a = rho * acc[j] + (1.0f - rho) * g * g;
      acc[j] = a;
      p[j] = p[j] - lr * g / (sqrtf(fmaxf(a, 0.0f)) + eps);
    }
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                  float* __restrict__ vel, const float* __restrict__ l2, int64_t n,
                                                  float lr, float mom, float gs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float g = fmaf(2.0f * l2[i], p[i], grad[i] * gs);
  const float v = mom * vel[i] - lr * g;
  vel[i] = v;
  p[i] = p[i] + v;
}

__global__ __launch_bounds__(1024) void l2_loss_kernel(const float* __restrict__ p, const float* __restrict__ l2,
                                                       int64_t n, float* out) {
  // single workgroup (no scratch buffer in the call, re-entrant), fixed order: the buffer is ~1.2 M floats (2 x 4.8 MB), a metric only.
  // Round 6: 16 waves, 16-byte loads, four independent double accumulators per thread - the 256-thread scalar loop of rounds 1 - 5
  // took ~1.5 ms per call, and fit_generator reads this value every 16 steps (bench.py ab_fit_generator: 0.1 ms per step)
  __shared__ double red[1024];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(l2)) & 15) == 0;
  const int64_t n4 = aligned ? n / 4 : 0;            // unaligned views take the scalar loop below
  const float4* p4 = reinterpret_cast<const float4*>(p);
  const float4* l4 = reinterpret_cast<const float4*>(l2);
  for (int64_t i = threadIdx.x; i < n4; i += 1024) {
    const float4 a = p4[i], c = l4[i];
    s0 += (double)c.x * (double)a.x * (double)a.x;
    s1 += (double)c.y * (double)a.y * (double)a.y;
    s2 += (double)c.z * (double)a.z * (double)a.z;
    s3 += (double)c.w * (double)a.w * (double)a.w;
  }
  for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) s0 += (double)l2[i] * (double)p[i] * (double)p[i];
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)red[0];
}

}  // namespace

extern "C" {

int kws_rmsprop_step(float* p, const float* grad, float* acc, const float* l2, int64_t n, float lr, float rho,
                     float eps, float grad_scale, void* stream) {
  KWS_REQUIRE(p && grad && acc && l2 && n > 0, "rmsprop: bad arguments");
  const int64_t n4 = ceil_div64(n, 4);
  KwsProfScope prof("optimizer", 8.0 * n, 24.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)ceil_div64(n4, 256)), dim3(256), 0, (hipStream_t)stream, p, grad,
                     acc, l2, n, lr, rho, eps, grad_scale);
  KWS_LAUNCH_CHECK("rmsprop_kernel");
  return KWS_OK;
}

int kws_sgd_momentum_step(float* p, const float* grad, float* vel, const float* l2, int64_t n, float lr,
                          float momentum, float grad_scale, void* stream) {
  KWS_REQUIRE(p && grad && vel && l2 && n > 0, "sgd: bad arguments");
  KwsProfScope prof("optimizer", 6.0 * n, 24.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, p, grad, vel,
                     l2, n, lr, momentum, grad_scale);
  KWS_LAUNCH_CHECK("sgd_kernel");
  return KWS_OK;
}

int kws_l2_loss(const float* p, const float* l2, int64_t n, float* out, void* stream) {
  KWS_REQUIRE(p && l2 && out && n > 0, "l2_loss: bad arguments");
  KwsProfScope prof("l2_loss", 3.0 * n, 8.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(l2_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, l2, n, out);
  KWS_LAUNCH_CHECK("l2_loss_kernel");
  return KWS_OK;
}

}  // extern "C"
