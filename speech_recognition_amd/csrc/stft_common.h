// Device helpers of the register-FFT feature kernel (stft4.hip): complex arithmetic, the in-register 16-point FFT, launch arguments.
#pragma once
#include "internal.h"

namespace kws_fft {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// forward 4-point DFT (W4 = -i)
__device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
  const float2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = mul_mi(csub(x1, x3));
  x0 = cadd(s02, s13);
  x1 = cadd(d02, d13);
  x2 = csub(s02, s13);
  x3 = csub(d02, d13);
}

// in-place forward 16-point FFT, natural order in and out (radix-4 x radix-4, all indices static)
__device__ __forceinline__ void fft16(float2 (&a)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  // stage 1: DFT4 over s for each r (elements r, r+4, r+8, r+12) -> b[r][q] stored at a[r + 4q]
#pragma unroll
  for (int r = 0; r < 4; ++r) dft4(a[r], a[r + 4], a[r + 8], a[r + 12]);
  // twiddle W16^(r q)
  a[1 + 4] = cmul(a[1 + 4], make_float2(C1, -S1));     // r=1,q=1: W^1
  a[1 + 8] = cmul(a[1 + 8], make_float2(R2, -R2));     // r=1,q=2: W^2
  a[1 + 12] = cmul(a[1 + 12], make_float2(S1, -C1));   // r=1,q=3: W^3
  a[2 + 4] = cmul(a[2 + 4], make_float2(R2, -R2));     // r=2,q=1: W^2
  a[2 + 8] = mul_mi(a[2 + 8]);                         // r=2,q=2: W^4 = -i
  a[2 + 12] = cmul(a[2 + 12], make_float2(-R2, -R2));  // r=2,q=3: W^6
  a[3 + 4] = cmul(a[3 + 4], make_float2(S1, -C1));     // r=3,q=1: W^3
  a[3 + 8] = cmul(a[3 + 8], make_float2(-R2, -R2));    // r=3,q=2: W^6
  a[3 + 12] = cmul(a[3 + 12], make_float2(-C1, S1));   // r=3,q=3: W^9
  // stage 2: for each q, DFT4 over r of c[r][q] (at a[r + 4q]) -> A[q + 4p] ; write back in natural order
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
  // now a[4q + p] holds A[q + 4p]: transpose the 4x4 index grid to natural order
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int p = q + 1; p < 4; ++p) {
      const float2 t = a[4 * q + p];
      a[4 * q + p] = a[4 * p + q];
      a[4 * p + q] = t;
    }
}

struct Stft2Args {
  kws_stft_plan pl;
  const float* x;
  float* out;
  int B, L, F;
  int quads_per_clip;
  int64_t total_quads;
};


}  // namespace kws_fft
