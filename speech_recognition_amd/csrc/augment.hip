// Batch augmentation from an HBM-resident clip bank, the TTA transforms
// (a17) and the 32->12 class head (a18).  All HBM-bound streaming kernels: one thread per 4 output
// samples, 16-B stores; sources are read with per-lane dword loads because the circular time shift
// and the noise offset make them arbitrarily aligned (consecutive lanes still read consecutive
// addresses, so every wave instruction is a contiguous 256-B request).
#include "common.h"

namespace {

template <typename BankT>
__device__ __forceinline__ float bank_sample(const BankT* p, int i);
template <>
__device__ __forceinline__ float bank_sample<float>(const float* p, int i) { return p[i]; }
template <>
__device__ __forceinline__ float bank_sample<int16_t>(const int16_t* p, int i) {
  return (float)p[i] * (1.0f / 32768.0f);  // DecodeWav: int16 / 32768, exact in f32
}

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));     // 4-byte aligned 16-byte vector
typedef short i16x4_u __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ void load4_unaligned(const float* p, float (&f)[4]) {
  const f32x4_u v = *reinterpret_cast<const f32x4_u*>(p);
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
__device__ __forceinline__ void load4_unaligned(const int16_t* p, float (&f)[4]) {
  const i16x4_u v = *reinterpret_cast<const i16x4_u*>(p);
  f[0] = (float)v.x * (1.0f / 32768.0f); f[1] = (float)v.y * (1.0f / 32768.0f);
  f[2] = (float)v.z * (1.0f / 32768.0f); f[3] = (float)v.w * (1.0f / 32768.0f);
}

// out[b,t] = noise[off_b + t] * bgv_b  +  bank[idx_b][(t - s_b) mod L] * fg_b
// Separate (unfused) multiplies and add, in the operand order of the TF graph
// (input_data.py:340-355: multiply, roll, multiply, add(background_mul, shifted_foreground)), so the
// result is bit-identical to an f32 NumPy evaluation.
template <typename BankT>
__global__ __launch_bounds__(256) void augment_kernel(const BankT* __restrict__ bank, int64_t n_clips, int L,
                                                      const int32_t* __restrict__ clip_idx,
                                                      const float* __restrict__ fg_vol,
                                                      const int32_t* __restrict__ shift,
                                                      const float* __restrict__ noise, int64_t noise_len,
                                                      const int64_t* __restrict__ noise_off,
                                                      const float* __restrict__ bg_vol, float* __restrict__ out,
                                                      int L4) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= L4) return;
  const int t0 = q * 4;
  int64_t ci = clip_idx[b];
  if (ci < 0 || ci >= n_clips) ci = 0;  // host validates; never fault on a bad index
  const BankT* src = bank + ci * (int64_t)L;
  const float fg = fg_vol[b];
  const float bgv = bg_vol[b];
  int s = shift[b] % L;
  if (s < 0) s += L;
  int p = t0 - s;
  if (p < 0) p += L;
  float v[4];
  const bool use_noise = noise != nullptr && bgv != 0.f;
  const int64_t o0 = use_noise ? noise_off[b] + t0 : 0;
  // Fast path (all but the threads at the roll seam, the clip end or the noise ends): the four source samples
  // are contiguous, so each stream is ONE unaligned 16-byte (8-byte for int16) load instead of four dependent,
  // branch-guarded dword loads.
  if (t0 + 3 < L && p + 3 < L && (!use_noise || (o0 >= 0 && o0 + 3 < noise_len))) {
    float f[4], n[4] = {0.f, 0.f, 0.f, 0.f};
    load4_unaligned(src + p, f);
    if (use_noise) {
      const f32x4_u nv = *reinterpret_cast<const f32x4_u*>(noise + o0);
      n[0] = nv.x; n[1] = nv.y; n[2] = nv.z; n[3] = nv.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(use_noise ? __fmul_rn(n[e], bgv) : 0.f, __fmul_rn(f[e], fg));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = t0 + e;
      float r = 0.f;
      if (t < L) {
        int pe = p + e;
        if (pe >= L) pe -= L;
        const float f = __fmul_rn(bank_sample<BankT>(src, pe), fg);
        float n = 0.f;
        if (use_noise) {
          const int64_t o = o0 + e;
          n = __fmul_rn((o >= 0 && o < noise_len) ? noise[o] : 0.f, bgv);
        }
        r = __fadd_rn(n, f);
      }
      v[e] = r;
    }
  }
  float* ob = out + (int64_t)b * L + t0;
  if (t0 + 3 < L && (L & 3) == 0) {
    *reinterpret_cast<float4*>(ob) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t0 + e < L) ob[e] = v[e];
  }
}

__global__ __launch_bounds__(256) void tta_kernel(const float* __restrict__ x, float* __restrict__ out, int L, int kind) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  const float* xb = x + (int64_t)b * L;
  float v;
  if (kind == 1) {
    int p = t + 1500 % L;  // np.roll(X, -1500, axis=1): out[t] = x[(t + 1500) mod L]
    p %= L;
    v = xb[p];
  } else {
    v = xb[t];
    if (kind == 2) v = __fmul_rn(1.2f, v);
    else if (kind == 3) v = fminf(fmaxf(__fmul_rn(1.1f, v), -1.0f), 1.0f);
    else if (kind == 4) v = __fmul_rn(0.9f, v);
  }
  out[(int64_t)b * L + t] = v;
}

struct CombineArgs {
  const float* terms[8];
  int n_terms;
  float divisor;
};
__global__ __launch_bounds__(256) void tta_combine_kernel(CombineArgs a, float* __restrict__ out,
                                                          int32_t* __restrict__ amax, int B, int C) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int best = 0;
  float bv = 0.f;
  for (int c = 0; c < C; ++c) {
    float s = a.terms[0][(int64_t)b * C + c];
    for (int k = 1; k < a.n_terms; ++k) s += a.terms[k][(int64_t)b * C + c];  // (p0 + p1) + p2 ... like NumPy
    s = s / a.divisor;
    out[(int64_t)b * C + c] = s;
    if (c == 0 || s > bv) {
      bv = s;
      best = c;
    }
  }
  if (amax) amax[b] = best;
}

__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ p_in, int C_in,
                                                   const int32_t* __restrict__ map, int C_out,
                                                   float* __restrict__ p_out, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float z[64];
  for (int s = 0; s < C_out; ++s) z[s] = -INFINITY;
  for (int i = 0; i < C_in; ++i) {
    const int s = map[i];
    if (s >= 0 && s < C_out) z[s] = fmaxf(z[s], p_in[(int64_t)b * C_in + i]);
  }
  float m = z[0];
  for (int s = 1; s < C_out; ++s) m = fmaxf(m, z[s]);
  float den = 0.f;
  for (int s = 0; s < C_out; ++s) {
    z[s] = expf(z[s] - m);
    den += z[s];
  }
  for (int s = 0; s < C_out; ++s) p_out[(int64_t)b * C_out + s] = z[s] / den;
}

template <typename BankT>
int launch_augment(const BankT* bank, int64_t n_clips, int L, const int32_t* clip_idx, const float* fg_vol,
                   const int32_t* shift, const float* noise, int64_t noise_len, const int64_t* noise_off,
                   const float* bg_vol, float* out, int B, void* stream) {
  KWS_REQUIRE(bank && clip_idx && fg_vol && shift && bg_vol && out, "augment: NULL pointer");
  KWS_REQUIRE(n_clips > 0 && L > 0 && B > 0, "augment: bad sizes n_clips=%lld L=%d B=%d", (long long)n_clips, L, B);
  KWS_REQUIRE(noise == nullptr || (noise_off != nullptr && noise_len > 0), "augment: noise given without offsets");
  const int L4 = ceil_div(L, 4);
  KwsProfScope prof("augment", 3.0 * B * L, (double)B * L * (sizeof(BankT) + 8.0), (hipStream_t)stream);
  // grid.y carries the clip and is capped at 65535: a whole partition at once (get_data(how_many=-1),
  // get_unprocessed_data(-1); input_data.py:404-407,553) goes out as several launches
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    dim3 g((unsigned)ceil_div(L4, 256), (unsigned)nb), b(256);
    hipLaunchKernelGGL((augment_kernel<BankT>), g, b, 0, (hipStream_t)stream, bank, n_clips, L, clip_idx + b0, fg_vol + b0,
                       shift + b0, noise, noise_len, noise_off ? noise_off + b0 : nullptr, bg_vol + b0,
                       out + (int64_t)b0 * L, L4);
    KWS_LAUNCH_CHECK("augment_kernel");
  }
  return KWS_OK;
}

}  // namespace

extern "C" {

int kws_augment_f32(const float* bank, int64_t n_clips, int L, const int32_t* clip_idx, const float* fg_vol,
                    const int32_t* shift, const float* noise, int64_t noise_len, const int64_t* noise_off,
                    const float* bg_vol, float* out, int B, void* stream) {
  return launch_augment<float>(bank, n_clips, L, clip_idx, fg_vol, shift, noise, noise_len, noise_off, bg_vol, out, B,
                               stream);
}

int kws_augment_i16(const int16_t* bank, int64_t n_clips, int L, const int32_t* clip_idx, const float* fg_vol,
                    const int32_t* shift, const float* noise, int64_t noise_len, const int64_t* noise_off,
                    const float* bg_vol, float* out, int B, void* stream) {
  return launch_augment<int16_t>(bank, n_clips, L, clip_idx, fg_vol, shift, noise, noise_len, noise_off, bg_vol, out,
                                 B, stream);
}

int kws_tta_transform(const float* x, float* out, int B, int L, int kind, void* stream) {
  KWS_REQUIRE(x && out && B > 0 && L > 0, "tta_transform: bad arguments");
  KWS_REQUIRE(kind >= 0 && kind <= 4, "tta_transform: kind %d unknown", kind);
  for (int b0 = 0; b0 < B; b0 += 65535) {      // grid.y cap, as in launch_augment
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    dim3 g((unsigned)ceil_div(L, 256), (unsigned)nb), b(256);
    hipLaunchKernelGGL(tta_kernel, g, b, 0, (hipStream_t)stream, x + (int64_t)b0 * L, out + (int64_t)b0 * L, L, kind);
    KWS_LAUNCH_CHECK("tta_kernel");
  }
  return KWS_OK;
}

int kws_tta_combine(const float* const* probs, int n_terms, float divisor, float* out_probs, int32_t* out_argmax,
                    int B, int C, void* stream) {
  KWS_REQUIRE(probs && out_probs && n_terms >= 1 && n_terms <= 8 && B > 0 && C > 0 && divisor != 0.f,
              "tta_combine: bad arguments");
  CombineArgs a{};
  for (int k = 0; k < n_terms; ++k) {
    KWS_REQUIRE(probs[k] != nullptr, "tta_combine: term %d is NULL", k);
    a.terms[k] = probs[k];
  }
  a.n_terms = n_terms;
  a.divisor = divisor;
  hipLaunchKernelGGL(tta_combine_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     out_probs, out_argmax, B, C);
  KWS_LAUNCH_CHECK("tta_combine_kernel");
  return KWS_OK;
}

int kws_head32to12(const float* p_in, int C_in, const int32_t* map, int C_out, float* p_out, int B, void* stream) {
  KWS_REQUIRE(p_in && map && p_out && C_in > 0 && C_out > 0 && C_out <= 64 && B > 0, "head32to12: bad arguments");
  hipLaunchKernelGGL(head_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, p_in, C_in, map,
                     C_out, p_out, B);
  KWS_LAUNCH_CHECK("head_kernel");
  return KWS_OK;
}

}  // extern "C"
