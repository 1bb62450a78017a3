// Stand-alone entry points for the small ops of the classifier tail, for callers that bind ONE reference
// call site instead of a whole network program (SURVEY 8b minimum set):
//   kws_dropout_{fwd,bwd}               keras Dropout (model.py:819,828), the counter-based masks of common.h
//   kws_attn_pool_{fwd,bwd}             Multiply -> GlobalMaxPool1D ++ GlobalAveragePooling1D (model.py:824-827)
//   kws_softmax_xent_smooth_{fwd,bwd}   smooth_categorical_crossentropy (utils.py:87-108, model.py:835-836)
//   kws_comm_* / kws_allreduce_grads    the gradient exchange of the data-parallel step as a direct RCCL call
// The network programs keep their fused tail (tail.hip: one workgroup per clip, everything in LDS); these kernels
// are the same arithmetic in the same order, one op per launch, all memory-bound and tiny.
#include <dlfcn.h>
#include <string.h>

#include "internal.h"

namespace {

// ---- dropout ------------------------------------------------------------------------------------------------
// element (row r, column i) of a [B, n] tensor keeps its value iff fmix32((row_offset + r) * n + i ...) < thresh
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, int n,
                                                      int64_t total, uint32_t key, uint32_t thresh, float inv_keep,
                                                      int64_t row_offset) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const uint32_t idx = (uint32_t)(row_offset * n + e);   // 32-bit counter, wraps like the oracle's
    out[e] = kws_keep(idx, key, thresh) ? x[e] * inv_keep : 0.f;
  }
}

// ---- attention pooling ------------------------------------------------------------------------------------
// x [B, T, C], att [B, T] -> feat [B, 2C] = [max_t(x * att) ; mean_t(x)]
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ att,
                                                            float* __restrict__ feat, int T, int C) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* xb = x + (int64_t)b * T * C;
  const float* ab = att + (int64_t)b * T;
  float mx = xb[c] * ab[0], sm = xb[c];
  for (int t = 1; t < T; ++t) {
    const float v = xb[(int64_t)t * C + c];
    mx = fmaxf(mx, v * ab[t]);
    sm += v;
  }
  feat[(int64_t)b * 2 * C + c] = mx;
  feat[(int64_t)b * 2 * C + C + c] = sm / (float)T;
}

// dfeat [B, 2C] -> dx [B, T, C], datt_part [B, T, n_cblk]: reduce_max shares its gradient equally among ties
// (_MinOrMaxGrad); the sum over channels of dxa * x is left as per-workgroup partials and folded in a second launch
// in a fixed order (no float atomics anywhere in this library).
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ att,
                                                            const float* __restrict__ dfeat, float* __restrict__ dx,
                                                            float* __restrict__ datt_part, int T, int C) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const float* xb = x + (int64_t)b * T * C;
  const float* ab = att + (int64_t)b * T;
  const bool live = c < C;
  float mx = 0.f, share = 0.f, davg = 0.f;
  if (live) {
    mx = xb[c] * ab[0];
    for (int t = 1; t < T; ++t) mx = fmaxf(mx, xb[(int64_t)t * C + c] * ab[t]);
    int n = 0;
    for (int t = 0; t < T; ++t) n += (xb[(int64_t)t * C + c] * ab[t] == mx) ? 1 : 0;
    share = dfeat[(int64_t)b * 2 * C + c] / (float)n;
    davg = dfeat[(int64_t)b * 2 * C + C + c] / (float)T;
  }
  for (int t = 0; t < T; ++t) {
    float contrib = 0.f;
    if (live) {
      const float xv = xb[(int64_t)t * C + c];
      const float dxa = (xv * ab[t] == mx) ? share : 0.f;
      dx[((int64_t)b * T + t) * C + c] = dxa * ab[t] + davg;
      contrib = dxa * xv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_down(contrib, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) datt_part[((int64_t)b * T + t) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void fold_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n,
                                                         int n_parts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < n_parts; ++k) s += part[i * n_parts + k];
  out[i] = s;
}

// ---- smoothed categorical cross-entropy on probabilities ----------------------------------------------------
// utils.py:100-108: tf.nn.softmax_cross_entropy_with_logits(labels = y (1 - s) + s / NC, logits = log(clip(p, 1e-7, 1 - 1e-7)))
// one thread per clip (NC <= 64)
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                       float* __restrict__ per_loss, float* __restrict__ per_correct, int B,
                                                       int NC, float smoothing) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float eps = 1e-7f;
  const float* pb = p + (int64_t)b * NC;
  const float* yb = y + (int64_t)b * NC;
  float S = 0.f;
  for (int q = 0; q < NC; ++q) S += fminf(fmaxf(pb[q], eps), 1.f - eps);
  const float logS = logf(S);
  float loss = 0.f;
  int am_p = 0, am_y = 0;
  for (int q = 0; q < NC; ++q) {
    const float ysm = yb[q] * (1.f - smoothing) + smoothing / (float)NC;
    loss -= ysm * (logf(fminf(fmaxf(pb[q], eps), 1.f - eps)) - logS);
    if (pb[q] > pb[am_p]) am_p = q;
    if (yb[q] > yb[am_y]) am_y = q;
  }
  per_loss[b] = loss;
  if (per_correct) per_correct[b] = (am_p == am_y) ? 1.f : 0.f;
}

// dL/dp (the clip passes the gradient inside [eps, 1 - eps] only), scaled by inv_loss_batch; and, when dlogits is
// given, carried through the softmax that produced p: dlogits = p * (dp - sum_q dp_q p_q)
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                       float* __restrict__ dp_out, float* __restrict__ dlogits, int B, int NC,
                                                       float smoothing, float inv_loss_batch) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float eps = 1e-7f;
  const float* pb = p + (int64_t)b * NC;
  const float* yb = y + (int64_t)b * NC;
  float S = 0.f, ysum = 0.f;
  for (int q = 0; q < NC; ++q) {
    S += fminf(fmaxf(pb[q], eps), 1.f - eps);
    ysum += yb[q] * (1.f - smoothing) + smoothing / (float)NC;
  }
  float dot = 0.f;
  for (int q = 0; q < NC; ++q) {
    const float ysm = yb[q] * (1.f - smoothing) + smoothing / (float)NC;
    const float pc = fminf(fmaxf(pb[q], eps), 1.f - eps);
    const float inside = (pb[q] >= eps && pb[q] <= 1.f - eps) ? 1.f : 0.f;
    const float dp = (-ysm / pc + ysum / S) * inv_loss_batch * inside;
    dot += dp * pb[q];
    if (dp_out) dp_out[(int64_t)b * NC + q] = dp;
  }
  if (dlogits)
    for (int q = 0; q < NC; ++q) {      // dp recomputed: a runtime-indexed register array would live in scratch
      const float ysm = yb[q] * (1.f - smoothing) + smoothing / (float)NC;
      const float pc = fminf(fmaxf(pb[q], eps), 1.f - eps);
      const float inside = (pb[q] >= eps && pb[q] <= 1.f - eps) ? 1.f : 0.f;
      const float dp = (-ysm / pc + ysum / S) * inv_loss_batch * inside;
      dlogits[(int64_t)b * NC + q] = pb[q] * (dp - dot);
    }
}

}  // namespace

extern "C" {

int kws_dropout_fwd(const float* x, float* out, int B, int n, float keep_prob, uint64_t seed, uint32_t step,
                    uint32_t layer_id, int64_t row_offset, void* stream) {
  KWS_REQUIRE(x && out && B > 0 && n > 0 && keep_prob > 0.f && keep_prob <= 1.f && row_offset >= 0,
              "dropout: bad arguments (B=%d n=%d keep_prob=%g)", B, n, keep_prob);
  const int64_t total = (int64_t)B * n;
  const uint32_t key = kws_dropout_key(seed, step, layer_id);
  const uint32_t thresh = kws_dropout_threshold(keep_prob);
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096);
  KwsProfScope prof("dropout", (double)total, 8.0 * total, (hipStream_t)stream);
  hipLaunchKernelGGL(dropout_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, n, total, key, thresh,
                     1.0f / keep_prob, row_offset);
  KWS_LAUNCH_CHECK("dropout_kernel");
  return KWS_OK;
}

// the backward pass of dropout is the same masked scaling applied to the incoming gradient
int kws_dropout_bwd(const float* dy, float* dx, int B, int n, float keep_prob, uint64_t seed, uint32_t step,
                    uint32_t layer_id, int64_t row_offset, void* stream) {
  return kws_dropout_fwd(dy, dx, B, n, keep_prob, seed, step, layer_id, row_offset, stream);
}

int kws_attn_pool_fwd(const float* x, const float* att, float* feat, int B, int T, int C, void* stream) {
  KWS_REQUIRE(x && att && feat && B > 0 && B <= 65535 && T > 0 && C > 0, "attn_pool_fwd: bad arguments");
  KwsProfScope prof("attn_pool", 3.0 * B * T * C, 4.0 * ((double)B * T * C + 2.0 * B * C), (hipStream_t)stream);
  hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3((unsigned)ceil_div(C, 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     x, att, feat, T, C);
  KWS_LAUNCH_CHECK("attn_pool_fwd_kernel");
  return KWS_OK;
}

int64_t kws_attn_pool_bwd_workspace_floats(int B, int T, int C) {
  return (int64_t)B * T * ceil_div(C > 0 ? C : 1, 256);
}

int kws_attn_pool_bwd(const float* x, const float* att, const float* dfeat, float* dx, float* datt, float* workspace,
                      int B, int T, int C, void* stream) {
  KWS_REQUIRE(x && att && dfeat && dx && datt && workspace && B > 0 && B <= 65535 && T > 0 && C > 0,
              "attn_pool_bwd: bad arguments");
  const int nblk = ceil_div(C, 256);
  KwsProfScope prof("attn_pool", 8.0 * B * T * C, 4.0 * (3.0 * B * T * C + 2.0 * B * C), (hipStream_t)stream);
  hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, att,
                     dfeat, dx, workspace, T, C);
  KWS_LAUNCH_CHECK("attn_pool_bwd_kernel");
  const int64_t n = (int64_t)B * T;
  hipLaunchKernelGGL(fold_parts_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                     datt, n, nblk);
  KWS_LAUNCH_CHECK("fold_parts_kernel");
  return KWS_OK;
}

int kws_softmax_xent_smooth_fwd(const float* probs, const float* labels, float* per_loss, float* per_correct, int B,
                                int NC, float label_smoothing, void* stream) {
  KWS_REQUIRE(probs && labels && per_loss && B > 0 && NC > 0 && NC <= 64 && label_smoothing >= 0.f && label_smoothing < 1.f,
              "softmax_xent_smooth_fwd: bad arguments (B=%d NC=%d)", B, NC);
  hipLaunchKernelGGL(xent_fwd_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, probs, labels,
                     per_loss, per_correct, B, NC, label_smoothing);
  KWS_LAUNCH_CHECK("xent_fwd_kernel");
  return KWS_OK;
}

int kws_softmax_xent_smooth_bwd(const float* probs, const float* labels, float* dprobs, float* dlogits, int B, int NC,
                                float label_smoothing, float inv_loss_batch, void* stream) {
  KWS_REQUIRE(probs && labels && (dprobs || dlogits) && B > 0 && NC > 0 && NC <= 64 && label_smoothing >= 0.f &&
                  label_smoothing < 1.f,
              "softmax_xent_smooth_bwd: bad arguments (B=%d NC=%d)", B, NC);
  hipLaunchKernelGGL(xent_bwd_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, probs, labels,
                     dprobs, dlogits, B, NC, label_smoothing, inv_loss_batch);
  KWS_LAUNCH_CHECK("xent_bwd_kernel");
  return KWS_OK;
}

}  // extern "C"

// ---- RCCL wrapper ---------------------------------------------------------------------------------------------
// librccl is resolved at run time (dlopen) so that libkws_hip.so itself has no link-time dependency on it: a
// process that already carries RCCL (PyTorch-ROCm does) keeps ONE copy, and single-GPU users never load it.
struct kws_comm {
  void* comm;
  int rank, world;
};

namespace {

struct Uid {       // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
  char bytes[128];
};

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Uid, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.handle) return KWS_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    kws_set_error("kws_comm: librccl not found (%s)", dlerror());
    return KWS_E_INVALID;
  }
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Uid, int))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
    kws_set_error("kws_comm: librccl lacks the nccl* entry points");
    return KWS_E_INVALID;
  }
  g_rccl.handle = h;
  return KWS_OK;
}

int rccl_fail(const char* what, int rc) {
  kws_set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return KWS_E_HIP;
}

}  // namespace

extern "C" {

int kws_comm_unique_id(void* id128) {
  KWS_REQUIRE(id128, "kws_comm_unique_id: NULL buffer");
  KWS_TRY(rccl_load());
  const int rc = g_rccl.GetUniqueId(id128);
  return rc == 0 ? KWS_OK : rccl_fail("ncclGetUniqueId", rc);
}

int kws_comm_create(int rank, int world, const void* id128, kws_comm_t** comm) {
  KWS_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "kws_comm_create: bad rank %d / world %d", rank, world);
  KWS_TRY(rccl_load());
  Uid uid;
  memcpy(uid.bytes, id128, sizeof(uid.bytes));
  void* c = nullptr;
  const int rc = g_rccl.CommInitRank(&c, world, uid, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  kws_comm* k = new kws_comm{c, rank, world};
  *comm = k;
  return KWS_OK;
}

int kws_comm_destroy(kws_comm_t* comm) {
  if (!comm) return KWS_OK;
  const int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(comm->comm) : 0;
  delete comm;
  return rc == 0 ? KWS_OK : rccl_fail("ncclCommDestroy", rc);
}

// in-place sum of `n` floats over all ranks, enqueued on `stream` (ncclFloat32 = 7, ncclSum = 0)
int kws_allreduce_grads(kws_comm_t* comm, float* grads, int64_t n, void* stream) {
  KWS_REQUIRE(comm && grads && n > 0, "kws_allreduce_grads: bad arguments");
  const int rc = g_rccl.AllReduce(grads, grads, (size_t)n, 7, 0, comm->comm, (hipStream_t)stream);
  return rc == 0 ? KWS_OK : rccl_fail("ncclAllReduce", rc);
}

}  // extern "C"
