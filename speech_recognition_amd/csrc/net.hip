// Network programs: the forward / forward+backward of one model as a native sequence of kernel
// launches on one HIP stream (no Python between layers, so the step can also be captured into a
// hipGraph by the caller).  SURVEY 8a rows a7-a15.
//
//   KWS_NET_TS_ATTENTION  conv_1d_time_sliced_with_attention_model   reference model.py:775-838
//
// Data flow per depthwise block l (training):
//   z_l = dw_l( relu6(bn_{l-1}(y_{l-1})) )      BN+ReLU6 applied on load, never materialised
//   y_l = z_l W_l                                f32 MFMA GEMM, BN statistics in the epilogue
// Backward keeps two scratch tensors (G: gradient wrt a BN output / pre-BN tensor, DZ: gradient wrt
// a depthwise output) and walks the blocks in reverse.
#include "net_internal.h"
#include <atomic>
#include <math.h>
#include <stdlib.h>

int64_t kws_net_add_tensor(kws_net* n, const std::string& name, std::vector<int64_t> shape, bool is_state, float l2,
                           int fan_in, int fan_out, float init) {
  kws_tensor_info_t t;
  memset(&t, 0, sizeof(t));
  snprintf(t.name, sizeof(t.name), "%s", name.c_str());
  int64_t size = 1;
  t.ndim = (int)shape.size();
  for (int i = 0; i < t.ndim; ++i) {
    t.shape[i] = shape[i];
    size *= shape[i];
  }
  t.size = size;
  t.is_state = is_state ? 1 : 0;
  t.l2 = l2;
  t.fan_in = fan_in;
  t.fan_out = fan_out;
  t.init = init;
  int64_t& cursor = is_state ? n->n_state : n->n_params;
  t.offset = cursor;
  cursor += (size + 3) / 4 * 4;  // keep every tensor 16-B aligned inside the flat buffer
  n->tensors.push_back(t);
  return t.offset;
}

BnRef kws_net_add_bn(kws_net* n, int idx, int C) {
  BnRef r;
  const std::string base = "batch_normalization_" + std::to_string(idx) + "/";
  r.gamma = kws_net_add_tensor(n, base + "gamma", {C}, false, 0.f, 0, 0, 1.f);
  r.beta = kws_net_add_tensor(n, base + "beta", {C}, false, 0.f, 0, 0, 0.f);
  r.mm = kws_net_add_tensor(n, base + "moving_mean", {C}, true, 0.f, 0, 0, 0.f);
  r.mv = kws_net_add_tensor(n, base + "moving_variance", {C}, true, 0.f, 0, 0, 1.f);
  r.C = C;
  return r;
}

namespace {

constexpr float BN_EPS = KWS_BN_EPS;
constexpr float BN_MOMENTUM = KWS_BN_MOMENTUM;
constexpr float L2_COEF = KWS_L2_COEF;
constexpr float DROP_KEEP = 0.6f;     // Dropout(0.4), model.py:819,828
constexpr float LABEL_SMOOTH = 0.1f;  // model.py:835-836

inline int64_t add_tensor(kws_net* n, const std::string& name, std::vector<int64_t> shape, bool is_state, float l2,
                          int fan_in, int fan_out, float init) {
  return kws_net_add_tensor(n, name, shape, is_state, l2, fan_in, fan_out, init);
}
inline BnRef add_bn(kws_net* n, int idx, int C) { return kws_net_add_bn(n, idx, C); }

void same_pad(int L, int k, int s, int* Lout, int* pl) {
  *Lout = (L + s - 1) / s;
  int p = (*Lout - 1) * s + k - L;
  if (p < 0) p = 0;
  *pl = p / 2;  // TF: extra padding goes to the right
}

int build_ts_attention(kws_net* n) {
  const kws_net_config_t& c = n->cfg;
  KWS_REQUIRE(c.num_classes >= 2 && c.num_classes <= 64, "net: num_classes %d out of range", c.num_classes);
  KWS_REQUIRE(c.filter_mult >= 1 && c.filter_mult <= 2, "net: filter_mult %d unsupported", c.filter_mult);
  KWS_REQUIRE(c.input_size >= 1600 && c.input_size % 4 == 0, "net: input_size %d unsupported", c.input_size);
  const int fm = c.filter_mult;
  n->L_in = c.input_size;
  // overlapping_time_slice_stack(x, 40, 20) SAME (model.py:805) fused with Conv1D(128,3,strides=2) (model.py:807)
  int Lf, plf;
  same_pad(n->L_in, 40, 20, &Lf, &plf);
  n->L1 = (Lf - 3) / 2 + 1;
  n->C1 = 128 * fm;
  n->conv1 = add_tensor(n, "conv1d_1/kernel", {3, 40, n->C1}, false, L2_COEF, 3 * 40, 3 * n->C1, 0.f);
  n->bn1 = add_bn(n, 1, n->C1);
  kws_gather_t g;
  g.L_out = n->L1; g.cin = 40; g.taps = 3; g.stride_t = 2 * 20; g.stride_j = 20; g.base_off = -plf;
  g.x_len = n->L_in; g.x_batch_stride = n->L_in;
  n->gather1 = g;
  // The three taps overlap (tap j covers samples 20 j .. 20 j + 39 of an 80-sample span): the convolution is a GEMM
  // over the 80 DISTINCT samples with the kernel rows that hit the same sample added up front - 2/3 of the FLOPs.
  n->K1f = g.stride_j * (g.taps - 1) + g.cin;
  kws_gather_t gf = g;
  gf.cin = n->K1f; gf.taps = 1; gf.stride_j = 0;
  n->gather1f = gf;
  static const int spec[11][2] = {{1, 128}, {2, 192}, {1, 192}, {2, 256}, {1, 256}, {2, 320},
                                  {1, 320}, {2, 384}, {1, 384}, {2, 512}, {1, 512}};  // model.py:812-817
  int L = n->L1, cin = n->C1;
  for (int i = 0; i < 11; ++i) {
    Block b;
    b.stride = spec[i][0];
    b.cin = cin;
    b.cout = spec[i][1] * fm;
    b.Lin = L;
    if (b.stride == 2) {
      same_pad(L, 3, 2, &b.Lout, &b.pad_l);  // _reduce_conv: padding='same'
    } else {
      b.Lout = L - 2;  // _context_conv: padding='valid'
      b.pad_l = 0;
    }
    KWS_REQUIRE(b.Lout >= 1, "net: input too short for block %d", i);
    b.dw = add_tensor(n, "depthwise_conv2d_" + std::to_string(i + 1) + "/depthwise_kernel", {1, 3, cin, 1}, false,
                      L2_COEF, 3 * cin, 3, 0.f);
    b.pw = add_tensor(n, "conv1d_" + std::to_string(i + 2) + "/kernel", {1, cin, b.cout}, false, L2_COEF, cin,
                      b.cout, 0.f);
    b.bn = add_bn(n, i + 2, b.cout);
    n->blocks.push_back(b);
    L = b.Lout;
    cin = b.cout;
  }
  n->T = L;
  n->C = cin;
  n->NC = c.num_classes;
  KWS_REQUIRE(n->T <= 16, "net: %d time steps at the tail (max 16)", n->T);
  n->d1k = add_tensor(n, "dense_1/kernel", {(int64_t)n->T * n->C, n->T}, false, L2_COEF, n->T * n->C, n->T, 0.f);
  n->d1b = add_tensor(n, "dense_1/bias", {n->T}, false, 0.f, 0, 0, 0.f);
  n->d2k = add_tensor(n, "dense_2/kernel", {2 * n->C, n->NC}, false, L2_COEF, 2 * n->C, n->NC, 0.f);
  return KWS_OK;
}

// ---- first convolution with overlapping taps, folded -------------------------------------------------
// Weff[s, n] = sum over taps j (ascending) with 0 <= s - hop*j < cin of W[j, s - hop*j, n]
__global__ __launch_bounds__(256) void fold_taps_kernel(const float* __restrict__ W, float* __restrict__ Weff, int taps,
                                                        int cin, int hop, int Kf, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Kf * N) return;
  const int s = i / N, n = i - s * N;
  float acc = 0.f;
  for (int j = 0; j < taps; ++j) {
    const int c = s - hop * j;
    if (c >= 0 && c < cin) acc += W[((int64_t)j * cin + c) * N + n];
  }
  Weff[i] = acc;
}
// dW[j, c, n] = dWeff[hop*j + c, n]: every tap row reads the gradient of the sample it multiplies - the same sum
// over (clip, frame) as the unfolded weight-gradient GEMM, not a regrouping
__global__ __launch_bounds__(256) void unfold_taps_kernel(const float* __restrict__ dWeff, float* __restrict__ dW, int taps,
                                                          int cin, int hop, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= taps * cin * N) return;
  const int n = i % N, jc = i / N;
  const int j = jc / cin, c = jc - j * cin;
  dW[i] = dWeff[(int64_t)(hop * j + c) * N + n];
}
int fold_conv1(const kws_net* net, const float* params, float* w1f, hipStream_t st) {
  const kws_gather_t& g = net->gather1;
  const int n = net->K1f * net->C1;
  hipLaunchKernelGGL(fold_taps_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, params + net->conv1, w1f, g.taps, g.cin,
                     g.stride_j, net->K1f, net->C1);
  KWS_LAUNCH_CHECK("fold_taps_kernel");
  return KWS_OK;
}
int unfold_conv1(const kws_net* net, const float* g1f, float* grads, hipStream_t st) {
  const kws_gather_t& g = net->gather1;
  const int n = g.taps * g.cin * net->C1;
  hipLaunchKernelGGL(unfold_taps_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, g1f, grads + net->conv1, g.taps, g.cin,
                     g.stride_j, net->C1);
  KWS_LAUNCH_CHECK("unfold_taps_kernel");
  return KWS_OK;
}

// ---- workspace layout ------------------------------------------------------------------------------
struct Layout {
  int64_t total = 0;  // bytes
  std::vector<int64_t> y;   // y[l], l = 0..11 (float offsets)
  std::vector<int64_t> z;   // z[i], i = 0..10
  int64_t G = 0, G2 = 0, DZ = 0, bn = 0, part = 0, coef = 0, tn = 0, red = 0, swg = 0, swg1 = 0;
  std::vector<int64_t> WT;  // transposed pointwise kernel of block i (dgrad GEMM operand)
  std::vector<int64_t> tns; // weight-gradient slabs of block i: a region of its own, summed for all blocks in ONE launch
  std::vector<int64_t> WPf, WPd;  // fp16 x 2 arm: fp16 planes [2][cout][cin] (forward) / [2][cin][cout] (input gradient)
  int64_t amax;                   // fp16 x 2 arm: 3 nb slot groups of |x| maxima: W[i] | z[i] | dy[i + 1]
  int64_t xd = 0, fd = 0, dl1 = 0, dl2 = 0, per_loss = 0, per_correct = 0, att = 0;
  int64_t w1f = 0, g1f = 0;  // folded first-convolution kernel and its gradient [K1f, C1]
  int64_t bn_stride = 0;
};

void make_layout(const kws_net* n, int B, bool training, Layout* lo) {
  Bump bp;
  const int nb = (int)n->blocks.size();
  lo->y.assign(nb + 1, 0);
  lo->z.assign(nb, 0);
  int64_t max_y = (int64_t)B * n->L1 * n->C1, max_z = 0, max_part = 0, max_dwpart = 0, max_tn = 0;
  int maxC = n->C1;
  max_part = std::max((int64_t)kws_gemm_num_row_tiles((int64_t)B * n->L1), (int64_t)kws_conv1_stats_rows((int64_t)B * n->L1)) *
             2 * n->C1;   // statistics rows of either first-convolution kernel
  max_tn = std::max(kws_gemm_tn_workspace_floats((int64_t)B * n->L1, n->K1f, n->C1),
                    kws_conv1_wgrad_workspace_floats((int64_t)B * n->L1));
  for (int i = 0; i < nb; ++i) {
    const Block& b = n->blocks[i];
    const int64_t M = (int64_t)B * b.Lout;
    if (M * b.cout > max_y) max_y = M * b.cout;
    if (M * b.cin > max_z) max_z = M * b.cin;
    const int64_t p = (int64_t)kws_gemm_num_row_tiles(M) * 2 * b.cout;
    if (p > max_part) max_part = p;
    const int64_t dp = kws_dwconv_bwd_part_floats(B, b.Lin, b.cin);
    if (dp > max_dwpart) max_dwpart = dp;
    const int64_t t = std::max(kws_gemm_tn_workspace_floats(M, b.cin, b.cout),
                               kws_gemm_tn_f16x2_workspace_floats(M, b.cin, b.cout));    // either arithmetic (run-time switch)
    if (t > max_tn) max_tn = t;
    if (b.cout > maxC) maxC = b.cout;
  }
  if (training) {
    lo->y[0] = bp.take((int64_t)B * n->L1 * n->C1);
    for (int i = 0; i < nb; ++i) {
      const Block& b = n->blocks[i];
      lo->z[i] = bp.take((int64_t)B * b.Lout * b.cin);
      lo->y[i + 1] = bp.take((int64_t)B * b.Lout * b.cout);
    }
    lo->G = bp.take(max_y);
    lo->G2 = bp.take(max_y);
    lo->DZ = bp.take(max_z);
    const int64_t tail_part = (int64_t)B * 5 * n->C;
    lo->part = bp.take(std::max(std::max(max_part, max_dwpart), tail_part));
    lo->coef = bp.take(2 * maxC);
    lo->red = bp.take((int64_t)KWS_REDUCE_SLICES * 5 * maxC);
    lo->swg = bp.take((int64_t)KWS_SMALL_WGRAD_SLICES *
                      std::max((int64_t)n->T * n->C * n->T, (int64_t)2 * n->C * n->NC));
    // round 4: the attention dense layer's slices keep a region of their own (both dense layers' slices wait for the call's slab sum)
    lo->swg1 = bp.take((int64_t)KWS_SMALL_WGRAD_SLICES * n->T * n->C * n->T);
    lo->WT.assign(nb, 0);
    for (int i = 0; i < nb; ++i) lo->WT[i] = bp.take((int64_t)n->blocks[i].cin * n->blocks[i].cout);
    lo->WPf.assign(nb, 0);
    lo->WPd.assign(nb, 0);
    for (int i = 0; i < nb; ++i) {                  // 2 fp16 per weight = 1 float; taken in every mode (1.2 M weights in all)
      lo->WPf[i] = bp.take((int64_t)n->blocks[i].cin * n->blocks[i].cout);
      lo->WPd[i] = bp.take((int64_t)n->blocks[i].cin * n->blocks[i].cout);
    }
    lo->amax = bp.take((int64_t)3 * nb * KWS_ABSMAX_WORDS);
    lo->tn = bp.take(max_tn);
    lo->tns.assign(nb, 0);
    for (int i = 0; i < nb; ++i)
      lo->tns[i] = bp.take(kws_gemm_tn_workspace_floats((int64_t)B * n->blocks[i].Lout, n->blocks[i].cin, n->blocks[i].cout));
    lo->xd = bp.take((int64_t)B * n->T * n->C);
    lo->fd = bp.take((int64_t)B * 2 * n->C);
    lo->dl1 = bp.take((int64_t)B * n->T);
    lo->dl2 = bp.take((int64_t)B * n->NC);
    lo->per_loss = bp.take(B);
    lo->per_correct = bp.take(B);
    lo->att = bp.take((int64_t)B * 16);
    lo->g1f = bp.take((int64_t)n->K1f * n->C1);
  } else {
    // inference ping-pong: two y buffers and one z buffer
    const int64_t ya = bp.take(max_y), yb = bp.take(max_y), zz = bp.take(max_z);
    lo->y[0] = ya;
    for (int i = 0; i < nb; ++i) {
      lo->z[i] = zz;
      lo->y[i + 1] = (i % 2 == 0) ? yb : ya;
    }
    lo->part = bp.take(64);
    lo->WPf.assign(nb, 0);                          // the split-GEMM arms in inference: forward planes and |x| maxima (W | z)
    for (int i = 0; i < nb; ++i) lo->WPf[i] = bp.take((int64_t)n->blocks[i].cin * n->blocks[i].cout);
    lo->amax = bp.take((int64_t)2 * nb * KWS_ABSMAX_WORDS);
  }
  lo->w1f = bp.take((int64_t)n->K1f * n->C1);
  lo->bn_stride = (4 * maxC + 63) / 64 * 64;
  lo->bn = bp.take(lo->bn_stride * (nb + 1));
  lo->total = bp.cur * 4;
}

}  // namespace

extern "C" {

int kws_net_create(const kws_net_config_t* cfg, kws_net_t** out) {
  KWS_REQUIRE(cfg && out, "net_create: NULL pointer");
  kws_net* n = new kws_net();
  n->cfg = *cfg;
  int rc;
  if (cfg->kind == KWS_NET_TS_ATTENTION) {
    rc = build_ts_attention(n);
  } else if (cfg->kind == KWS_NET_LOG_MFCC) {
    rc = lm_build(n);
  } else if (cfg->kind == KWS_NET_STEFFE) {
    rc = steffe_build(n);
  } else if (cfg->kind == KWS_NET_RESIDUAL) {
    rc = residual_build(n);
  } else if (cfg->kind == KWS_NET_MFCC_AND_RAW) {
    rc = mfcc_raw_build(n);
  } else {
    kws_set_error("net_create: kind %d not supported", cfg->kind);
    rc = KWS_E_INVALID;
  }
  if (rc != KWS_OK) {
    lm_free(n);
    delete n;
    return rc;
  }
  *out = n;
  return KWS_OK;
}

int kws_net_destroy(kws_net_t* net) {
  if (net) {
    lm_free(net);
  }
  delete net;
  return KWS_OK;
}

int64_t kws_net_num_params(const kws_net_t* net) { return net ? net->n_params : 0; }
int64_t kws_net_num_state(const kws_net_t* net) { return net ? net->n_state : 0; }
int kws_net_num_tensors(const kws_net_t* net) { return net ? (int)net->tensors.size() : 0; }

int kws_net_tensor_info(const kws_net_t* net, int idx, kws_tensor_info_t* info) {
  KWS_REQUIRE(net && info && idx >= 0 && idx < (int)net->tensors.size(), "net_tensor_info: bad index %d", idx);
  *info = net->tensors[idx];
  return KWS_OK;
}

int64_t kws_net_workspace_bytes(const kws_net_t* net, int max_batch, int training) {
  if (!net || max_batch <= 0) return 0;
  if ((net->cfg.kind == KWS_NET_LOG_MFCC || net->cfg.kind == KWS_NET_STEFFE || net->cfg.kind == KWS_NET_RESIDUAL ||
       net->cfg.kind == KWS_NET_MFCC_AND_RAW)) return lm_workspace_bytes(net, max_batch, training);
  Layout lo;
  make_layout(net, max_batch, training != 0, &lo);
  return lo.total;
}

int kws_net_debug_view(const kws_net_t* net, int batch, int training, int what, int index, int64_t* offset_floats,
                       int64_t* count) {
  KWS_REQUIRE(net && offset_floats && count && batch > 0, "net_debug_view: bad arguments");
  if ((net->cfg.kind == KWS_NET_LOG_MFCC || net->cfg.kind == KWS_NET_STEFFE || net->cfg.kind == KWS_NET_RESIDUAL ||
       net->cfg.kind == KWS_NET_MFCC_AND_RAW)) return lm_debug_view(net, batch, training, what, index, offset_floats, count);
  Layout lo;
  make_layout(net, batch, training != 0, &lo);
  const int nb = (int)net->blocks.size();
  if (what == 0) {  // pre-BN output y[index], index 0..nb
    KWS_REQUIRE(index >= 0 && index <= nb, "net_debug_view: y index %d", index);
    *offset_floats = lo.y[index];
    *count = index == 0 ? (int64_t)batch * net->L1 * net->C1
                        : (int64_t)batch * net->blocks[index - 1].Lout * net->blocks[index - 1].cout;
  } else if (what == 1) {  // depthwise output z[index]
    KWS_REQUIRE(index >= 0 && index < nb, "net_debug_view: z index %d", index);
    *offset_floats = lo.z[index];
    *count = (int64_t)batch * net->blocks[index].Lout * net->blocks[index].cin;
  } else if (what == 2) {  // bn table (scale|shift|mean|rstd) of BN index
    KWS_REQUIRE(index >= 0 && index <= nb, "net_debug_view: bn index %d", index);
    *offset_floats = lo.bn + lo.bn_stride * index;
    *count = 4 * (index == 0 ? net->C1 : net->blocks[index - 1].cout);
  } else if (what == 3) {  // attention weights [B, T] (training only)
    KWS_REQUIRE(training, "net_debug_view: att is a training-only view");
    *offset_floats = lo.att;
    *count = (int64_t)batch * net->T;
  } else {
    kws_set_error("net_debug_view: unknown view %d", what);
    return KWS_E_INVALID;
  }
  return KWS_OK;
}

int kws_net_predict(const kws_net_t* net, const float* params, const float* state, const float* x, int B,
                    float* probs, void* workspace, int64_t workspace_bytes, void* stream) {
  KWS_REQUIRE(net && params && state && x && probs && workspace && B > 0, "net_predict: bad arguments");
  if ((net->cfg.kind == KWS_NET_LOG_MFCC || net->cfg.kind == KWS_NET_STEFFE || net->cfg.kind == KWS_NET_RESIDUAL ||
       net->cfg.kind == KWS_NET_MFCC_AND_RAW))
    return lm_predict(net, params, state, x, B, probs, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  Layout lo;
  make_layout(net, B, false, &lo);
  if (lo.total > workspace_bytes) {
    kws_set_error("net_predict: workspace %lld B < %lld B needed for batch %d", (long long)workspace_bytes,
                  (long long)lo.total, B);
    return KWS_E_WORKSPACE;
  }
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)net->blocks.size();
  auto bn_at = [&](int l) { return ws + lo.bn + lo.bn_stride * l; };
  {  // the twelve inference tables in one launch (round 4: twelve 4.5 us launches per predict call before)
    KWS_REQUIRE(nb + 1 <= KWS_BN_INFER_BATCH, "net: %d blocks exceed the BatchNorm table batch", nb);
    const float *ga[KWS_BN_INFER_BATCH], *be[KWS_BN_INFER_BATCH], *mm[KWS_BN_INFER_BATCH], *mv[KWS_BN_INFER_BATCH];
    float* tb[KWS_BN_INFER_BATCH];
    int Cs[KWS_BN_INFER_BATCH];
    auto put = [&](const BnRef& r, int l) {
      ga[l] = params + r.gamma; be[l] = params + r.beta; mm[l] = state + r.mm; mv[l] = state + r.mv; tb[l] = bn_at(l); Cs[l] = r.C;
    };
    put(net->bn1, 0);
    for (int i = 0; i < nb; ++i) put(net->blocks[i].bn, i + 1);
    KWS_TRY(kws_bn_infer_prepare_batch(ga, be, mm, mv, BN_EPS, Cs, tb, nb + 1, st));
  }
  if (kws_conv1_supported(&net->gather1f, &net->gather1, net->C1)) {
    KWS_TRY(kws_conv1_fwd(x, &net->gather1f, &net->gather1, params + net->conv1, ws + lo.y[0], B, net->C1, nullptr, st));
  } else {
    KWS_TRY(fold_conv1(net, params, ws + lo.w1f, st));
    KWS_TRY(kws_gemm_gather_f32(x, &net->gather1f, ws + lo.w1f, ws + lo.y[0], B, net->C1, nullptr, st));
  }
  // the fp16 x 2 arithmetic arm (kws_net_set_gemm_mode) in inference: forward planes of the pointwise kernels, one launch;
  // a layer the arm's kernels cannot take (kws_gemm_nn_f16x2_supported) runs the f32 kernel
  const bool h2 = kws_net_get_gemm_mode(net) == 2;
  unsigned* amax0 = reinterpret_cast<unsigned*>(ws + lo.amax);
  auto w_slots = [&](int i) { return amax0 + (int64_t)i * KWS_ABSMAX_WORDS; };
  auto z_slots = [&](int i) { return amax0 + (int64_t)(nb + i) * KWS_ABSMAX_WORDS; };
  if (h2) {
    const float* sin_[24];
    void* sout[24];
    const unsigned* ssl[24];
    int64_t wn[24];
    int srows[24], scols[24], str[24];
    KWS_REQUIRE(nb <= 24, "net: %d blocks exceed the split batch", nb);
    for (int i = 0; i < nb; ++i) {
      sin_[i] = params + net->blocks[i].pw; sout[i] = ws + lo.WPf[i]; ssl[i] = w_slots(i);
      srows[i] = net->blocks[i].cin; scols[i] = net->blocks[i].cout; str[i] = 1;
      wn[i] = (int64_t)srows[i] * scols[i];
    }
    KWS_TRY(kws_absmax_batch_f32(sin_, wn, w_slots(0), nb, st));
    KWS_HIP(hipMemsetAsync(z_slots(0), 0, (size_t)nb * KWS_ABSMAX_WORDS * sizeof(unsigned), st));
    KWS_TRY(kws_f16x2_split_batch(sin_, sout, srows, scols, str, ssl, nb, st));
  }
  for (int i = 0; i < nb; ++i) {
    const Block& b = net->blocks[i];
    const int64_t M = (int64_t)B * b.Lout;
    KWS_TRY(kws_dwconv_fwd_amax_f32(ws + lo.y[i], bn_at(i), params + b.dw, ws + lo.z[i], B, b.Lin, b.Lout, b.cin, b.stride,
                                    b.pad_l, h2 ? z_slots(i) : nullptr, st));
    if (h2 && kws_gemm_nn_f16x2_supported(M, b.cin, b.cout))
      KWS_TRY(kws_gemm_nn_f16x2_f32(ws + lo.z[i], ws + lo.WPf[i], ws + lo.y[i + 1], M, b.cin, b.cout, z_slots(i), w_slots(i), nullptr, st));
    else
      KWS_TRY(kws_gemm_nn_f32(ws + lo.z[i], params + b.pw, ws + lo.y[i + 1], M, b.cin, b.cout, nullptr, st));
  }
  kws_ts_tail_args t;
  memset(&t, 0, sizeof(t));
  t.y = ws + lo.y[nb]; t.bn = bn_at(nb); t.W1 = params + net->d1k; t.b1 = params + net->d1b;
  t.W2 = params + net->d2k; t.probs = probs; t.B = B; t.T = net->T; t.C = net->C; t.NC = net->NC;
  t.keep_prob = 1.f; t.loss_batch = 1; t.train = 0;
  return kws_ts_tail_launch(&t, st);
}

// Arithmetic of the pointwise GEMMs of ONE net handle: 0 = f32 MFMA (the product path), 2 = power-of-two scaled fp16 x 2 split
// products (A/B arm, gemm_f16x2.hip).  Kept on the handle (no process-wide switch); bench.py's A/B leg flips it between steps.
extern "C" int kws_net_get_gemm_mode(const kws_net_t* net) { return net ? net->gemm_mode.load(std::memory_order_relaxed) : 0; }
extern "C" int kws_net_set_gemm_mode(kws_net_t* net, int mode) {
  KWS_REQUIRE(net != nullptr, "net_set_gemm_mode: NULL net");
  KWS_REQUIRE(mode >= 0 && mode <= 2,
              "net_set_gemm_mode: mode %d (0 = f32 MFMA, 1 = f32 MFMA with separate input- / weight-gradient launches, 2 = fp16 x 2 split)", mode);
  net->gemm_mode.store(mode, std::memory_order_relaxed);
  return KWS_OK;
}

// part 0: the whole step.  part 1: forward, tail and the backward pass down to block `split` (inclusive); part 2: the rest
// of the backward pass (blocks split-1 .. 0 and the first convolution).  Parts 1 + 2 enqueue exactly the launches of
// part 0 in the same order - every intermediate lives in the caller's workspace - so the gradients are bit-identical.
static int ts_train(const kws_net_t* net, const float* params, float* state, const float* x, const float* y_onehot, int B,
                    float* grads, float* probs, float* metrics, uint64_t seed, uint32_t step, int64_t row_offset,
                    int loss_batch, void* workspace, int64_t workspace_bytes, void* stream, int phase, int split) {
  Layout lo;
  make_layout(net, B, true, &lo);
  if (lo.total > workspace_bytes) {
    kws_set_error("net_train_fwd_bwd: workspace %lld B < %lld B needed for batch %d", (long long)workspace_bytes,
                  (long long)lo.total, B);
    return KWS_E_WORKSPACE;
  }
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)net->blocks.size();
  auto bn_at = [&](int l) { return ws + lo.bn + lo.bn_stride * l; };
  float* part = ws + lo.part;
  float* DZ = ws + lo.DZ;
  float* coef = ws + lo.coef;
  float* red = ws + lo.red;

  const bool run_head = phase != 2;                   // forward + tail + the late blocks' backward
  // fp16 x 2 arm (kws_net_set_gemm_mode(net, 2), gemm_f16x2.hip): the pointwise forward, input-gradient and weight-gradient
  // GEMMs run as three f16 MFMA products of scaled two-way operand splits; the first convolution stays f32, and so does any
  // GEMM whose shape the arm's kernels cannot take (kws_gemm_*_f16x2_supported: K granule, 2 GB buffer views)
  const bool h2 = kws_net_get_gemm_mode(net) == 2;
  const int gmode = kws_net_get_gemm_mode(net);
  const bool pair_bwd = gmode == 0;                          // mode 1: f32 MFMA with separate dgrad / wgrad launches
  // fp16 x 2 arm: slot groups of the operands' |x| maxima (common.h): W of block i, z of block i, dy of block i's output
  unsigned* amax0 = reinterpret_cast<unsigned*>(ws + lo.amax);
  auto w_slots = [&](int i) { return amax0 + (int64_t)i * KWS_ABSMAX_WORDS; };
  auto z_slots = [&](int i) { return amax0 + (int64_t)(nb + i) * KWS_ABSMAX_WORDS; };
  auto g_slots = [&](int i) { return amax0 + (int64_t)(2 * nb + i) * KWS_ABSMAX_WORDS; };
  auto transpose_all = [&]() -> int {   // the pointwise kernels [cin][cout] -> [cout][cin]: all of them in one launch
    static_assert(KWS_TRANSPOSE_BATCH >= 11, "one batch holds every block");
    const float* tin[KWS_TRANSPOSE_BATCH];
    float* tout[KWS_TRANSPOSE_BATCH];
    int trows[KWS_TRANSPOSE_BATCH], tcols[KWS_TRANSPOSE_BATCH];
    KWS_REQUIRE(nb <= KWS_TRANSPOSE_BATCH, "net: %d blocks exceed the transpose batch", nb);
    for (int i = 0; i < nb; ++i) {
      tin[i] = params + net->blocks[i].pw; tout[i] = ws + lo.WT[i];
      trows[i] = net->blocks[i].cin; tcols[i] = net->blocks[i].cout;
    }
    return kws_transpose_batch_f32(tin, tout, trows, tcols, nb, st);
  };
  kws_ts_tail_args t;
  memset(&t, 0, sizeof(t));
  int tail_S = 0;                                     // > 0: the tail's dense weight gradients wait as slices for this call's slab sum
  if (run_head) {
  KWS_HIP(hipMemsetAsync(grads, 0, (size_t)net->n_params * 4, st));
  if (h2) {                                         // maxima of the pointwise kernels (this also zeroes their groups), then
    const float* win[24];                           // zero the activations' groups, then both operand forms as fp16 planes
    int64_t wn[24];
    const float* sin_[24];
    void* sout[24];
    const unsigned* ssl[24];
    int srows[24], scols[24], str[24];
    KWS_REQUIRE(2 * nb <= 24, "net: %d blocks exceed the split batch", nb);
    for (int i = 0; i < nb; ++i) {
      win[i] = params + net->blocks[i].pw;
      wn[i] = (int64_t)net->blocks[i].cin * net->blocks[i].cout;
      sin_[2 * i] = sin_[2 * i + 1] = params + net->blocks[i].pw;
      srows[2 * i] = srows[2 * i + 1] = net->blocks[i].cin;
      scols[2 * i] = scols[2 * i + 1] = net->blocks[i].cout;
      ssl[2 * i] = ssl[2 * i + 1] = w_slots(i);
      sout[2 * i] = ws + lo.WPf[i]; str[2 * i] = 1;
      sout[2 * i + 1] = ws + lo.WPd[i]; str[2 * i + 1] = 0;
    }
    KWS_TRY(kws_absmax_batch_f32(win, wn, w_slots(0), nb, st));
    KWS_HIP(hipMemsetAsync(z_slots(0), 0, (size_t)2 * nb * KWS_ABSMAX_WORDS * sizeof(unsigned), st));
    KWS_TRY(kws_f16x2_split_batch(sin_, sout, srows, scols, str, ssl, 2 * nb, st));
  }
  // ---------------- forward ----------------
  {
    const int64_t M = (int64_t)B * net->L1;
    if (kws_conv1_supported(&net->gather1f, &net->gather1, net->C1)) {
      KWS_TRY(kws_conv1_fwd(x, &net->gather1f, &net->gather1, params + net->conv1, ws + lo.y[0], B, net->C1, part, st));
    } else {
      KWS_TRY(fold_conv1(net, params, ws + lo.w1f, st));
      KWS_TRY(kws_gemm_gather_f32(x, &net->gather1f, ws + lo.w1f, ws + lo.y[0], B, net->C1, part, st));
    }
    const int rows1 = kws_conv1_supported(&net->gather1f, &net->gather1, net->C1) ? kws_conv1_stats_rows(M) : kws_gemm_gather_stats_rows(M);
    KWS_TRY(kws_bn_stats_finalize(part, rows1, M, net->C1, params + net->bn1.gamma,
                                  params + net->bn1.beta, BN_EPS, BN_MOMENTUM, state + net->bn1.mm, state + net->bn1.mv,
                                  bn_at(0), red, st));
  }
  for (int i = 0; i < nb; ++i) {
    const Block& b = net->blocks[i];
    const int64_t M = (int64_t)B * b.Lout;
    KWS_TRY(kws_dwconv_fwd_amax_f32(ws + lo.y[i], bn_at(i), params + b.dw, ws + lo.z[i], B, b.Lin, b.Lout, b.cin, b.stride,
                                    b.pad_l, h2 ? z_slots(i) : nullptr, st));
    int stat_rows;
    if (h2 && kws_gemm_nn_f16x2_supported(M, b.cin, b.cout)) {
      KWS_TRY(kws_gemm_nn_f16x2_f32(ws + lo.z[i], ws + lo.WPf[i], ws + lo.y[i + 1], M, b.cin, b.cout, z_slots(i), w_slots(i), part, st));
      stat_rows = kws_gemm_nn_f16x2_stats_rows(M);
    } else {
      KWS_TRY(kws_gemm_nn_f32(ws + lo.z[i], params + b.pw, ws + lo.y[i + 1], M, b.cin, b.cout, part, st));
      stat_rows = kws_gemm_nn_stats_rows(M, b.cin, b.cout);
    }
    KWS_TRY(kws_bn_stats_finalize(part, stat_rows, M, b.cout, params + b.bn.gamma, params + b.bn.beta,
                                  BN_EPS, BN_MOMENTUM, state + b.bn.mm, state + b.bn.mv, bn_at(i + 1), red, st));
  }
  // ---------------- tail forward + backward ----------------
  t.y = ws + lo.y[nb]; t.bn = bn_at(nb); t.W1 = params + net->d1k; t.b1 = params + net->d1b;
  t.W2 = params + net->d2k; t.labels = y_onehot; t.probs = probs; t.g = ws + ((nb % 2) ? lo.G2 : lo.G); t.part = part; t.xd = ws + lo.xd;
  t.fd = ws + lo.fd; t.dl1 = ws + lo.dl1; t.dl2 = ws + lo.dl2; t.per_loss = ws + lo.per_loss;
  t.per_correct = ws + lo.per_correct; t.att = ws + lo.att; t.B = B; t.T = net->T; t.C = net->C; t.NC = net->NC; t.seed = seed;
  t.step = step; t.keep_prob = DROP_KEEP; t.label_smoothing = LABEL_SMOOTH; t.loss_batch = loss_batch;
  t.row_offset = row_offset; t.train = 1;
  KWS_TRY(kws_ts_tail_launch(&t, st));
  // off the dependency chain: metrics, the two dense layers' weight gradients, the attention bias gradient.  Gemm mode 0: ONE
  // launch writes the metrics, the bias gradient and the weight gradients' SLICES (tail.hip tail_post_kernel); the slices are
  // summed with the pointwise layers' slabs at the end of this call (in kws_small_wgrad_launch's own order: bit-identical).
  // Mode 1 / shapes the fused kernel does not take: the six launches of rounds 1 - 3.
  bool tail_fused = false;
  if (pair_bwd) {
    kws_tail_post_args tp;
    tp.X2 = t.fd; tp.D2 = t.dl2; tp.ws2 = ws + lo.swg; tp.K2 = 2 * net->C; tp.N2 = net->NC;
    tp.X1 = t.xd; tp.D1 = t.dl1; tp.ws1 = ws + lo.swg1; tp.K1 = net->T * net->C; tp.N1 = net->T;
    tp.bias1 = grads + net->d1b; tp.per_loss = t.per_loss; tp.per_correct = t.per_correct; tp.metrics = metrics; tp.B = B;
    int S_tail = 0;
    const int rc = kws_tail_post_launch(&tp, &S_tail, st);
    if (rc < 0) return rc;
    if (rc == 0) {
      tail_fused = true;
      tail_S = S_tail;
    }
  }
  if (!tail_fused) {
  KWS_TRY(kws_metrics_launch(t.per_loss, t.per_correct, B, metrics, st));
  KWS_TRY(kws_small_wgrad_launch(t.fd, t.dl2, grads + net->d2k, nullptr, B, 2 * net->C, net->NC, ws + lo.swg, st));
  KWS_TRY(kws_small_wgrad_launch(t.xd, t.dl1, grads + net->d1k, grads + net->d1b, B, net->T * net->C, net->T,
                                 ws + lo.swg, st));
  }
  {
    const BnRef& r = net->blocks[nb - 1].bn;
    KWS_TRY(kws_dw_bwd_finalize(part, B, (int64_t)B * net->T, r.C, nullptr, grads + r.gamma,
                                grads + r.beta, coef, red, st));
  }
  }  // run_head
  // ---------------- backward through the blocks ----------------
  // One stream: the weight-gradient GEMM of a block depends only on dy and z and COULD run beside the HBM-bound depthwise /
  // BN kernels, but the early-layer GEMMs (32 FLOP/B) draw 3.7 TB/s themselves - with the fused two-pass depthwise
  // backward a side-stream program took the same 5.14 - 5.20 ms (measured in rounds 1 and 2, then removed).
  // The f32 dgrad GEMMs read the pointwise kernels transposed (the fp16 arm too, for the layers it hands back)
  if (run_head) KWS_TRY(transpose_all());
  float* Gb[2] = {ws + lo.G, ws + lo.G2};          // gradient wrt y[l] lives in Gb[l % 2]; the tail wrote Gb[nb % 2]
  // the f32 weight-gradient GEMMs leave their slabs in per-block regions; ONE launch sums them when this call's blocks are done
  const float* sl_ws[KWS_SLAB_BATCH];
  float* sl_out[KWS_SLAB_BATCH];
  int64_t sl_n[KWS_SLAB_BATCH];
  int sl_S[KWS_SLAB_BATCH], n_sl = 0;
  KWS_REQUIRE(nb + 2 <= KWS_SLAB_BATCH, "net: %d blocks exceed the slab batch", nb);
  if (tail_S > 0) {   // (negative S: kws_small_wgrad_launch's summation order)
    sl_ws[n_sl] = ws + lo.swg; sl_out[n_sl] = grads + net->d2k; sl_n[n_sl] = (int64_t)2 * net->C * net->NC; sl_S[n_sl] = -tail_S; ++n_sl;
    sl_ws[n_sl] = ws + lo.swg1; sl_out[n_sl] = grads + net->d1k; sl_n[n_sl] = (int64_t)net->T * net->C * net->T; sl_S[n_sl] = -tail_S; ++n_sl;
  }
  const int i_hi = phase == 2 ? split - 1 : nb - 1, i_lo = phase == 1 ? split : 0;
  for (int i = i_hi; i >= i_lo; --i) {
    const Block& b = net->blocks[i];
    const int64_t M = (int64_t)B * b.Lout;
    float* Gcur = Gb[(i + 1) % 2];
    float* Gnext = Gb[i % 2];
    // Gcur holds dy of this block's pointwise output: written by the depthwise backward of block i+1 (pass 2
    // below); only the tail hands over a masked gradient that still needs its BatchNorm backward
    if (i == nb - 1)
      KWS_TRY(kws_bn_bwd_apply_amax(Gcur, ws + lo.y[i + 1], bn_at(i + 1), params + b.bn.gamma, coef, M, b.cout,
                                    h2 ? g_slots(i) : nullptr, st));
    // f32 arm, gemm mode 0 (default): the input-gradient GEMM and the weight-gradient GEMM of the layer - independent of each
    // other - go out as ONE launch whose weight-gradient workgroups start on a CU as soon as its input-gradient workgroup has
    // ended (gemm.hip gemm_dgrad_wgrad_kernel; same code, bit-identical dZ and slabs).  Mode 1 keeps the two launches of
    // rounds 1 - 3 (the A/B reference), and so does any shape the fused kernel does not take (rc 1).
    bool paired = false;
    if (pair_bwd) {
      const int rc = kws_gemm_dgrad_wgrad_f32(Gcur, ws + lo.WT[i], DZ, ws + lo.z[i], M, b.cin, b.cout, ws + lo.tns[i], &sl_S[n_sl], st);
      if (rc < 0) return rc;
      if (rc == 0) {
        sl_ws[n_sl] = ws + lo.tns[i]; sl_out[n_sl] = grads + b.pw; sl_n[n_sl] = (int64_t)b.cin * b.cout;
        ++n_sl;
        paired = true;
      }
    }
    if (!paired) {
    if (h2 && kws_gemm_nn_f16x2_supported(M, b.cout, b.cin))
      KWS_TRY(kws_gemm_nn_f16x2_f32(Gcur, ws + lo.WPd[i], DZ, M, b.cout, b.cin, g_slots(i), w_slots(i), nullptr, st));
    else
      KWS_TRY(kws_gemm_nn_f32(Gcur, ws + lo.WT[i], DZ, M, b.cout, b.cin, nullptr, st));
    if (h2 && kws_gemm_tn_f16x2_supported(M, b.cin, b.cout))
      KWS_TRY(kws_gemm_tn_f16x2_f32(ws + lo.z[i], Gcur, grads + b.pw, M, b.cin, b.cout, z_slots(i), g_slots(i), ws + lo.tn, st));
    else {
      sl_ws[n_sl] = ws + lo.tns[i]; sl_out[n_sl] = grads + b.pw; sl_n[n_sl] = (int64_t)b.cin * b.cout;
      KWS_TRY(kws_gemm_tn_slabs_f32(ws + lo.z[i], Gcur, M, b.cin, b.cout, ws + lo.tns[i], &sl_S[n_sl], st));
      ++n_sl;
    }
    }
    const BnRef& prev = (i == 0) ? net->bn1 : net->blocks[i - 1].bn;
    // depthwise backward + BatchNorm backward of this block's input without materialising the masked
    // gradient: reduce, fold (dw, dgamma, dbeta, c1 | c2), recompute and write dy of the previous block
    KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.y[i], bn_at(i), params + b.dw, nullptr, nullptr, part, 1, B, b.Lin, b.Lout,
                                  b.cin, b.stride, b.pad_l, st));
    const int n_parts = (int)(kws_dwconv_bwd_part_floats(B, b.Lin, b.cin) / (5 * b.cin));
    KWS_TRY(kws_dw_bwd_finalize(part, n_parts, (int64_t)B * b.Lin, b.cin, grads + b.dw,
                                grads + prev.gamma, grads + prev.beta, coef, red, st));
    KWS_TRY(kws_dwconv_bwd_bn_amax_f32(DZ, ws + lo.y[i], bn_at(i), params + b.dw, coef, Gnext, nullptr, 2, B, b.Lin, b.Lout,
                                       b.cin, b.stride, b.pad_l, (h2 && i > 0) ? g_slots(i - 1) : nullptr, st));
  }
  // the slab sums of this call's pointwise weight gradients: beside the first convolution's weight gradient in ONE launch when
  // that kernel runs in this call (round 4: conv1_wgrad_slabsum_kernel - the two are independent, one is MFMA / memory bound, the
  // other HBM bound), their own launch otherwise (a part's gradients are final when it returns)
  const bool sum_with_conv1 = n_sl > 0 && phase != 1 && kws_conv1_supported(&net->gather1f, &net->gather1, net->C1) &&
                              kws_net_get_gemm_mode(net) != 1;
  if (n_sl > 0 && !sum_with_conv1) KWS_TRY(kws_reduce_slabs_batch(sl_ws, sl_out, sl_n, sl_S, n_sl, st));
  if (phase != 1) {
    const int64_t M = (int64_t)B * net->L1;
    (void)M;                                        // Gb[0] already holds dy of the first convolution
    if (kws_conv1_supported(&net->gather1f, &net->gather1, net->C1)) {
      KWS_TRY(kws_conv1_wgrad_slabs(x, &net->gather1f, &net->gather1, Gb[0], grads + net->conv1, B, net->C1, ws + lo.tn,
                                    sl_ws, sl_out, sl_n, sl_S, sum_with_conv1 ? n_sl : 0, st));
    } else {
      KWS_TRY(kws_gemm_tn_gather_f32(x, &net->gather1f, Gb[0], ws + lo.g1f, B, net->C1, ws + lo.tn, st));
      KWS_TRY(unfold_conv1(net, ws + lo.g1f, grads, st));
    }
  }
  return KWS_OK;
}

int kws_net_train_fwd_bwd(const kws_net_t* net, const float* params, float* state, const float* x,
                          const float* y_onehot, int B, float* grads, float* probs, float* metrics, uint64_t seed,
                          uint32_t step, int64_t row_offset, int loss_batch, void* workspace, int64_t workspace_bytes,
                          void* stream) {
  KWS_REQUIRE(net && params && state && x && y_onehot && grads && probs && metrics && workspace && B > 0,
              "net_train_fwd_bwd: bad arguments");
  KWS_REQUIRE(loss_batch >= B, "net_train_fwd_bwd: loss_batch %d < B %d", loss_batch, B);
  if ((net->cfg.kind == KWS_NET_LOG_MFCC || net->cfg.kind == KWS_NET_STEFFE || net->cfg.kind == KWS_NET_RESIDUAL ||
       net->cfg.kind == KWS_NET_MFCC_AND_RAW))
    return lm_train(net, params, state, x, y_onehot, B, grads, probs, metrics, seed, step, row_offset, loss_batch,
                    (float*)workspace, workspace_bytes, (hipStream_t)stream);
  return ts_train(net, params, state, x, y_onehot, B, grads, probs, metrics, seed, step, row_offset, loss_batch, workspace,
                  workspace_bytes, stream, 0, 0);
}

int kws_net_num_blocks(const kws_net_t* net) {
  return (net && net->cfg.kind == KWS_NET_TS_ATTENTION) ? (int)net->blocks.size() : 0;
}

int64_t kws_net_grad_ready_offset(const kws_net_t* net, int split_block) {
  if (!net || net->cfg.kind != KWS_NET_TS_ATTENTION || split_block < 1 || split_block >= (int)net->blocks.size()) return -1;
  // block `split_block`'s backward writes the gradients of the BatchNorm in front of it; everything from there to the end
  // of the flat buffer (Keras layer order) is final once part 1 has run
  return net->blocks[split_block - 1].bn.gamma;
}

int kws_net_train_fwd_bwd_part(const kws_net_t* net, const float* params, float* state, const float* x,
                               const float* y_onehot, int B, float* grads, float* probs, float* metrics, uint64_t seed,
                               uint32_t step, int64_t row_offset, int loss_batch, void* workspace,
                               int64_t workspace_bytes, int part, int split_block, void* stream) {
  KWS_REQUIRE(net && params && state && x && y_onehot && grads && probs && metrics && workspace && B > 0,
              "net_train_fwd_bwd_part: bad arguments");
  KWS_REQUIRE(loss_batch >= B, "net_train_fwd_bwd_part: loss_batch %d < B %d", loss_batch, B);
  KWS_REQUIRE(net->cfg.kind == KWS_NET_TS_ATTENTION, "net_train_fwd_bwd_part: only the raw-waveform attention net is split");
  KWS_REQUIRE((part == 1 || part == 2) && split_block >= 1 && split_block < (int)net->blocks.size(),
              "net_train_fwd_bwd_part: part %d split_block %d", part, split_block);
  return ts_train(net, params, state, x, y_onehot, B, grads, probs, metrics, seed, step, row_offset, loss_batch, workspace,
                  workspace_bytes, stream, part, split_block);
}

}  // extern "C"
