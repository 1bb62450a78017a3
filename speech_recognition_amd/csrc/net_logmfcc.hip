// Network program of conv_1d_log_mfcc_model (reference model.py:1400-1479; SURVEY 8a row a19, layer
// table Appendix B.2): Conv1D(64,3)+BN+ReLU6 on [98,40] features, 10 residual blocks of
// 2 x [depthwise k3 SAME -> pointwise -> BN -> ReLU6] + MaxPool1D(pool=stride) + Add (1x1 stride-2
// Conv1D + BN shortcut on the strided blocks), softmax-over-time attention, GAP, Dropout(.2),
// Dense(num_classes)+softmax, categorical cross-entropy.
//
// Same building blocks as the raw-waveform net (f32-MFMA GEMMs with BN statistics in the epilogue,
// depthwise kernels applying BN+ReLU6 on load, fixed-order partial-slab reductions) plus resblock.hip.
// Keras variable names follow per-class auto-numbering in layer creation order.
#include "net_internal.h"

namespace {

constexpr float DROP_KEEP = 0.8f;  // Dropout(0.2), model.py:1471

struct LmBlock {
  int nf, stride, cin, Lin, Lout;
  bool has_short;
  int64_t ws;
  BnRef bns;
  int bns_idx;
  kws_gather_t gs;
  int64_t dw1, pw1, dw2, pw2;
  BnRef bn1, bn2;
  int bn1_idx, bn2_idx;
};

}  // namespace

struct LmProgram {
  int T0, F, C0, L0;
  int Fp;  // F rounded up to the 16-byte vectors of the gathered GEMM (spectrogram input: 257 -> 260)
  int64_t conv1;
  BnRef bn0;
  kws_gather_t g0;
  std::vector<LmBlock> blocks;
  int T, C, NC;
  int64_t att_dw, att_pw, dk, db;
  BnRef att_bn;
  int att_bn_idx;
  int n_bn;
};

namespace {

struct LmLayout {
  int64_t total = 0;
  int64_t y0 = 0, a0 = 0;
  std::vector<int64_t> ys, z1, y1, z2, y2, o;
  int64_t bn = 0, bn_stride = 0, part = 0, red = 0, coef = 0, WT = 0, tn = 0, swg = 0;
  int64_t dOa = 0, dOb = 0, G = 0, DZ = 0, DXS = 0;
  int64_t u = 0, fd = 0, dl = 0, gu = 0, coef2 = 0, per_loss = 0, per_correct = 0, att = 0;
  int64_t xpad = 0, wpad = 0, gwpad = 0;  // only when Fp != F
};

void lm_layout(const kws_net* n, int B, LmLayout* lo) {
  const LmProgram& p = *n->lm;
  Bump bp;
  const int nb = (int)p.blocks.size();
  lo->ys.assign(nb, 0); lo->z1.assign(nb, 0); lo->y1.assign(nb, 0); lo->z2.assign(nb, 0); lo->y2.assign(nb, 0);
  lo->o.assign(nb, 0);
  lo->y0 = bp.take((int64_t)B * p.L0 * p.C0);
  lo->a0 = bp.take((int64_t)B * p.L0 * p.C0);
  int64_t max_o = (int64_t)B * p.L0 * p.C0, max_y = max_o, max_z = 0, max_xs = 64, max_part = 0, max_wt = 0, max_tn = 0;
  auto upd_gemm = [&](int64_t M, int K, int N) {
    max_part = std::max(max_part, (int64_t)kws_gemm_num_row_tiles(M) * 2 * N);
    max_wt = std::max(max_wt, (int64_t)K * N);
    max_tn = std::max(max_tn, kws_gemm_tn_workspace_floats(M, K, N));
  };
  upd_gemm((int64_t)B * p.L0, 3 * p.Fp, p.C0);
  max_part = std::max(max_part, kws_block_out_bwd_part_floats(B, p.L0, p.C0, 1));
  for (int i = 0; i < nb; ++i) {
    const LmBlock& b = p.blocks[i];
    if (b.has_short) lo->ys[i] = bp.take((int64_t)B * b.Lout * b.nf);
    lo->z1[i] = bp.take((int64_t)B * b.Lin * b.cin);
    lo->y1[i] = bp.take((int64_t)B * b.Lin * b.nf);
    lo->z2[i] = bp.take((int64_t)B * b.Lin * b.nf);
    lo->y2[i] = bp.take((int64_t)B * b.Lin * b.nf);
    lo->o[i] = bp.take((int64_t)B * b.Lout * b.nf);
    max_o = std::max(max_o, std::max((int64_t)B * b.Lout * b.nf, (int64_t)B * b.Lin * b.cin));
    max_y = std::max(max_y, (int64_t)B * b.Lin * b.nf);
    max_z = std::max(max_z, std::max((int64_t)B * b.Lin * b.cin, (int64_t)B * b.Lin * b.nf));
    max_xs = std::max(max_xs, (int64_t)B * b.Lout * b.cin);
    upd_gemm((int64_t)B * b.Lin, b.cin, b.nf);
    upd_gemm((int64_t)B * b.Lin, b.nf, b.nf);
    if (b.has_short) upd_gemm((int64_t)B * b.Lout, b.cin, b.nf);
    max_part = std::max(max_part, kws_dwconv_bwd_part_floats(B, b.Lin, b.cin));
    max_part = std::max(max_part, kws_dwconv_bwd_part_floats(B, b.Lin, b.nf));
    max_part = std::max(max_part, kws_block_out_bwd_part_floats(B, b.Lin, b.nf, b.stride));
    max_part = std::max(max_part, kws_block_out_bwd_part_floats(B, b.Lout, b.nf, 1));
  }
  max_part = std::max(max_part, (int64_t)B * 5 * p.C);
  lo->bn_stride = 4 * 256;
  lo->bn = bp.take(lo->bn_stride * (p.n_bn + 1));
  lo->part = bp.take(max_part);
  lo->red = bp.take((int64_t)KWS_REDUCE_SLICES * 5 * 256);
  lo->coef = bp.take(2 * 256);
  lo->WT = bp.take(max_wt);
  lo->tn = bp.take(max_tn);
  lo->swg = bp.take((int64_t)KWS_SMALL_WGRAD_SLICES * p.C * p.NC);
  lo->dOa = bp.take(max_o);
  lo->dOb = bp.take(max_o);
  lo->G = bp.take(max_y);
  lo->DZ = bp.take(max_z);
  lo->DXS = bp.take(max_xs);
  lo->u = bp.take((int64_t)B * 16);
  lo->fd = bp.take((int64_t)B * p.C);
  lo->dl = bp.take((int64_t)B * p.NC);
  lo->gu = bp.take((int64_t)B * 16);
  lo->coef2 = bp.take(4);
  lo->per_loss = bp.take(B);
  lo->per_correct = bp.take(B);
  lo->att = bp.take((int64_t)B * 16);
  if (p.Fp != p.F) {
    lo->xpad = bp.take((int64_t)B * p.T0 * p.Fp);
    lo->wpad = bp.take((int64_t)3 * p.Fp * p.C0);
    lo->gwpad = bp.take((int64_t)3 * p.Fp * p.C0);
  }
  lo->total = bp.cur * 4;
}

struct Ctx {
  const kws_net* n;
  const LmProgram* p;
  const float* params;
  float* state;        // may be written (training)
  float* ws;
  LmLayout lo;
  int B;
  bool training;
  hipStream_t st;
  float* bn_at(int idx) const { return ws + lo.bn + lo.bn_stride * idx; }
};

// BN statistics -> table (training) or moving statistics -> table (inference)
int bn_table(const Ctx& c, const BnRef& r, int idx, int64_t M, int stat_rows) {
  if (c.training)
    return kws_bn_stats_finalize(c.ws + c.lo.part, stat_rows, M, r.C, c.params + r.gamma,
                                 c.params + r.beta, KWS_BN_EPS, KWS_BN_MOMENTUM, c.state + r.mm, c.state + r.mv,
                                 c.bn_at(idx), c.ws + c.lo.red, c.st);
  return kws_bn_infer_prepare(c.params + r.gamma, c.params + r.beta, c.state + r.mm, c.state + r.mv, KWS_BN_EPS, r.C,
                              c.bn_at(idx), c.st);
}

// rows of F floats -> rows of Fp floats (Fp % 4 == 0), zero filled; one 16-byte store per thread
__global__ __launch_bounds__(256) void repitch_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t rows,
                                                      int F, int Fp) {
  const int q = Fp / 4;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < rows * q; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / q;
    const int f = (int)(i - r * q) * 4;
    const float* src = in + r * F + f;
    float4 v;
    v.x = f + 0 < F ? src[0] : 0.f;
    v.y = f + 1 < F ? src[1] : 0.f;
    v.z = f + 2 < F ? src[2] : 0.f;
    v.w = f + 3 < F ? src[3] : 0.f;
    *reinterpret_cast<float4*>(out + r * Fp + f) = v;
  }
}

// Feature counts that are not a multiple of 4 (conv_1d_spectrogram: 257 bins): the gathered GEMM reads 16-byte
// vectors, so the input rows and the [3, F, 64] kernel are re-pitched to Fp with zero columns / rows (plain 2-D
// copies), and the padded weight gradient is copied back without them.
int pad_first_conv(const Ctx& c, const float* x, const float** x_used, const float** w_used) {
  const LmProgram& p = *c.p;
  *x_used = x;
  *w_used = c.params + p.conv1;
  if (p.Fp == p.F) return KWS_OK;
  float* xp = c.ws + c.lo.xpad;
  float* wp = c.ws + c.lo.wpad;
  const size_t rows = (size_t)c.B * p.T0;
  const int64_t n4 = (int64_t)rows * (p.Fp / 4);
  hipLaunchKernelGGL(repitch_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n4, 256), 8192)), dim3(256), 0, c.st, x, xp,
                     (int64_t)rows, p.F, p.Fp);
  KWS_LAUNCH_CHECK("repitch_kernel");
  KWS_HIP(hipMemsetAsync(wp, 0, (size_t)3 * p.Fp * p.C0 * 4, c.st));
  KWS_HIP(hipMemcpy2DAsync(wp, (size_t)p.Fp * p.C0 * 4, c.params + p.conv1, (size_t)p.F * p.C0 * 4, (size_t)p.F * p.C0 * 4, 3,
                           hipMemcpyDeviceToDevice, c.st));
  *x_used = xp;
  *w_used = wp;
  return KWS_OK;
}

int forward(const Ctx& c, const float* x, kws_lm_tail_args* t) {
  const LmProgram& p = *c.p;
  const LmLayout& lo = c.lo;
  float* ws = c.ws;
  const int B = c.B;
  float* stats = c.training ? ws + lo.part : nullptr;
  const float *x0, *w0;
  KWS_TRY(pad_first_conv(c, x, &x0, &w0));
  KWS_TRY(kws_gemm_gather_f32(x0, &p.g0, w0, ws + lo.y0, B, p.C0, stats, c.st));
  KWS_TRY(bn_table(c, p.bn0, 1, (int64_t)B * p.L0, kws_gemm_gather_stats_rows((int64_t)B * p.L0)));
  KWS_TRY(kws_bn_relu6_apply(ws + lo.y0, c.bn_at(1), ws + lo.a0, (int64_t)B * p.L0, p.C0, 1, c.st));
  const float* xin = ws + lo.a0;
  for (size_t i = 0; i < p.blocks.size(); ++i) {
    const LmBlock& b = p.blocks[i];
    const int64_t M = (int64_t)B * b.Lin;
    if (b.has_short) {
      KWS_TRY(kws_gemm_gather_f32(xin, &b.gs, c.params + b.ws, ws + lo.ys[i], B, b.nf, stats, c.st));
      KWS_TRY(bn_table(c, b.bns, b.bns_idx, (int64_t)B * b.Lout, kws_gemm_gather_stats_rows((int64_t)B * b.Lout)));
    }
    KWS_TRY(kws_dwconv_fwd_f32(xin, nullptr, c.params + b.dw1, ws + lo.z1[i], B, b.Lin, b.Lin, b.cin, 1, 1, c.st));
    KWS_TRY(kws_gemm_nn_f32(ws + lo.z1[i], c.params + b.pw1, ws + lo.y1[i], M, b.cin, b.nf, stats, c.st));
    KWS_TRY(bn_table(c, b.bn1, b.bn1_idx, M, kws_gemm_nn_stats_rows(M, b.cin, b.nf)));
    KWS_TRY(kws_dwconv_fwd_f32(ws + lo.y1[i], c.bn_at(b.bn1_idx), c.params + b.dw2, ws + lo.z2[i], B, b.Lin, b.Lin,
                               b.nf, 1, 1, c.st));
    KWS_TRY(kws_gemm_nn_f32(ws + lo.z2[i], c.params + b.pw2, ws + lo.y2[i], M, b.nf, b.nf, stats, c.st));
    KWS_TRY(bn_table(c, b.bn2, b.bn2_idx, M, kws_gemm_nn_stats_rows(M, b.nf, b.nf)));
    KWS_TRY(kws_block_out_fwd(ws + lo.y2[i], c.bn_at(b.bn2_idx), b.has_short ? ws + lo.ys[i] : xin,
                              b.has_short ? c.bn_at(b.bns_idx) : nullptr, ws + lo.o[i], B, b.Lin, b.nf, b.stride, c.st));
    xin = ws + lo.o[i];
  }
  memset(t, 0, sizeof(*t));
  t->x = xin; t->wa = c.params + p.att_dw; t->Wa = c.params + p.att_pw;
  t->bn_gamma = c.params + p.att_bn.gamma; t->bn_beta = c.params + p.att_bn.beta;
  t->mm = c.state + p.att_bn.mm; t->mv = c.state + p.att_bn.mv;
  t->Wd = c.params + p.dk; t->bd = c.params + p.db;
  t->u = ws + lo.u; t->bn = c.bn_at(p.att_bn_idx);
  t->B = B; t->T = p.T; t->C = p.C; t->NC = p.NC; t->keep_prob = c.training ? DROP_KEEP : 1.f; t->loss_batch = 1;
  return KWS_OK;
}

}  // namespace

int lm_build(kws_net* n) {
  const kws_net_config_t& c = n->cfg;
  KWS_REQUIRE(c.num_classes >= 2 && c.num_classes <= 64, "net: num_classes %d out of range", c.num_classes);
  KWS_REQUIRE(c.spectrogram_length >= 19 && c.num_features >= 4 &&
                  c.input_size == c.spectrogram_length * c.num_features,
              "net: log-mfcc input %d != %d x %d", c.input_size, c.spectrogram_length, c.num_features);
  KWS_REQUIRE((c.spectrogram_length - 2) % 8 == 0, "net: spectrogram_length-2 = %d must be a multiple of 8",
              c.spectrogram_length - 2);
  LmProgram* p = new LmProgram();
  n->lm = p;
  int n_conv = 0, n_bn = 0, n_dw = 0;
  auto conv = [&](int k, int cin, int cout, bool l2) {
    ++n_conv;
    return kws_net_add_tensor(n, "conv1d_" + std::to_string(n_conv) + "/kernel", {k, cin, cout}, false,
                              l2 ? KWS_L2_COEF : 0.f, k * cin, k * cout, 0.f);
  };
  auto bn = [&](int C, int* idx) {
    ++n_bn;
    *idx = n_bn;
    return kws_net_add_bn(n, n_bn, C);
  };
  auto dw = [&](int C) {
    ++n_dw;
    return kws_net_add_tensor(n, "depthwise_conv2d_" + std::to_string(n_dw) + "/depthwise_kernel", {1, 3, C, 1}, false,
                              KWS_L2_COEF, 3 * C, 3, 0.f);
  };
  p->T0 = c.spectrogram_length; p->F = c.num_features; p->C0 = 64; p->L0 = p->T0 - 2;
  p->conv1 = conv(3, p->F, p->C0, true);
  int idx0;
  p->bn0 = bn(p->C0, &idx0);
  kws_gather_t g0;
  p->Fp = (p->F + 3) & ~3;
  g0.L_out = p->L0; g0.cin = p->Fp; g0.taps = 3; g0.stride_t = p->Fp; g0.stride_j = p->Fp; g0.base_off = 0;
  g0.x_len = p->T0 * p->Fp; g0.x_batch_stride = p->T0 * p->Fp;
  p->g0 = g0;
  static const int spec[10][2] = {{64, 1}, {64, 1}, {128, 2}, {128, 1}, {192, 2}, {192, 1}, {192, 1}, {256, 2},
                                  {256, 1}, {256, 1}};  // model.py:1453-1462
  int cin = p->C0, L = p->L0;
  for (int i = 0; i < 10; ++i) {
    LmBlock b;
    b.nf = spec[i][0]; b.stride = spec[i][1]; b.cin = cin; b.Lin = L; b.Lout = L / b.stride;
    b.has_short = b.stride != 1;
    b.ws = 0; b.bns_idx = 0;
    memset(&b.gs, 0, sizeof(b.gs));
    if (b.has_short) {
      b.ws = conv(1, cin, b.nf, false);  // shortcut Conv1D has no kernel_regularizer (model.py:1431-1432)
      b.bns = bn(b.nf, &b.bns_idx);
      b.gs.L_out = b.Lout; b.gs.cin = cin; b.gs.taps = 1; b.gs.stride_t = b.stride * cin; b.gs.stride_j = 0;
      b.gs.base_off = 0; b.gs.x_len = L * cin; b.gs.x_batch_stride = (int64_t)L * cin;
    } else {
      KWS_REQUIRE(cin == b.nf, "net: identity shortcut needs cin == nf");
    }
    b.dw1 = dw(cin);
    b.pw1 = conv(1, cin, b.nf, true);
    b.bn1 = bn(b.nf, &b.bn1_idx);
    b.dw2 = dw(b.nf);
    b.pw2 = conv(1, b.nf, b.nf, true);
    b.bn2 = bn(b.nf, &b.bn2_idx);
    p->blocks.push_back(b);
    cin = b.nf;
    L = b.Lout;
  }
  p->T = L; p->C = cin; p->NC = c.num_classes;
  KWS_REQUIRE(p->T <= 16, "net: %d time steps at the tail (max 16)", p->T);
  p->att_dw = dw(cin);
  p->att_pw = conv(1, cin, 1, true);
  p->att_bn = bn(1, &p->att_bn_idx);
  p->dk = kws_net_add_tensor(n, "dense_1/kernel", {cin, p->NC}, false, KWS_L2_COEF, cin, p->NC, 0.f);
  p->db = kws_net_add_tensor(n, "dense_1/bias", {p->NC}, false, 0.f, 0, 0, 0.f);
  p->n_bn = n_bn;
  return KWS_OK;
}

void lm_free(kws_net* n) {
  delete n->lm;
  n->lm = nullptr;
}

int64_t lm_workspace_bytes(const kws_net* n, int B, int training) {
  (void)training;
  LmLayout lo;
  lm_layout(n, B, &lo);
  return lo.total;
}

int lm_debug_view(const kws_net* n, int B, int training, int what, int index, int64_t* offset_floats, int64_t* count) {
  (void)training;
  LmLayout lo;
  lm_layout(n, B, &lo);
  const LmProgram& p = *n->lm;
  const int nb = (int)p.blocks.size();
  // what: 0 = pre-BN tensor of BN `index` (1-based Keras numbering), 2 = BN table of BN `index`,
  //       3 = attention weights, 4 = attention logits u
  if (what == 2) {
    KWS_REQUIRE(index >= 1 && index <= p.n_bn, "lm_debug_view: bn index %d", index);
    *offset_floats = lo.bn + lo.bn_stride * index;
    *count = 4 * 256;
    return KWS_OK;
  }
  if (what == 3) { *offset_floats = lo.att; *count = (int64_t)B * p.T; return KWS_OK; }
  if (what == 4) { *offset_floats = lo.u; *count = (int64_t)B * p.T; return KWS_OK; }
  if (what == 0) {
    if (index == 1) { *offset_floats = lo.y0; *count = (int64_t)B * p.L0 * p.C0; return KWS_OK; }
    for (int i = 0; i < nb; ++i) {
      const LmBlock& b = p.blocks[i];
      if (b.has_short && index == b.bns_idx) { *offset_floats = lo.ys[i]; *count = (int64_t)B * b.Lout * b.nf; return KWS_OK; }
      if (index == b.bn1_idx) { *offset_floats = lo.y1[i]; *count = (int64_t)B * b.Lin * b.nf; return KWS_OK; }
      if (index == b.bn2_idx) { *offset_floats = lo.y2[i]; *count = (int64_t)B * b.Lin * b.nf; return KWS_OK; }
    }
  }
  kws_set_error("lm_debug_view: unknown view %d/%d", what, index);
  return KWS_E_INVALID;
}

int lm_predict(const kws_net* n, const float* params, const float* state, const float* x, int B, float* probs,
               float* ws, int64_t ws_bytes, hipStream_t st) {
  Ctx c;
  c.n = n; c.p = n->lm; c.params = params; c.state = const_cast<float*>(state); c.ws = ws; c.B = B;
  c.training = false; c.st = st;
  lm_layout(n, B, &c.lo);
  if (c.lo.total > ws_bytes) {
    kws_set_error("net_predict: workspace %lld B < %lld B needed for batch %d", (long long)ws_bytes,
                  (long long)c.lo.total, B);
    return KWS_E_WORKSPACE;
  }
  kws_lm_tail_args t;
  KWS_TRY(forward(c, x, &t));
  t.probs = probs;
  return kws_lm_tail_fwd(&t, 0, st);
}

int lm_train(const kws_net* n, const float* params, float* state, const float* x, const float* y_onehot, int B,
             float* grads, float* probs, float* metrics, uint64_t seed, uint32_t step, int64_t row_offset,
             int loss_batch, float* ws, int64_t ws_bytes, hipStream_t st) {
  Ctx c;
  c.n = n; c.p = n->lm; c.params = params; c.state = state; c.ws = ws; c.B = B; c.training = true; c.st = st;
  lm_layout(n, B, &c.lo);
  if (c.lo.total > ws_bytes) {
    kws_set_error("net_train_fwd_bwd: workspace %lld B < %lld B needed for batch %d", (long long)ws_bytes,
                  (long long)c.lo.total, B);
    return KWS_E_WORKSPACE;
  }
  const LmProgram& p = *c.p;
  const LmLayout& lo = c.lo;
  KWS_HIP(hipMemsetAsync(grads, 0, (size_t)n->n_params * 4, st));
  kws_lm_tail_args t;
  KWS_TRY(forward(c, x, &t));
  float* part = ws + lo.part;
  float* red = ws + lo.red;
  float* coef = ws + lo.coef;
  float* G = ws + lo.G;
  float* DZ = ws + lo.DZ;
  float* dO = ws + lo.dOa;
  float* dX = ws + lo.dOb;
  // ---- tail forward + backward ----
  t.labels = y_onehot; t.probs = probs; t.dX = dO; t.fd = ws + lo.fd; t.dl = ws + lo.dl; t.gu = ws + lo.gu;
  t.part = part; t.coef = ws + lo.coef2; t.d_gamma = grads + p.att_bn.gamma; t.d_beta = grads + p.att_bn.beta;
  t.per_loss = ws + lo.per_loss; t.per_correct = ws + lo.per_correct; t.att = ws + lo.att;
  t.seed = seed; t.step = step; t.loss_batch = loss_batch; t.row_offset = row_offset;
  KWS_TRY(kws_lm_tail_fwd(&t, 1, st));
  KWS_TRY(kws_metrics_launch(t.per_loss, t.per_correct, B, metrics, st));
  KWS_TRY(kws_small_wgrad_launch(t.fd, t.dl, grads + p.dk, grads + p.db, B, p.C, p.NC, ws + lo.swg, st));
  KWS_TRY(kws_lm_tail_bwd(&t, st));
  KWS_TRY(kws_dw_bwd_finalize(part, B, 1, p.C, grads + p.att_dw, nullptr, grads + p.att_pw, nullptr, red, st));
  // ---- residual blocks, last to first ----
  for (int i = (int)p.blocks.size() - 1; i >= 0; --i) {
    const LmBlock& b = p.blocks[i];
    const int64_t M = (int64_t)B * b.Lin;
    const float* xin = i == 0 ? ws + lo.a0 : ws + lo.o[i - 1];
    // main branch: join backward (maxpool routing + ReLU6 mask) -> BN2 -> pointwise 2
    KWS_TRY(kws_block_out_bwd(dO, ws + lo.y2[i], c.bn_at(b.bn2_idx), G, part, B, b.Lin, b.nf, b.stride, 1, st));
    int np = (int)(kws_block_out_bwd_part_floats(B, b.Lin, b.nf, b.stride) / (5 * b.nf));
    KWS_TRY(kws_dw_bwd_finalize(part, np, M, b.nf, nullptr, grads + b.bn2.gamma, grads + b.bn2.beta, coef, red, st));
    KWS_TRY(kws_bn_bwd_apply(G, ws + lo.y2[i], c.bn_at(b.bn2_idx), params + b.bn2.gamma, coef, M, b.nf, st));
    KWS_TRY(kws_transpose_f32(params + b.pw2, ws + lo.WT, b.nf, b.nf, st));
    KWS_TRY(kws_gemm_nn_f32(G, ws + lo.WT, DZ, M, b.nf, b.nf, nullptr, st));
    KWS_TRY(kws_gemm_tn_f32(ws + lo.z2[i], G, grads + b.pw2, M, b.nf, b.nf, ws + lo.tn, st));
    // depthwise 2 -> BN1 -> pointwise 1
    // (two passes over dz and y1 instead of "store g, then kws_bn_bwd_apply": the masked gradient is never stored)
    KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.y1[i], c.bn_at(b.bn1_idx), params + b.dw2, nullptr, nullptr, part, 1, B,
                                  b.Lin, b.Lin, b.nf, 1, 1, st));
    np = (int)(kws_dwconv_bwd_part_floats(B, b.Lin, b.nf) / (5 * b.nf));
    KWS_TRY(kws_dw_bwd_finalize(part, np, M, b.nf, grads + b.dw2, grads + b.bn1.gamma, grads + b.bn1.beta, coef, red, st));
    KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.y1[i], c.bn_at(b.bn1_idx), params + b.dw2, coef, G, nullptr, 2, B, b.Lin,
                                  b.Lin, b.nf, 1, 1, st));
    KWS_TRY(kws_transpose_f32(params + b.pw1, ws + lo.WT, b.cin, b.nf, st));
    KWS_TRY(kws_gemm_nn_f32(G, ws + lo.WT, DZ, M, b.nf, b.cin, nullptr, st));
    KWS_TRY(kws_gemm_tn_f32(ws + lo.z1[i], G, grads + b.pw1, M, b.cin, b.nf, ws + lo.tn, st));
    // depthwise 1 on the (materialised) block input
    KWS_TRY(kws_dwconv_bwd_f32(DZ, xin, nullptr, params + b.dw1, dX, part, B, b.Lin, b.Lin, b.cin, 1, 1, st));
    np = (int)(kws_dwconv_bwd_part_floats(B, b.Lin, b.cin) / (5 * b.cin));
    KWS_TRY(kws_dw_bwd_finalize(part, np, M, b.cin, grads + b.dw1, nullptr, nullptr, nullptr, red, st));
    // residual branch
    if (!b.has_short) {
      KWS_TRY(kws_add_f32(dX, dO, dX, M * b.cin, st));
    } else {
      const int64_t Mo = (int64_t)B * b.Lout;
      KWS_TRY(kws_block_out_bwd(dO, ws + lo.ys[i], c.bn_at(b.bns_idx), dO, part, B, b.Lout, b.nf, 1, 0, st));
      np = (int)(kws_block_out_bwd_part_floats(B, b.Lout, b.nf, 1) / (5 * b.nf));
      KWS_TRY(kws_dw_bwd_finalize(part, np, Mo, b.nf, nullptr, grads + b.bns.gamma, grads + b.bns.beta, coef, red, st));
      KWS_TRY(kws_bn_bwd_apply(dO, ws + lo.ys[i], c.bn_at(b.bns_idx), params + b.bns.gamma, coef, Mo, b.nf, st));
      KWS_TRY(kws_gemm_tn_gather_f32(xin, &b.gs, dO, grads + b.ws, B, b.nf, ws + lo.tn, st));
      KWS_TRY(kws_transpose_f32(params + b.ws, ws + lo.WT, b.cin, b.nf, st));
      KWS_TRY(kws_gemm_nn_f32(dO, ws + lo.WT, ws + lo.DXS, Mo, b.nf, b.cin, nullptr, st));
      KWS_TRY(kws_add_strided_f32(dX, ws + lo.DXS, B, b.Lin, b.Lout, b.cin, b.stride, st));
    }
    std::swap(dO, dX);
  }
  // ---- first convolution ----
  {
    const int64_t M = (int64_t)B * p.L0;
    KWS_TRY(kws_block_out_bwd(dO, ws + lo.y0, c.bn_at(1), G, part, B, p.L0, p.C0, 1, 1, st));
    const int np = (int)(kws_block_out_bwd_part_floats(B, p.L0, p.C0, 1) / (5 * p.C0));
    KWS_TRY(kws_dw_bwd_finalize(part, np, M, p.C0, nullptr, grads + p.bn0.gamma, grads + p.bn0.beta, coef, red, st));
    KWS_TRY(kws_bn_bwd_apply(G, ws + lo.y0, c.bn_at(1), params + p.bn0.gamma, coef, M, p.C0, st));
    if (p.Fp == p.F) {
      KWS_TRY(kws_gemm_tn_gather_f32(x, &p.g0, G, grads + p.conv1, B, p.C0, ws + lo.tn, st));
    } else {  // the forward of this step left the padded input in xpad
      float* gw = ws + lo.gwpad;
      KWS_TRY(kws_gemm_tn_gather_f32(ws + lo.xpad, &p.g0, G, gw, B, p.C0, ws + lo.tn, st));
      KWS_HIP(hipMemcpy2DAsync(grads + p.conv1, (size_t)p.F * p.C0 * 4, gw, (size_t)p.Fp * p.C0 * 4, (size_t)p.F * p.C0 * 4, 3,
                               hipMemcpyDeviceToDevice, st));
    }
  }
  return KWS_OK;
}
