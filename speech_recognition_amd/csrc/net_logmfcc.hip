// Residual-block network programs.  One layer table (LmProgram) drives five reference models:
//   style 0  conv_1d_log_mfcc_model (model.py:1400-1479) and conv_1d_spectrogram_model
//            (model.py:1482-1561: the same program on 257-bin input, first convolution on re-pitched copies)
//   style 1  steffeNet (model.py:1663-1726): stride in the block's first depthwise convolution, context block,
//            global max ++ average pooling tail
//   style 2  conv_1d_residual_model (model.py:841-908): 3-wide SAME max-pool joins, plain blocks after the stack,
//            global-average tail
//   style 3  conv_1d_mfcc_and_raw_model (model.py:1563-1660): two stems on packed [mfcc | raw] rows, concatenated
// The original description of style 0 follows.
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
constexpr float STEFFE_DROP_KEEP = 0.5f;     // Dropout(0.5), model.py:1716
constexpr float STEFFE_LABEL_SMOOTH = 0.1f;  // model.py:1722-1724

struct LmBlock {
  int nf, stride, cin, Lin, Lout;
  // where the stride sits: log-mfcc blocks pool after the second pointwise (s1 = 1, pool = stride, Lmid = Lin);
  // steffeNet blocks stride their FIRST depthwise convolution (s1 = stride, pool = 1, Lmid = Lout)
  int s1, pool, Lmid, pad1;
  int pool3 = 0, ppad = 0;  // conv_1d_residual: MaxPool1D(3, stride, 'same') join, window of output t starts at t*stride - ppad
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

// a depthwise block without a residual (conv_1d_residual's _reduce_block: strided SAME, then VALID)
struct LmPlain {
  int64_t dw, pw;
  BnRef bn;
  int bn_idx, cin, cout, stride, pad_l, Lin, Lout;
};

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
  int maxC;
  // steffeNet (style 1): raw input, first convolution of K0 = 75 taps on one channel (gathered as ONE tap of
  // K0p = 76 samples against a zero-padded kernel), a context block between the first convolution and the
  // residual stack, global max ++ average pooling tail
  int style = 0, K0 = 0, K0p = 0;
  // conv_1d_residual (style 2): raw input through the time-slice gather, 3-wide max-pool joins, plain blocks after the
  // residual stack, global average pooling tail with bias and plain CE
  std::vector<LmPlain> plain;
  // conv_1d_mfcc_and_raw (style 3): input rows are [mfcc T0*F | raw L_raw]; two first convolutions (Cm + Cr = C0
  // channels, concatenated after BN + ReLU6), style-2 blocks, global-average tail
  int Cm = 0, Cr = 0, Din = 0;
  int64_t conv1r = 0;
  BnRef bn0r;
  kws_gather_t g0r;
  float drop_keep = 0.5f;
  int64_t ctx_dw = 0, ctx_pw = 0;
  BnRef ctx_bn;
  int ctx_bn_idx = 0;
};

namespace {

struct LmLayout {
  int64_t total = 0;
  int64_t y0 = 0, a0 = 0;
  std::vector<int64_t> ys, z1, y1, z2, y2, o;
  int64_t bn = 0, bn_stride = 0, part = 0, red = 0, coef = 0, WT = 0, tn = 0, swg = 0;
  int64_t tnq = 0, tnq_floats = 0;         // slabs of the pointwise weight gradients of one backward pass (KwsSlabQueue)
  int64_t dwq = 0, dwq_floats = 0;         // partial rows of the depthwise weight gradients folded by one launch (DwFinQueue)
  int64_t dOa = 0, dOb = 0, G = 0, DZ = 0, DXS = 0;
  int64_t u = 0, fd = 0, dl = 0, gu = 0, coef2 = 0, per_loss = 0, per_correct = 0, att = 0;
  int64_t xpad = 0, wpad = 0, gwpad = 0;  // only when Fp != F (style 0) / always (style 1: padded first kernel)
  int64_t zc = 0, yc = 0, ac = 0;          // steffeNet context block
  std::vector<int64_t> pz, py;             // plain blocks (style 2)
  int64_t alast = 0;
  int64_t a0m = 0, a0r = 0;                // style 3: activated branch outputs before the concatenation
  std::vector<int64_t> wt_pw1, wt_pw2, wt_ws, wt_plain;   // transposed pointwise kernels for the dgrad GEMMs
  int64_t wt_ctx = 0;
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
  int64_t sum_tnq = 0;                     // the pointwise weight gradients keep their slabs until one batched sum
  int64_t sum_dwq = 0;                     // depthwise backward kernels whose rows hold only a weight gradient: folded at the end
  auto upd_dwq = [&](int L, int C) { sum_dwq += (kws_dwconv_bwd_part_floats(B, L, C) + 63) / 64 * 64; };
  auto upd_pw = [&](int64_t M, int K, int N) {
    upd_gemm(M, K, N);
    sum_tnq += (kws_gemm_tn_workspace_floats(M, K, N) + 63) / 64 * 64;
  };
  upd_pw((int64_t)B * p.L0, p.style == 1 ? p.K0p : 3 * p.Fp, p.C0);     // (gathered weight gradients queue their slabs too: round 5)
  if (p.style == 3) {
    upd_pw((int64_t)B * p.L0, p.g0r.taps * p.g0r.cin, p.Cr);
    lo->a0m = bp.take((int64_t)B * p.L0 * p.Cm);
    lo->a0r = bp.take((int64_t)B * p.L0 * p.Cr);
  }
  max_part = std::max(max_part, kws_block_out_bwd_part_floats(B, p.L0, p.C0, 1));
  if (p.style == 1) {
    lo->zc = bp.take((int64_t)B * p.L0 * p.C0);
    lo->yc = bp.take((int64_t)B * p.L0 * p.C0);
    lo->ac = bp.take((int64_t)B * p.L0 * p.C0);
    max_z = (int64_t)B * p.L0 * p.C0;
    upd_pw((int64_t)B * p.L0, p.C0, p.C0);
    max_part = std::max(max_part, kws_dwconv_bwd_part_floats(B, p.L0, p.C0));
    upd_dwq(p.L0, p.C0);
  }
  for (int i = 0; i < nb; ++i) {
    const LmBlock& b = p.blocks[i];
    if (b.has_short) lo->ys[i] = bp.take((int64_t)B * b.Lout * b.nf);
    lo->z1[i] = bp.take((int64_t)B * b.Lmid * b.cin);
    lo->y1[i] = bp.take((int64_t)B * b.Lmid * b.nf);
    lo->z2[i] = bp.take((int64_t)B * b.Lmid * b.nf);
    lo->y2[i] = bp.take((int64_t)B * b.Lmid * b.nf);
    lo->o[i] = bp.take((int64_t)B * b.Lout * b.nf);
    max_o = std::max(max_o, std::max((int64_t)B * b.Lout * b.nf, (int64_t)B * b.Lin * b.cin));
    max_y = std::max(max_y, (int64_t)B * b.Lmid * b.nf);
    max_z = std::max(max_z, std::max((int64_t)B * b.Lmid * b.cin, (int64_t)B * b.Lmid * b.nf));
    max_xs = std::max(max_xs, (int64_t)B * b.Lout * b.cin);
    upd_pw((int64_t)B * b.Lmid, b.cin, b.nf);
    upd_pw((int64_t)B * b.Lmid, b.nf, b.nf);
    if (b.has_short) upd_pw((int64_t)B * b.Lout, b.cin, b.nf);
    max_part = std::max(max_part, kws_dwconv_bwd_part_floats(B, b.Lin, b.cin));
    upd_dwq(b.Lin, b.cin);
    max_part = std::max(max_part, kws_dwconv_bwd_part_floats(B, b.Lmid, b.nf));
    max_part = std::max(max_part, b.pool3 ? kws_block_out3_bwd_part_floats(B, b.Lmid, b.nf)
                                          : kws_block_out_bwd_part_floats(B, b.Lmid, b.nf, b.pool));
    max_part = std::max(max_part, kws_block_out_bwd_part_floats(B, b.Lout, b.nf, 1));
  }
  lo->pz.assign(p.plain.size(), 0); lo->py.assign(p.plain.size(), 0);
  for (size_t j = 0; j < p.plain.size(); ++j) {
    const LmPlain& q = p.plain[j];
    lo->pz[j] = bp.take((int64_t)B * q.Lout * q.cin);
    lo->py[j] = bp.take((int64_t)B * q.Lout * q.cout);
    max_y = std::max(max_y, (int64_t)B * q.Lout * q.cout);
    max_z = std::max(max_z, (int64_t)B * q.Lout * q.cin);
    max_o = std::max(max_o, std::max((int64_t)B * q.Lin * q.cin, (int64_t)B * q.Lout * q.cout));
    upd_pw((int64_t)B * q.Lout, q.cin, q.cout);
    max_part = std::max(max_part, kws_dwconv_bwd_part_floats(B, q.Lin, q.cin));
    if (j == 0) upd_dwq(q.Lin, q.cin);
    max_part = std::max(max_part, kws_block_out_bwd_part_floats(B, q.Lout, q.cout, 1));
  }
  if (!p.plain.empty()) lo->alast = bp.take((int64_t)B * p.T * p.C);
  max_part = std::max(max_part, (int64_t)B * 5 * p.C);
  const int feat = p.style == 1 ? 2 * p.C : p.C;   // width of the dense layer's input
  lo->bn_stride = 4 * p.maxC;
  lo->bn = bp.take(lo->bn_stride * (p.n_bn + 1));
  lo->part = bp.take(max_part);
  lo->red = bp.take((int64_t)KWS_REDUCE_SLICES * 5 * p.maxC);
  lo->coef = bp.take(2 * p.maxC);
  lo->WT = bp.take(max_wt);
  lo->wt_pw1.assign(nb, 0); lo->wt_pw2.assign(nb, 0); lo->wt_ws.assign(nb, 0); lo->wt_plain.assign(p.plain.size(), 0);
  for (int i = 0; i < nb; ++i) {
    const LmBlock& b = p.blocks[i];
    lo->wt_pw1[i] = bp.take((int64_t)b.cin * b.nf);
    lo->wt_pw2[i] = bp.take((int64_t)b.nf * b.nf);
    if (b.has_short) lo->wt_ws[i] = bp.take((int64_t)b.cin * b.nf);
  }
  for (size_t j = 0; j < p.plain.size(); ++j) lo->wt_plain[j] = bp.take((int64_t)p.plain[j].cin * p.plain[j].cout);
  if (p.style == 1) lo->wt_ctx = bp.take((int64_t)p.C0 * p.C0);
  lo->tn = bp.take(max_tn);
  lo->tnq_floats = sum_tnq;
  lo->tnq = bp.take(sum_tnq);
  lo->dwq_floats = sum_dwq;
  lo->dwq = bp.take(sum_dwq);
  lo->swg = bp.take((int64_t)KWS_SMALL_WGRAD_SLICES * feat * p.NC);
  lo->dOa = bp.take(max_o);
  lo->dOb = bp.take(max_o);
  lo->G = bp.take(max_y);
  lo->DZ = bp.take(max_z);
  lo->DXS = bp.take(max_xs);
  lo->u = bp.take((int64_t)B * 16);
  lo->fd = bp.take((int64_t)B * feat);
  lo->dl = bp.take((int64_t)B * p.NC);
  lo->gu = bp.take((int64_t)B * 16);
  lo->coef2 = bp.take(4);
  lo->per_loss = bp.take(B);
  lo->per_correct = bp.take(B);
  lo->att = bp.take((int64_t)B * 16);
  if (p.style == 1) {
    lo->wpad = bp.take((int64_t)p.K0p * p.C0);
    lo->gwpad = bp.take((int64_t)p.K0p * p.C0);
  } else if (p.Fp != p.F) {
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

// out[r, 0:cols] = in[r, 0:cols] for row pitches ld_in / ld_out (channel concatenation and its backward split)
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                        int ld_out, int64_t rows, int cols4) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < rows * cols4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    *reinterpret_cast<float4*>(out + r * ld_out + c) = *reinterpret_cast<const float4*>(in + r * ld_in + c);
  }
}
int copy_cols(const float* in, int ld_in, float* out, int ld_out, int64_t rows, int cols, hipStream_t st) {
  KWS_REQUIRE(cols % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0, "copy_cols: widths must be multiples of 4");
  const int64_t n4 = rows * (cols / 4);
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n4, 256), 8192)), dim3(256), 0, st, in,
                     ld_in, out, ld_out, rows, cols / 4);
  KWS_LAUNCH_CHECK("copy_cols_kernel");
  return KWS_OK;
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
  if (p.style == 1) {  // [75, 1, C0] kernel + one zero row: the gather reads 76 samples per output row
    float* wp = c.ws + c.lo.wpad;
    KWS_HIP(hipMemcpyAsync(wp, c.params + p.conv1, (size_t)p.K0 * p.C0 * 4, hipMemcpyDeviceToDevice, c.st));
    KWS_HIP(hipMemsetAsync(wp + (size_t)p.K0 * p.C0, 0, (size_t)(p.K0p - p.K0) * p.C0 * 4, c.st));
    *w_used = wp;
    return KWS_OK;
  }
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
  bool z1_first = false;      // the first block's first depthwise output already written (see below)
  if (p.style == 3) {  // two stems on the packed [mfcc | raw] rows, concatenated after BN + ReLU6
    const int64_t M = (int64_t)B * p.L0;
    float* y0m = ws + lo.y0;
    float* y0r = ws + lo.y0 + M * p.Cm;
    KWS_TRY(kws_gemm_gather_f32(x, &p.g0, c.params + p.conv1, y0m, B, p.Cm, stats, c.st));
    KWS_TRY(bn_table(c, p.bn0, 1, M, kws_gemm_gather_stats_rows(M)));
    KWS_TRY(kws_gemm_gather_f32(x, &p.g0r, c.params + p.conv1r, y0r, B, p.Cr, stats, c.st));
    KWS_TRY(bn_table(c, p.bn0r, 2, M, kws_gemm_gather_stats_rows(M)));
    KWS_TRY(kws_bn_relu6_apply(y0m, c.bn_at(1), ws + lo.a0m, M, p.Cm, 1, c.st));
    KWS_TRY(kws_bn_relu6_apply(y0r, c.bn_at(2), ws + lo.a0r, M, p.Cr, 1, c.st));
    KWS_TRY(copy_cols(ws + lo.a0m, p.Cm, ws + lo.a0, p.C0, M, p.Cm, c.st));
    KWS_TRY(copy_cols(ws + lo.a0r, p.Cr, ws + lo.a0 + p.Cm, p.C0, M, p.Cr, c.st));
  } else {
  const float *x0, *w0;
  KWS_TRY(pad_first_conv(c, x, &x0, &w0));
  KWS_TRY(kws_gemm_gather_f32(x0, &p.g0, w0, ws + lo.y0, B, p.C0, stats, c.st));
  KWS_TRY(bn_table(c, p.bn0, 1, (int64_t)B * p.L0, kws_gemm_gather_stats_rows((int64_t)B * p.L0)));
  // the first convolution's activation - and, where the first block starts with a k 3 / stride 1 / 'same' depthwise convolution over it
  // (conv_1d_log_mfcc), that convolution's output in the same pass (round 6, kws_block_out_dw_fwd without a residual: bit-identical)
  if (p.style == 0 && !p.blocks.empty() && kws_net_get_gemm_mode(c.n) != 1) {
    const LmBlock& b0 = p.blocks[0];
    if (b0.s1 == 1 && b0.pad1 == 1 && b0.Lmid == b0.Lin && b0.Lin == p.L0 && b0.cin == p.C0) {
      KWS_TRY(kws_block_out_dw_fwd(ws + lo.y0, c.bn_at(1), nullptr, nullptr, c.params + b0.dw1, ws + lo.a0, ws + lo.z1[0], B, p.L0, p.C0, 1, c.st));
      z1_first = true;
    }
  }
  if (!z1_first) KWS_TRY(kws_bn_relu6_apply(ws + lo.y0, c.bn_at(1), ws + lo.a0, (int64_t)B * p.L0, p.C0, 1, c.st));
  }
  const float* xin = ws + lo.a0;
  if (p.style == 1) {  // _context_conv(x, 256, 3, 'same'): depthwise -> pointwise -> BN -> ReLU6, materialised
    const int64_t M = (int64_t)B * p.L0;
    KWS_TRY(kws_dwconv_fwd_f32(xin, nullptr, c.params + p.ctx_dw, ws + lo.zc, B, p.L0, p.L0, p.C0, 1, 1, c.st));
    KWS_TRY(kws_gemm_nn_f32(ws + lo.zc, c.params + p.ctx_pw, ws + lo.yc, M, p.C0, p.C0, stats, c.st));
    KWS_TRY(bn_table(c, p.ctx_bn, p.ctx_bn_idx, M, kws_gemm_nn_stats_rows(M, p.C0, p.C0)));
    KWS_TRY(kws_bn_relu6_apply(ws + lo.yc, c.bn_at(p.ctx_bn_idx), ws + lo.ac, M, p.C0, 1, c.st));
    xin = ws + lo.ac;
  }
  bool z1_ready = z1_first;
  const bool fuse_join_dw = kws_net_get_gemm_mode(c.n) != 1;
  for (size_t i = 0; i < p.blocks.size(); ++i) {
    const LmBlock& b = p.blocks[i];
    const int64_t M = (int64_t)B * b.Lmid;
    if (b.has_short) {
      // the shortcut convolution (1 x 1, stride s): over an even-length input it is every s-th ROW of the block input - a plain GEMM
      // with a row pitch on the wave-specialised kernel (round 6); the gathered kernel otherwise
      const int64_t Ms = (int64_t)B * b.Lout;
      int lda = 0, srows = kws_gemm_gather_stats_rows(Ms);
      int rs = 1;
      if (kws_gather_strided_rows(&b.gs, &lda)) {
        rs = kws_gemm_nn_strided_f32(xin, lda, c.params + b.ws, ws + lo.ys[i], Ms, b.gs.cin, b.nf, stats, c.st);
        if (rs < 0) return rs;
        if (rs == 0 && stats) srows = kws_gemm_nn_stats_rows(Ms, b.gs.cin, b.nf);
      }
      if (rs != 0) KWS_TRY(kws_gemm_gather_f32(xin, &b.gs, c.params + b.ws, ws + lo.ys[i], B, b.nf, stats, c.st));
      KWS_TRY(bn_table(c, b.bns, b.bns_idx, Ms, srows));
    }
    if (!z1_ready)      // (else: written by the previous block's join, kws_block_out_dw_fwd below)
      KWS_TRY(kws_dwconv_fwd_f32(xin, nullptr, c.params + b.dw1, ws + lo.z1[i], B, b.Lin, b.Lmid, b.cin, b.s1, b.pad1, c.st));
    z1_ready = false;
    KWS_TRY(kws_gemm_nn_f32(ws + lo.z1[i], c.params + b.pw1, ws + lo.y1[i], M, b.cin, b.nf, stats, c.st));
    KWS_TRY(bn_table(c, b.bn1, b.bn1_idx, M, kws_gemm_nn_stats_rows(M, b.cin, b.nf)));
    KWS_TRY(kws_dwconv_fwd_f32(ws + lo.y1[i], c.bn_at(b.bn1_idx), c.params + b.dw2, ws + lo.z2[i], B, b.Lmid, b.Lmid,
                               b.nf, 1, 1, c.st));
    KWS_TRY(kws_gemm_nn_f32(ws + lo.z2[i], c.params + b.pw2, ws + lo.y2[i], M, b.nf, b.nf, stats, c.st));
    KWS_TRY(bn_table(c, b.bn2, b.bn2_idx, M, kws_gemm_nn_stats_rows(M, b.nf, b.nf)));
    if (b.pool3)
      KWS_TRY(kws_block_out3_fwd(ws + lo.y2[i], c.bn_at(b.bn2_idx), b.has_short ? ws + lo.ys[i] : xin,
                                 b.has_short ? c.bn_at(b.bns_idx) : nullptr, ws + lo.o[i], B, b.Lmid, b.Lout, b.nf, b.stride,
                                 b.ppad, c.st));
    else {
      // round 6: the join writes the NEXT block's first depthwise output too where that is a k 3 / stride 1 / 'same' convolution over
      // this block's output (every log-mfcc block): one tensor pass and one launch less per block, bit-identical (gemm mode 1, the A/B
      // reference schedule, keeps the two launches)
      const LmBlock* nx = i + 1 < p.blocks.size() ? &p.blocks[i + 1] : nullptr;
      const bool fuse = nx && fuse_join_dw && nx->s1 == 1 && nx->pad1 == 1 && nx->Lmid == nx->Lin && nx->Lin == b.Lout && nx->cin == b.nf;
      if (fuse) {
        KWS_TRY(kws_block_out_dw_fwd(ws + lo.y2[i], c.bn_at(b.bn2_idx), b.has_short ? ws + lo.ys[i] : xin,
                                     b.has_short ? c.bn_at(b.bns_idx) : nullptr, c.params + nx->dw1, ws + lo.o[i], ws + lo.z1[i + 1], B,
                                     b.Lmid, b.nf, b.pool, c.st));
        z1_ready = true;
      } else {
        KWS_TRY(kws_block_out_fwd(ws + lo.y2[i], c.bn_at(b.bn2_idx), b.has_short ? ws + lo.ys[i] : xin,
                                  b.has_short ? c.bn_at(b.bns_idx) : nullptr, ws + lo.o[i], B, b.Lmid, b.nf, b.pool, c.st));
      }
    }
    xin = ws + lo.o[i];
  }
  if (!p.plain.empty()) {  // _reduce_block: depthwise -> pointwise -> BN -> ReLU6, twice; the last activation is materialised
    const float* pbn = nullptr;
    for (size_t j = 0; j < p.plain.size(); ++j) {
      const LmPlain& q = p.plain[j];
      const int64_t M = (int64_t)B * q.Lout;
      KWS_TRY(kws_dwconv_fwd_f32(xin, pbn, c.params + q.dw, ws + lo.pz[j], B, q.Lin, q.Lout, q.cin, q.stride, q.pad_l, c.st));
      KWS_TRY(kws_gemm_nn_f32(ws + lo.pz[j], c.params + q.pw, ws + lo.py[j], M, q.cin, q.cout, stats, c.st));
      KWS_TRY(bn_table(c, q.bn, q.bn_idx, M, kws_gemm_nn_stats_rows(M, q.cin, q.cout)));
      xin = ws + lo.py[j];
      pbn = c.bn_at(q.bn_idx);
    }
    KWS_TRY(kws_bn_relu6_apply(xin, pbn, ws + lo.alast, (int64_t)B * p.T, p.C, 1, c.st));
    xin = ws + lo.alast;
  }
  memset(t, 0, sizeof(*t));
  t->x = xin;
  if (p.style != 0) return KWS_OK;  // the caller sets up the global-pooling tail
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
  // any length >= 3: the strided blocks are Keras 'same' layers - MaxPool1D(2, 2) and Conv1D(nf, 1, strides=2) both give
  // ceil(L / 2), e.g. the function's own default spectrogram_length = 65 -> 63 -> 32 -> 16 -> 8 (model.py:1410)
  KWS_REQUIRE(c.spectrogram_length >= 3, "net: spectrogram_length %d is too short", c.spectrogram_length);
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
    b.nf = spec[i][0]; b.stride = spec[i][1]; b.cin = cin; b.Lin = L; b.Lout = (L + b.stride - 1) / b.stride;
    b.has_short = b.stride != 1;
    b.s1 = 1; b.pool = b.stride; b.Lmid = b.Lin; b.pad1 = 1;
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
  p->maxC = 256;
  return KWS_OK;
}

// steffeNet, reference model.py:1663-1726
int steffe_build(kws_net* n) {
  const kws_net_config_t& c = n->cfg;
  KWS_REQUIRE(c.num_classes >= 2 && c.num_classes <= 64, "net: num_classes %d out of range", c.num_classes);
  KWS_REQUIRE(c.input_size >= 3200 && c.input_size % 2 == 0, "net: steffeNet input_size %d", c.input_size);
  LmProgram* p = new LmProgram();
  n->lm = p;
  p->style = 1;
  p->drop_keep = STEFFE_DROP_KEEP;
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
  auto same = [](int L, int k, int stride, int* Lout, int* pad_l) {   // TF 'SAME'
    *Lout = (L + stride - 1) / stride;
    const int pad = std::max((*Lout - 1) * stride + k - L, 0);
    *pad_l = pad / 2;
  };
  // Conv1D(256, 75, strides=50, padding='same', use_bias=False): no kernel_regularizer (model.py:1705)
  p->K0 = 75; p->K0p = 76; p->C0 = 256; p->T0 = c.input_size; p->F = 1; p->Fp = 1;
  int pl0;
  same(c.input_size, p->K0, 50, &p->L0, &pl0);
  KWS_REQUIRE(pl0 % 2 == 0, "net: steffeNet left padding %d must be even (8-byte gather loads)", pl0);
  p->conv1 = conv(p->K0, 1, p->C0, false);
  int idx0;
  p->bn0 = bn(p->C0, &idx0);
  kws_gather_t g0;
  g0.L_out = p->L0; g0.cin = p->K0p; g0.taps = 1; g0.stride_t = 50; g0.stride_j = 0; g0.base_off = -pl0;
  g0.x_len = c.input_size; g0.x_batch_stride = c.input_size;
  p->g0 = g0;
  p->ctx_dw = dw(p->C0);                                  // _context_conv(x, 256, 3, padding='same'), model.py:1708
  p->ctx_pw = conv(1, p->C0, p->C0, true);
  p->ctx_bn = bn(p->C0, &p->ctx_bn_idx);
  static const int widths[6] = {320, 384, 512, 768, 1024, 1536};  // model.py:1709
  int cin = p->C0, L = p->L0;
  p->maxC = p->C0;
  for (int wi = 0; wi < 6; ++wi) {
    for (int stride = 2; stride >= 1; --stride) {
      LmBlock b;
      b.nf = widths[wi]; b.stride = stride; b.cin = cin; b.Lin = L;
      same(L, 3, stride, &b.Lout, &b.pad1);
      b.s1 = stride; b.pool = 1; b.Lmid = b.Lout;
      b.has_short = stride != 1;
      b.ws = 0; b.bns_idx = 0;
      memset(&b.gs, 0, sizeof(b.gs));
      if (b.has_short) {
        b.ws = conv(1, cin, b.nf, false);                 // Conv1D(nh, 1, strides, 'same', no bias), model.py:1692-1693
        b.bns = bn(b.nf, &b.bns_idx);
        b.gs.L_out = b.Lout; b.gs.cin = cin; b.gs.taps = 1; b.gs.stride_t = stride * cin; b.gs.stride_j = 0;
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
      p->maxC = std::max(p->maxC, b.nf);
    }
  }
  p->T = L; p->C = cin; p->NC = c.num_classes;
  KWS_REQUIRE(p->T >= 1 && p->T <= 64, "net: %d time steps at the tail", p->T);
  p->dk = kws_net_add_tensor(n, "dense_1/kernel", {2 * cin, p->NC}, false, KWS_L2_COEF, 2 * cin, p->NC, 0.f);
  p->n_bn = n_bn;
  return KWS_OK;
}

// conv_1d_residual_model, reference model.py:841-908
int residual_build(kws_net* n) {
  const kws_net_config_t& c = n->cfg;
  KWS_REQUIRE(c.num_classes >= 2 && c.num_classes <= 64, "net: num_classes %d out of range", c.num_classes);
  KWS_REQUIRE(c.input_size >= 4000 && c.input_size % 2 == 0, "net: conv_1d_residual input_size %d", c.input_size);
  const int fm = c.filter_mult > 0 ? c.filter_mult : 1;
  LmProgram* p = new LmProgram();
  n->lm = p;
  p->style = 2;
  p->drop_keep = 0.5f;                           // Dropout(0.5), model.py:899
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
  auto same = [](int L, int k, int stride, int* Lout, int* pad_l) {   // TF 'SAME'
    *Lout = (L + stride - 1) / stride;
    const int pad = std::max((*Lout - 1) * stride + k - L, 0);
    *pad_l = pad / 2;
  };
  // overlapping_time_slice_stack(x, 40, 20) SAME fused with Conv1D(64, 3, strides=2) (model.py:881-884), as the
  // raw-waveform net's first convolution
  int Lf, plf;
  same(c.input_size, 40, 20, &Lf, &plf);
  p->T0 = c.input_size; p->F = 40; p->Fp = 40; p->C0 = 64 * fm; p->L0 = (Lf - 3) / 2 + 1;
  p->conv1 = conv(3, 40, p->C0, true);
  int idx0;
  p->bn0 = bn(p->C0, &idx0);
  kws_gather_t g0;
  g0.L_out = p->L0; g0.cin = 40; g0.taps = 3; g0.stride_t = 2 * 20; g0.stride_j = 20; g0.base_off = -plf;
  g0.x_len = c.input_size; g0.x_batch_stride = c.input_size;
  p->g0 = g0;
  static const int spec[13][2] = {{128, 2}, {256, 2}, {256, 1}, {256, 1}, {256, 1}, {256, 1}, {256, 1}, {256, 1},
                                  {256, 1}, {256, 1}, {512, 2}, {728, 2}, {728, 2}};  // model.py:888-894
  int cin = p->C0, L = p->L0;
  p->maxC = p->C0;
  for (int i = 0; i < 13; ++i) {
    LmBlock b;
    b.nf = spec[i][0] * fm; b.stride = spec[i][1]; b.cin = cin; b.Lin = L;
    b.s1 = 1; b.pool = b.stride; b.Lmid = L; b.pad1 = 1; b.pool3 = 1;
    same(L, 3, b.stride, &b.Lout, &b.ppad);
    b.has_short = b.stride != 1;
    b.ws = 0; b.bns_idx = 0;
    memset(&b.gs, 0, sizeof(b.gs));
    if (b.has_short) {
      b.ws = conv(1, cin, b.nf, false);
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
    p->maxC = std::max(p->maxC, b.nf);
  }
  // _reduce_block(x, 1024, 3): _reduce_conv (strides 2, 'same') then _context_conv ('valid'), model.py:895
  const int cr = 1024 * fm;
  for (int j = 0; j < 2; ++j) {
    LmPlain q;
    q.cin = cin; q.cout = cr; q.Lin = L;
    if (j == 0) { q.stride = 2; same(L, 3, 2, &q.Lout, &q.pad_l); }
    else { q.stride = 1; q.pad_l = 0; q.Lout = L - 2; }
    KWS_REQUIRE(q.Lout >= 1, "net: conv_1d_residual input too short");
    q.dw = dw(cin);
    q.pw = conv(1, cin, cr, true);
    q.bn = bn(cr, &q.bn_idx);
    p->plain.push_back(q);
    cin = cr;
    L = q.Lout;
    p->maxC = std::max(p->maxC, cr);
  }
  p->T = L; p->C = cin; p->NC = c.num_classes;
  KWS_REQUIRE(p->T <= 64, "net: %d time steps at the tail", p->T);
  p->dk = kws_net_add_tensor(n, "dense_1/kernel", {cin, p->NC}, false, KWS_L2_COEF, cin, p->NC, 0.f);
  p->db = kws_net_add_tensor(n, "dense_1/bias", {p->NC}, false, 0.f, 0, 0, 0.f);
  p->n_bn = n_bn;
  return KWS_OK;
}

// conv_1d_mfcc_and_raw_model, reference model.py:1563-1660.  Input rows: [mfcc T*F | raw L].
int mfcc_raw_build(kws_net* n) {
  const kws_net_config_t& c = n->cfg;
  const int T = c.spectrogram_length, F = c.num_features;
  KWS_REQUIRE(c.num_classes >= 2 && c.num_classes <= 64, "net: num_classes %d out of range", c.num_classes);
  KWS_REQUIRE(T >= 19 && F >= 4 && F % 4 == 0 && c.input_size > T * F, "net: mfcc_and_raw input %d, features %d x %d",
              c.input_size, T, F);
  const int Lraw = c.input_size - T * F;
  const int frame_len = 480, frame_step = 160;   // window_size_samples / window_stride_samples of prepare_model_settings
  KWS_REQUIRE(1 + (Lraw - frame_len) / frame_step == T && (T * F) % 4 == 0 && Lraw % 2 == 0,
              "net: mfcc_and_raw needs 1 + (raw %d - 480) / 160 == spectrogram_length %d", Lraw, T);
  LmProgram* p = new LmProgram();
  n->lm = p;
  p->style = 3;
  p->drop_keep = 0.7f;                           // Dropout(0.3), model.py:1648
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
  auto same = [](int L, int k, int stride, int* Lout, int* pad_l) {
    *Lout = (L + stride - 1) / stride;
    const int pad = std::max((*Lout - 1) * stride + k - L, 0);
    *pad_l = pad / 2;
  };
  p->T0 = T; p->F = F; p->Fp = F; p->Cm = 64; p->Cr = 96; p->C0 = p->Cm + p->Cr; p->L0 = T - 2; p->Din = c.input_size;
  int idx;
  p->conv1 = conv(3, F, p->Cm, true);            // model.py:1615
  p->bn0 = bn(p->Cm, &idx);
  p->conv1r = conv(3, frame_len, p->Cr, true);   // model.py:1625
  p->bn0r = bn(p->Cr, &idx);
  kws_gather_t g;
  g.L_out = p->L0; g.cin = F; g.taps = 3; g.stride_t = F; g.stride_j = F; g.base_off = 0;
  g.x_len = T * F; g.x_batch_stride = p->Din;
  p->g0 = g;
  // overlapping_time_slice_stack(x, 480, 160, 'VALID') fused with Conv1D(96, 3): row t reads 3 frames 160 apart
  g.cin = frame_len; g.stride_t = frame_step; g.stride_j = frame_step; g.base_off = T * F; g.x_len = p->Din;
  p->g0r = g;
  static const int spec[10][2] = {{160, 1}, {160, 1}, {192, 2}, {192, 1}, {256, 2}, {256, 1}, {320, 2}, {320, 1},
                                  {384, 2}, {384, 1}};  // model.py:1632-1641
  int cin = p->C0, L = p->L0;
  p->maxC = p->C0;
  for (int i = 0; i < 10; ++i) {
    LmBlock b;
    b.nf = spec[i][0]; b.stride = spec[i][1]; b.cin = cin; b.Lin = L;
    b.s1 = 1; b.pool = b.stride; b.Lmid = L; b.pad1 = 1; b.pool3 = 1;
    same(L, 3, b.stride, &b.Lout, &b.ppad);
    b.has_short = b.stride != 1;
    b.ws = 0; b.bns_idx = 0;
    memset(&b.gs, 0, sizeof(b.gs));
    if (b.has_short) {
      b.ws = conv(1, cin, b.nf, false);
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
    p->maxC = std::max(p->maxC, b.nf);
  }
  p->T = L; p->C = cin; p->NC = c.num_classes;
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
    *count = 4 * p.maxC;
    return KWS_OK;
  }
  if (what == 3) { *offset_floats = lo.att; *count = (int64_t)B * p.T; return KWS_OK; }
  if (what == 4) { *offset_floats = lo.u; *count = (int64_t)B * p.T; return KWS_OK; }
  if (what == 5) { *offset_floats = lo.o[nb - 1]; *count = (int64_t)B * p.T * p.C; return KWS_OK; }
  if (what == 0) {
    if (p.style == 3 && index <= 2) {
      *offset_floats = lo.y0 + (index == 2 ? (int64_t)B * p.L0 * p.Cm : 0);
      *count = (int64_t)B * p.L0 * (index == 2 ? p.Cr : p.Cm);
      return KWS_OK;
    }
    if (index == 1) { *offset_floats = lo.y0; *count = (int64_t)B * p.L0 * p.C0; return KWS_OK; }
    for (size_t j = 0; j < p.plain.size(); ++j)
      if (index == p.plain[j].bn_idx) { *offset_floats = lo.py[j]; *count = (int64_t)B * p.plain[j].Lout * p.plain[j].cout; return KWS_OK; }
    if (p.style == 1 && index == p.ctx_bn_idx) { *offset_floats = lo.yc; *count = (int64_t)B * p.L0 * p.C0; return KWS_OK; }
    for (int i = 0; i < nb; ++i) {
      const LmBlock& b = p.blocks[i];
      if (b.has_short && index == b.bns_idx) { *offset_floats = lo.ys[i]; *count = (int64_t)B * b.Lout * b.nf; return KWS_OK; }
      if (index == b.bn1_idx) { *offset_floats = lo.y1[i]; *count = (int64_t)B * b.Lmid * b.nf; return KWS_OK; }
      if (index == b.bn2_idx) { *offset_floats = lo.y2[i]; *count = (int64_t)B * b.Lmid * b.nf; return KWS_OK; }
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
  if (n->lm->style != 0) {
    kws_gp_tail_args g;
    memset(&g, 0, sizeof(g));
    g.x = t.x; g.Wd = params + n->lm->dk; g.probs = probs; g.B = B; g.T = n->lm->T; g.C = n->lm->C; g.NC = n->lm->NC;
    g.keep_prob = 1.f; g.loss_batch = 1; g.pool_max = n->lm->style == 1;
    g.bd = n->lm->style >= 2 ? params + n->lm->db : nullptr;
    return kws_gp_tail_launch(&g, 0, st);
  }
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
  {  // every pointwise kernel transposed for its dgrad GEMM, KWS_TRANSPOSE_BATCH matrices per launch
    std::vector<const float*> tin;
    std::vector<float*> tout;
    std::vector<int> trows, tcols;
    auto add = [&](int64_t src, int64_t dst, int rows, int cols) {
      tin.push_back(params + src); tout.push_back(ws + dst); trows.push_back(rows); tcols.push_back(cols);
    };
    for (size_t i = 0; i < p.blocks.size(); ++i) {
      const LmBlock& b = p.blocks[i];
      add(b.pw1, lo.wt_pw1[i], b.cin, b.nf);
      add(b.pw2, lo.wt_pw2[i], b.nf, b.nf);
      if (b.has_short) add(b.ws, lo.wt_ws[i], b.cin, b.nf);
    }
    for (size_t j = 0; j < p.plain.size(); ++j) add(p.plain[j].pw, lo.wt_plain[j], p.plain[j].cin, p.plain[j].cout);
    if (p.style == 1) add(p.ctx_pw, lo.wt_ctx, p.C0, p.C0);
    for (size_t o = 0; o < tin.size(); o += KWS_TRANSPOSE_BATCH) {
      const int nbat = (int)std::min<size_t>(KWS_TRANSPOSE_BATCH, tin.size() - o);
      KWS_TRY(kws_transpose_batch_f32(tin.data() + o, tout.data() + o, trows.data() + o, tcols.data() + o, nbat, st));
    }
  }
  float* part = ws + lo.part;
  float* red = ws + lo.red;
  float* coef = ws + lo.coef;
  float* G = ws + lo.G;
  float* DZ = ws + lo.DZ;
  float* dO = ws + lo.dOa;
  float* dX = ws + lo.dOb;
  KwsSlabQueue sq;
  sq.base = ws + lo.tnq; sq.cap = lo.tnq_floats;
  sq.allow_pair = kws_net_get_gemm_mode(n) != 1;    // mode 1: the A/B reference schedule (separate input- / weight-gradient launches)
  // Depthwise backward kernels that leave ONLY a weight gradient behind (a block's first depthwise convolution, the context
  // block's, the first plain block's: their input is a materialised activation, no BatchNorm in front): nothing on the
  // dependency chain needs the fold of their partial rows, so the rows stay in regions of their own and one launch folds up to
  // KWS_DW_FIN_BATCH layers at the end of the pass (round 4: one small launch per block off the chain).
  struct DwFinQueue {
    const float* part[KWS_DW_FIN_BATCH];
    float* dw[KWS_DW_FIN_BATCH];
    int n_parts[KWS_DW_FIN_BATCH], C[KWS_DW_FIN_BATCH];
    int count = 0;
    float* base = nullptr;
    int64_t used = 0, cap = 0;
    hipStream_t st = nullptr;
    int flush() {
      if (count == 0) return KWS_OK;
      const int rc = kws_dw_grad_finalize_batch(part, n_parts, C, dw, count, st);
      count = 0;
      used = 0;
      return rc;
    }
    // a region for the rows of one depthwise backward launch (floats = kws_dwconv_bwd_part_floats) whose fold writes dW
    int take(int64_t floats, int Cc, float* dW, float** out) {
      const int64_t need = (floats + 63) / 64 * 64;
      if (count == KWS_DW_FIN_BATCH || used + need > cap) KWS_TRY(flush());
      if (need > cap) return KWS_E_WORKSPACE;
      *out = base + used;
      part[count] = base + used; dw[count] = dW; n_parts[count] = (int)(floats / (5 * Cc)); C[count] = Cc;
      used += need;
      ++count;
      return KWS_OK;
    }
  } dq;
  dq.base = ws + lo.dwq; dq.cap = lo.dwq_floats; dq.st = st;
  // join backward + BatchNorm backward in two passes (round 4): reductions, fold (dgamma, dbeta, c1 | c2), then the masked /
  // pool-routed gradient is recomputed and dy written directly - 5 tensor passes instead of the 6 of "kws_block_out_bwd,
  // fold, kws_bn_bwd_apply", <= 256 partial rows (no slice fold), one launch less per join.  `outp` may be `dOin` (pool 1).
  auto join_bwd = [&](const float* dOin, const float* yv, const BnRef& r, int bn_idx, float* outp, int L, int C, int pool,
                      int relu) -> int {
    KWS_TRY(kws_block_join_bwd(dOin, yv, c.bn_at(bn_idx), nullptr, nullptr, nullptr, part, 1, B, L, C, pool, relu, st));
    KWS_TRY(kws_dw_bwd_finalize(part, kws_block_join_bwd_parts(B, L, C, pool), (int64_t)B * L, C, nullptr, grads + r.gamma,
                                grads + r.beta, coef, red, st));
    return kws_block_join_bwd(dOin, yv, c.bn_at(bn_idx), params + r.gamma, coef, outp, nullptr, 2, B, L, C, pool, relu, st);
  };
  // ---- tail forward + backward ----
  if (p.style != 0) {
    kws_gp_tail_args g;
    memset(&g, 0, sizeof(g));
    g.x = t.x; g.Wd = params + p.dk; g.labels = y_onehot; g.probs = probs; g.dX = dO; g.fd = ws + lo.fd; g.dl = ws + lo.dl;
    g.per_loss = ws + lo.per_loss; g.per_correct = ws + lo.per_correct; g.B = B; g.T = p.T; g.C = p.C; g.NC = p.NC;
    g.pool_max = p.style == 1; g.loss_kind = p.style == 1 ? 0 : 1;
    g.bd = p.style >= 2 ? params + p.db : nullptr;
    g.seed = seed; g.step = step; g.keep_prob = p.drop_keep; g.label_smoothing = STEFFE_LABEL_SMOOTH;
    g.loss_batch = loss_batch; g.row_offset = row_offset;
    KWS_TRY(kws_gp_tail_launch(&g, 1, st));
    KWS_TRY(kws_metrics_launch(g.per_loss, g.per_correct, B, metrics, st));
    KWS_TRY(kws_small_wgrad_launch(g.fd, g.dl, grads + p.dk, p.style >= 2 ? grads + p.db : nullptr, B,
                                   p.style == 1 ? 2 * p.C : p.C, p.NC, ws + lo.swg, st));
    // ---- plain blocks after the residual stack, last to first: dO is the gradient wrt the materialised activation
    for (int j = (int)p.plain.size() - 1; j >= 0; --j) {
      const LmPlain& q = p.plain[j];
      const int64_t M = (int64_t)B * q.Lout;
      if (j == (int)p.plain.size() - 1) {
        KWS_TRY(join_bwd(dO, ws + lo.py[j], q.bn, q.bn_idx, G, q.Lout, q.cout, 1, 1));
      }  // else: G already holds dy of this block (pass 2 of the next block's depthwise backward)
      KWS_TRY(sq.pair(G, ws + lo.wt_plain[j], DZ, ws + lo.pz[j], grads + q.pw, M, q.cin, q.cout, st));   // dgrad + wgrad: one launch
      const int np = (int)(kws_dwconv_bwd_part_floats(B, q.Lin, q.cin) / (5 * q.cin));
      if (j > 0) {
        const LmPlain& r = p.plain[j - 1];
        KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.py[j - 1], c.bn_at(r.bn_idx), params + q.dw, nullptr, nullptr, part, 1, B,
                                      q.Lin, q.Lout, q.cin, q.stride, q.pad_l, st));
        KWS_TRY(kws_dw_bwd_finalize(part, np, (int64_t)B * q.Lin, q.cin, grads + q.dw, grads + r.bn.gamma, grads + r.bn.beta,
                                    coef, red, st));
        KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.py[j - 1], c.bn_at(r.bn_idx), params + q.dw, coef, G, nullptr, 2, B, q.Lin,
                                      q.Lout, q.cin, q.stride, q.pad_l, st));
      } else {
        const float* xin = ws + lo.o[p.blocks.size() - 1];
        float* dpart;
        KWS_TRY(dq.take(kws_dwconv_bwd_part_floats(B, q.Lin, q.cin), q.cin, grads + q.dw, &dpart));
        KWS_TRY(kws_dwconv_bwd_f32(DZ, xin, nullptr, params + q.dw, dX, dpart, B, q.Lin, q.Lout, q.cin, q.stride, q.pad_l, st));
        std::swap(dO, dX);
      }
    }
  } else {
  t.labels = y_onehot; t.probs = probs; t.dX = dO; t.fd = ws + lo.fd; t.dl = ws + lo.dl; t.gu = ws + lo.gu;
  t.part = part; t.coef = ws + lo.coef2; t.d_gamma = grads + p.att_bn.gamma; t.d_beta = grads + p.att_bn.beta;
  t.per_loss = ws + lo.per_loss; t.per_correct = ws + lo.per_correct; t.att = ws + lo.att;
  t.seed = seed; t.step = step; t.loss_batch = loss_batch; t.row_offset = row_offset;
  KWS_TRY(kws_lm_tail_fwd(&t, 1, st));
  KWS_TRY(kws_metrics_launch(t.per_loss, t.per_correct, B, metrics, st));
  KWS_TRY(kws_small_wgrad_launch(t.fd, t.dl, grads + p.dk, grads + p.db, B, p.C, p.NC, ws + lo.swg, st));
  KWS_TRY(kws_lm_tail_bwd(&t, st));
  KWS_TRY(kws_dw_bwd_finalize(part, B, 1, p.C, grads + p.att_dw, nullptr, grads + p.att_pw, nullptr, red, st));
  }
  // ---- residual blocks, last to first ----
  for (int i = (int)p.blocks.size() - 1; i >= 0; --i) {
    const LmBlock& b = p.blocks[i];
    const int64_t M = (int64_t)B * b.Lmid;
    const float* xin = i == 0 ? (p.style == 1 ? ws + lo.ac : ws + lo.a0) : ws + lo.o[i - 1];
    // main branch: join backward (maxpool routing + ReLU6 mask) -> BN2 -> pointwise 2
    int np;
    if (b.pool3) {   // the 3-wide SAME join keeps the one-pass form (an input position collects up to three windows)
      KWS_TRY(kws_block_out3_bwd(dO, ws + lo.y2[i], c.bn_at(b.bn2_idx), G, part, B, b.Lmid, b.Lout, b.nf, b.stride, b.ppad, st));
      np = (int)(kws_block_out3_bwd_part_floats(B, b.Lmid, b.nf) / (5 * b.nf));
      KWS_TRY(kws_dw_bwd_finalize(part, np, M, b.nf, nullptr, grads + b.bn2.gamma, grads + b.bn2.beta, coef, red, st));
      KWS_TRY(kws_bn_bwd_apply(G, ws + lo.y2[i], c.bn_at(b.bn2_idx), params + b.bn2.gamma, coef, M, b.nf, st));
    } else {
      KWS_TRY(join_bwd(dO, ws + lo.y2[i], b.bn2, b.bn2_idx, G, b.Lmid, b.nf, b.pool, 1));
    }
    KWS_TRY(sq.pair(G, ws + lo.wt_pw2[i], DZ, ws + lo.z2[i], grads + b.pw2, M, b.nf, b.nf, st));   // dgrad + wgrad: one launch where eligible
    // depthwise 2 -> BN1 -> pointwise 1
    // (two passes over dz and y1 instead of "store g, then kws_bn_bwd_apply": the masked gradient is never stored)
    KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.y1[i], c.bn_at(b.bn1_idx), params + b.dw2, nullptr, nullptr, part, 1, B,
                                  b.Lmid, b.Lmid, b.nf, 1, 1, st));
    np = (int)(kws_dwconv_bwd_part_floats(B, b.Lmid, b.nf) / (5 * b.nf));
    KWS_TRY(kws_dw_bwd_finalize(part, np, M, b.nf, grads + b.dw2, grads + b.bn1.gamma, grads + b.bn1.beta, coef, red, st));
    KWS_TRY(kws_dwconv_bwd_bn_f32(DZ, ws + lo.y1[i], c.bn_at(b.bn1_idx), params + b.dw2, coef, G, nullptr, 2, B, b.Lmid,
                                  b.Lmid, b.nf, 1, 1, st));
    KWS_TRY(sq.pair(G, ws + lo.wt_pw1[i], DZ, ws + lo.z1[i], grads + b.pw1, M, b.cin, b.nf, st));
    // depthwise 1 on the (materialised) block input
    float* dpart;      // only a weight gradient comes out of these rows: folded with the other blocks' at the end of the pass
    KWS_TRY(dq.take(kws_dwconv_bwd_part_floats(B, b.Lin, b.cin), b.cin, grads + b.dw1, &dpart));
    // round 6: with a strided shortcut whose rows sit at whole multiples of the stride, the shortcut branch runs FIRST and its input
    // gradient joins the depthwise input gradient inside that kernel (same single addition per element: bit-identical) instead of an
    // add_strided pass afterwards; gemm mode 1 (the A/B reference schedule) keeps the order and the launches of rounds 3 - 5
    const bool short_first = b.has_short && sq.allow_pair && b.stride >= 2 && (int64_t)(b.Lout - 1) * b.stride < b.Lin;
    if (b.has_short && short_first) {
      const int64_t Mo = (int64_t)B * b.Lout;
      KWS_TRY(join_bwd(dO, ws + lo.ys[i], b.bns, b.bns_idx, dO, b.Lout, b.nf, 1, 0));   // the shortcut's BN: no mask, in place
      KWS_TRY(sq.gemm_gather(xin, &b.gs, dO, grads + b.ws, B, b.nf, st));
      KWS_TRY(kws_gemm_nn_f32(dO, ws + lo.wt_ws[i], ws + lo.DXS, Mo, b.nf, b.cin, nullptr, st));
      KWS_TRY(kws_dwconv_bwd_acc_strided_f32(DZ, xin, params + b.dw1, ws + lo.DXS, b.stride, b.Lout, dX, dpart, B, b.Lin, b.Lmid, b.cin,
                                             b.s1, b.pad1, st));
    } else if (!b.has_short) {   // identity shortcut: the join's other gradient is added while the depthwise input gradient is written
      KWS_TRY(kws_dwconv_bwd_acc_f32(DZ, xin, params + b.dw1, dO, dX, dpart, B, b.Lin, b.Lmid, b.cin, b.s1, b.pad1, st));
    } else {
      KWS_TRY(kws_dwconv_bwd_f32(DZ, xin, nullptr, params + b.dw1, dX, dpart, B, b.Lin, b.Lmid, b.cin, b.s1, b.pad1, st));
      // residual branch
      const int64_t Mo = (int64_t)B * b.Lout;
      KWS_TRY(join_bwd(dO, ws + lo.ys[i], b.bns, b.bns_idx, dO, b.Lout, b.nf, 1, 0));   // the shortcut's BN: no mask, in place
      KWS_TRY(sq.gemm_gather(xin, &b.gs, dO, grads + b.ws, B, b.nf, st));   // slabs queued: summed with the pass's other weight gradients
      KWS_TRY(kws_gemm_nn_f32(dO, ws + lo.wt_ws[i], ws + lo.DXS, Mo, b.nf, b.cin, nullptr, st));
      KWS_TRY(kws_add_strided_f32(dX, ws + lo.DXS, B, b.Lin, b.Lout, b.cin, b.stride, st));
    }
    std::swap(dO, dX);
  }
  if (p.style == 1) {  // ---- context block: dO is the gradient wrt its activated output ----
    const int64_t M = (int64_t)B * p.L0;
    KWS_TRY(join_bwd(dO, ws + lo.yc, p.ctx_bn, p.ctx_bn_idx, G, p.L0, p.C0, 1, 1));
    KWS_TRY(sq.pair(G, ws + lo.wt_ctx, DZ, ws + lo.zc, grads + p.ctx_pw, M, p.C0, p.C0, st));
    float* dpart;
    KWS_TRY(dq.take(kws_dwconv_bwd_part_floats(B, p.L0, p.C0), p.C0, grads + p.ctx_dw, &dpart));
    KWS_TRY(kws_dwconv_bwd_f32(DZ, ws + lo.a0, nullptr, params + p.ctx_dw, dX, dpart, B, p.L0, p.L0, p.C0, 1, 1, st));
    std::swap(dO, dX);
  }
  if (p.style == 3) {  // ---- the two stems: split the gradient of the concatenation, then each like a first convolution
    const int64_t M = (int64_t)B * p.L0;
    float* dOm = dX;
    float* dOr = dX + M * p.Cm;
    KWS_TRY(copy_cols(dO, p.C0, dOm, p.Cm, M, p.Cm, st));
    KWS_TRY(copy_cols(dO + p.Cm, p.C0, dOr, p.Cr, M, p.Cr, st));
    struct Stem { float* d; const float* y; int idx; int C; const BnRef* bn; const kws_gather_t* g; int64_t w; };
    const Stem stems[2] = {{dOm, ws + lo.y0, 1, p.Cm, &p.bn0, &p.g0, p.conv1},
                           {dOr, ws + lo.y0 + M * p.Cm, 2, p.Cr, &p.bn0r, &p.g0r, p.conv1r}};
    for (const Stem& sm : stems) {
      KWS_TRY(join_bwd(sm.d, sm.y, *sm.bn, sm.idx, G, p.L0, sm.C, 1, 1));
      KWS_TRY(sq.gemm_gather(x, sm.g, G, grads + sm.w, B, sm.C, st));
    }
    KWS_TRY(sq.flush(st));
    KWS_TRY(dq.flush());
    return KWS_OK;
  }
  KWS_TRY(dq.flush());     // the depthwise weight gradients whose rows were parked
  // ---- first convolution ----
  {
    KWS_TRY(join_bwd(dO, ws + lo.y0, p.bn0, 1, G, p.L0, p.C0, 1, 1));
    if (p.style != 1 && p.Fp == p.F) {   // its gradient is written in place: the slabs join the pass's batched sum
      KWS_TRY(sq.gemm_gather(x, &p.g0, G, grads + p.conv1, B, p.C0, st));
      return sq.flush(st);
    }
    KWS_TRY(sq.flush(st));   // the pointwise weight gradients of the whole pass: one sum (two past 16 layers)
    if (p.style == 1) {  // gradient of the zero-padded [76, C0] kernel; its first 75 rows are the kernel's
      float* gw = ws + lo.gwpad;
      KWS_TRY(kws_gemm_tn_gather_f32(x, &p.g0, G, gw, B, p.C0, ws + lo.tn, st));
      KWS_HIP(hipMemcpyAsync(grads + p.conv1, gw, (size_t)p.K0 * p.C0 * 4, hipMemcpyDeviceToDevice, st));
    } else {  // (Fp != F here: the unpadded case left above) the forward of this step left the padded input in xpad
      float* gw = ws + lo.gwpad;
      KWS_TRY(kws_gemm_tn_gather_f32(ws + lo.xpad, &p.g0, G, gw, B, p.C0, ws + lo.tn, st));
      KWS_HIP(hipMemcpy2DAsync(grads + p.conv1, (size_t)p.F * p.C0 * 4, gw, (size_t)p.Fp * p.C0 * 4, (size_t)p.F * p.C0 * 4, 3,
                               hipMemcpyDeviceToDevice, st));
    }
  }
  return KWS_OK;
}
