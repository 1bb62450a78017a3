// Shared between the two network programs (net.hip: conv_1d_time_sliced_with_attention,
// net_logmfcc.hip: conv_1d_log_mfcc).  Not part of the public C ABI.
#pragma once
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "internal.h"

constexpr float KWS_BN_EPS = 1e-3f;
constexpr float KWS_BN_MOMENTUM = 0.99f;
constexpr float KWS_L2_COEF = 1e-5f;

struct BnRef {
  int64_t gamma, beta;  // param offsets
  int64_t mm, mv;       // state offsets
  int C;
};
struct Block {
  int stride, pad_l, cin, cout, Lin, Lout;
  int64_t dw, pw;  // param offsets
  BnRef bn;        // BN after the pointwise conv
};

struct LmProgram;  // conv_1d_log_mfcc layer table (net_logmfcc.hip)

struct kws_net {
  kws_net_config_t cfg;
  std::vector<kws_tensor_info_t> tensors;
  int64_t n_params = 0, n_state = 0;
  // TS_ATTENTION
  int L_in = 0;        // samples per clip
  int L1 = 0, C1 = 0;  // conv1 output
  int64_t conv1 = 0;
  BnRef bn1;
  std::vector<Block> blocks;
  int T = 0, C = 0, NC = 0;
  int64_t d1k = 0, d1b = 0, d2k = 0;
  kws_gather_t gather1;   // the reference's view: 3 taps of 40 samples, taps 20 samples apart (K = 120)
  kws_gather_t gather1f;  // folded view used by the GEMMs: ONE tap of 80 contiguous samples (see net.hip fold_taps_kernel)
  int K1f = 0;            // folded K
  // LOG_MFCC
  LmProgram* lm = nullptr;
  // arithmetic of the pointwise GEMMs (kws_net_set_gemm_mode): 0 = f32 MFMA, 2 = fp16 x 2 split products (A/B arm)
  std::atomic<int> gemm_mode{0};
};

// Appends a Keras-named tensor to the flat parameter (or state) buffer; returns its float offset.
int64_t kws_net_add_tensor(kws_net* n, const std::string& name, std::vector<int64_t> shape, bool is_state, float l2,
                           int fan_in, int fan_out, float init);
BnRef kws_net_add_bn(kws_net* n, int idx, int C);

struct Bump {
  int64_t cur = 0;  // in floats
  int64_t take(int64_t floats) {
    const int64_t o = cur;
    cur += (floats + 63) / 64 * 64;  // 256-B granules
    return o;
  }
};

// ---- conv_1d_log_mfcc program -----------------------------------------------------------------------
int lm_build(kws_net* n);
int steffe_build(kws_net* n);
int residual_build(kws_net* n);
int mfcc_raw_build(kws_net* n);
void lm_free(kws_net* n);
int64_t lm_workspace_bytes(const kws_net* n, int B, int training);
int lm_debug_view(const kws_net* n, int B, int training, int what, int index, int64_t* offset_floats, int64_t* count);
int lm_predict(const kws_net* n, const float* params, const float* state, const float* x, int B, float* probs,
               float* ws, int64_t ws_bytes, hipStream_t st);
int lm_train(const kws_net* n, const float* params, float* state, const float* x, const float* y_onehot, int B,
             float* grads, float* probs, float* metrics, uint64_t seed, uint32_t step, int64_t row_offset,
             int loss_batch, float* ws, int64_t ws_bytes, hipStream_t st);

// ---- residual-block / log-mfcc tail launchers (resblock.hip) ------------------------------------------
// o = maxpool_P(relu6(bn(y))) + (res_bn ? res_bn.scale*res + res_bn.shift : res)
int kws_block_out_fwd(const float* y, const float* bn, const float* res, const float* res_bn, float* o, int B,
                      int L, int C, int pool, hipStream_t st);
// resblock.hip (round 6): that join and the NEXT block's first depthwise convolution (k 3, stride 1, pad (1, 1); kernel w [3, C]) in one pass
int kws_block_out_dw_fwd(const float* y, const float* bn, const float* res, const float* res_bn, const float* w, float* o, float* z, int B,
                         int L, int C, int pool, hipStream_t st);
// g[b,u,c] = [u wins its pool window] * dO[b,u/P,c] * (relu ? relu6'(bn(y)) : 1); part = [blocks][5][C] sums of
// (g, g*xhat, 0, 0, 0)
// two-pass join backward + BatchNorm backward (resblock.hip block_join_bwd_kernel): pass 1 leaves kws_block_join_bwd_parts()
// partial rows [5][C] (<= 256 per launch: no slice fold in front of kws_dw_bwd_finalize), pass 2 writes dy directly
int kws_block_join_bwd_parts(int B, int L, int C, int pool);
int kws_block_join_bwd(const float* dO, const float* y, const float* bn, const float* gamma, const float* coef, float* out,
                       float* part, int pass, int B, int L, int C, int pool, int relu, hipStream_t st);
int64_t kws_block_out_bwd_part_floats(int B, int L, int C, int pool);
int kws_block_out_bwd(const float* dO, const float* y, const float* bn, float* g, float* part, int B, int L, int C,
                      int pool, int relu, hipStream_t st);
int kws_block_out3_fwd(const float* y, const float* bn, const float* res, const float* res_bn, float* o, int B, int L,
                       int Lo, int C, int stride, int pad_l, hipStream_t st);
int64_t kws_block_out3_bwd_part_floats(int B, int L, int C);
int kws_block_out3_bwd(const float* dO, const float* y, const float* bn, float* g, float* part, int B, int L, int Lo,
                       int C, int stride, int pad_l, hipStream_t st);
int kws_add_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t st);
// out[b, stride*t, :] += in[b, t, :]
int kws_add_strided_f32(float* out, const float* in, int B, int L_out, int L_in, int C, int stride, hipStream_t st);

struct kws_lm_tail_args {
  const float* x;       // [B, T, C] block-stack output
  const float* wa;      // [3, C] attention depthwise kernel
  const float* Wa;      // [C]    attention pointwise kernel (C -> 1)
  const float* bn_gamma; const float* bn_beta; float* mm; float* mv;   // attention BN (1 channel)
  const float* Wd;      // [C, NC]
  const float* bd;      // [NC]
  const float* labels;  // [B, NC]
  float* probs;         // [B, NC]
  float* u;             // [B, T] attention logits (pre-BN)
  float* bn;            // [4] scale|shift|mean|rstd of the attention BN
  float* dX;            // [B, T, C] gradient wrt x (train)
  float* fd;            // [B, C] dropped features
  float* dl;            // [B, NC]
  float* gu;            // [B, T] masked gradient wrt the attention BN output
  float* part;          // [B][5][C] per-clip partials (dWa | 0 | dwa taps)
  float* coef;          // [2]
  float* d_gamma; float* d_beta;   // grads of the attention BN
  float* per_loss; float* per_correct; float* att;
  int B, T, C, NC; uint64_t seed; uint32_t step; float keep_prob; int loss_batch; int64_t row_offset;
};
// Global-pooling tails; training also writes dX, the dropped features and dlogits (for the dense wgrad).
//   steffeNet (model.py:1712-1718, 1722-1724): GlobalMaxPooling1D ++ GlobalAveragePooling1D -> Dropout -> Dense(no
//     bias) + softmax -> label-smoothed CE: pool_max = 1, bd = NULL, loss_kind = 0
//   conv_1d_residual (model.py:898-905): GlobalAveragePooling1D -> Dropout -> Dense + softmax -> keras
//     categorical_crossentropy: pool_max = 0, loss_kind = 1
struct kws_gp_tail_args {
  const float* x;       // [B, T, C] block-stack output
  const float* Wd;      // [F, NC], F = 2C (max ++ avg) or C (avg)
  const float* bd;      // [NC] or NULL
  int pool_max, loss_kind;
  const float* labels;  // [B, NC]
  float* probs;         // [B, NC]
  float* dX;            // [B, T, C]
  float* fd;            // [B, F]
  float* dl;            // [B, NC]
  float* per_loss;
  float* per_correct;
  int B, T, C, NC;
  uint64_t seed; uint32_t step; float keep_prob; float label_smoothing; int loss_batch; int64_t row_offset;
};
int kws_gp_tail_launch(const kws_gp_tail_args* a, int training, hipStream_t st);
int kws_lm_tail_fwd(const kws_lm_tail_args* a, int training, hipStream_t st);   // logits -> BN stats -> probs (+ tail backward when training)
int kws_lm_tail_bwd(const kws_lm_tail_args* a, hipStream_t st);                 // attention BN backward + dX accumulation + per-clip partials
