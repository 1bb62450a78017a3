// Launchers shared between translation units of libkws_hip.so (not part of the public C ABI).
#pragma once
#include "common.h"

struct kws_ts_tail_args {
  const float* y; const float* bn; const float* W1; const float* b1; const float* W2; const float* labels;
  float* probs; float* g; float* part; float* xd; float* fd; float* dl1; float* dl2; float* per_loss;
  float* per_correct; float* att; int B, T, C, NC; uint64_t seed; uint32_t step; float keep_prob; float label_smoothing;
  int loss_batch; int64_t row_offset; int train;
};
int kws_ts_tail_launch(const kws_ts_tail_args* p, hipStream_t st);
constexpr int KWS_SMALL_WGRAD_SLICES = 32;  // scratch: KWS_SMALL_WGRAD_SLICES * K * N floats
int kws_small_wgrad_launch(const float* X, const float* D, float* out, float* out_bias, int B, int K, int N,
                           float* scratch, hipStream_t st);
int kws_metrics_launch(const float* per_loss, const float* per_correct, int B, float* metrics, hipStream_t st);
// tail.hip (round 4): what follows the tail kernel off the dependency chain - the SLICES of the two dense layers' weight
// gradients (X2 [B, K2] x D2 [B, N2] and X1 x D1, KWS_SMALL_WGRAD_SLICES slices each into ws2 / ws1), the bias gradient of the
// first (column sums of D1, may be NULL) and the batch metrics - as ONE launch instead of six.  The slices are NOT summed here:
// the caller adds (ws, out, K N, -S) to a slab batch (negative S: kws_small_wgrad_launch's summation order, bit-identical).
// Returns 1 (nothing launched) for shapes the fused kernel does not take.
struct kws_tail_post_args {
  const float* X2; const float* D2; float* ws2; int K2, N2;
  const float* X1; const float* D1; float* ws1; int K1, N1;
  float* bias1;
  const float* per_loss; const float* per_correct; float* metrics;
  int B;
};
int kws_tail_post_launch(const kws_tail_post_args* a, int* S_out, hipStream_t st);
// conv1.hip: the raw-waveform net's folded first convolution as a Toeplitz GEMM (forward with BN statistics rows, weight
// gradient with its own slab workspace); kws_conv1_supported() says whether a (folded, unfolded) descriptor pair / width qualifies
bool kws_conv1_supported(const kws_gather_t* g, const kws_gather_t* unfolded, int N);
int kws_conv1_stats_rows(int64_t M);
// g = the folded view (one tap of 80 samples), unfolded = the reference's view (taps x cin, taps `stride_j` apart); W and
// dW are the UNFOLDED kernel / gradient [taps][cin][128]: the forward kernel folds W while loading it into registers, the
// weight gradient's slab sum writes the taps directly
int kws_conv1_fwd(const float* x, const kws_gather_t* g, const kws_gather_t* unfolded, const float* W, float* y, int B, int N,
                  float* stats, hipStream_t st);
int64_t kws_conv1_wgrad_workspace_floats(int64_t M);
int kws_conv1_wgrad(const float* x, const kws_gather_t* g, const kws_gather_t* unfolded, const float* G, float* dW, int B,
                    int N, float* workspace, hipStream_t st);
int kws_conv1_wgrad_slabs(const float* x, const kws_gather_t* g, const kws_gather_t* unfolded, const float* G, float* dW, int B,
                          int N, float* workspace, const float* const* sl_ws, float* const* sl_out, const int64_t* sl_n,
                          const int* sl_S, int n_sl, hipStream_t st);
// gemm.hip: first stage of a two-stage slab sum: every group of `per_group` slabs is summed over the group's first slab
extern "C" int kws_reduce_slab_groups_f32(float* ws, int64_t n, int S, int per_group, hipStream_t st);
// gemm.hip: out[i] = sum over S slabs of ws[s][i], fixed order (n % 4 == 0)
extern "C" int kws_reduce_slabs_f32(const float* ws, float* out, int64_t n, int S, hipStream_t st);
// (the |x|-maximum producers kws_dwconv_fwd_amax_f32 / kws_dwconv_bwd_bn_amax_f32 / kws_bn_bwd_apply_amax are public: include/kws_hip.h)
extern "C" int kws_dwconv_bwd_acc_f32(const float* dz, const float* y, const float* w, const float* add, float* g, float* part,
                                      int B, int L_in, int L_out, int C, int stride, int pad_l, hipStream_t st);
// dwconv.hip (round 6): as above with row q of add [B, add_len, C] added at position add_stride q (a strided shortcut's input gradient)
extern "C" int kws_dwconv_bwd_acc_strided_f32(const float* dz, const float* y, const float* w, const float* add, int add_stride, int add_len,
                                              float* g, float* part, int B, int L_in, int L_out, int C, int stride, int pad_l, hipStream_t st);
// gemm.hip: the weight-gradient GEMM without its slab sum, and the slab sums of several of them in one launch
constexpr int KWS_SLAB_BATCH = 16;
extern "C" int kws_gemm_tn_slabs_f32(const float* A, const float* G, int64_t M, int K, int N, float* workspace, int* S,
                                     hipStream_t stream);
// gemm.hip: a layer's input-gradient GEMM and the slabs of its weight-gradient GEMM in one launch (returns 1 = not eligible, nothing launched)
extern "C" int kws_gemm_dgrad_wgrad_f32(const float* dY, const float* WT, float* dZ, const float* Z, int64_t M, int cin, int cout,
                                        float* workspace, int* S, hipStream_t stream);
// gemm.hip (round 6): a gather that is every s-th ROW of a matrix (1 x 1 convolution with stride s over an even-length input) as a
// plain GEMM with a row pitch on the wave-specialised kernels; both return 1 (nothing launched) for shapes those do not take
extern "C" bool kws_gather_strided_rows(const kws_gather_t* g, int* lda);
extern "C" int kws_gemm_nn_strided_f32(const float* A, int lda, const float* W, float* C, int64_t M, int K, int N, float* stats_part,
                                       hipStream_t stream);
extern "C" int kws_gemm_tn_slabs_strided_f32(const float* A, int lda, const float* G, int64_t M, int K, int N, float* workspace, int* S,
                                             hipStream_t stream);
// gemm.hip (round 5): the gathered weight-gradient GEMM without its slab sum (queue the slabs with a NEGATIVE count)
extern "C" int kws_gemm_tn_gather_slabs_f32(const float* X, const kws_gather_t* g, const float* G, int B, int N, float* workspace, int* S,
                                            hipStream_t stream);
extern "C" int kws_reduce_slabs_batch(const float* const* ws, float* const* out, const int64_t* n, const int* S, int count,
                                      hipStream_t stream);
extern "C" int64_t kws_gemm_tn_workspace_floats(int64_t M, int K, int N);
// (kws_slab_batch_fill: an entry with S[i] < 0 holds |S[i]| slabs and is summed in the order of reduce_slabs_kernel: `order` below.)
// The slab sums of several weight-gradient GEMMs as one grid (gemm.hip reduce_slabs_batch_kernel; since round 4 also the tail blocks
// of conv1.hip's conv1_wgrad_slabsum_kernel): one workgroup of 256 threads = 64 float4 columns of one GEMM; 4 slab groups
// (k = grp, grp + 4, ...) with four loads in flight each, combined in a fixed order: bit-reproducible.
struct SlabBatch {
  const float* ws[KWS_SLAB_BATCH];
  float* out[KWS_SLAB_BATCH];
  int64_t n4[KWS_SLAB_BATCH];
  int S[KWS_SLAB_BATCH], blk_end[KWS_SLAB_BATCH];
  int order[KWS_SLAB_BATCH];   // 0: four accumulators per slab group (k, k+4, k+8, k+12); 1: the two of reduce_slabs_kernel (k, k+4) -
  int n;                       // what kws_small_wgrad_launch's own slab sum uses, so that its slices can join a batch bit for bit
};
extern "C" int kws_slab_batch_fill(SlabBatch* b, const float* const* ws, float* const* out, const int64_t* n, const int* S, int count,
                                   int* blocks_out, double* bytes_out);
#ifdef __HIPCC__
__device__ __forceinline__ void kws_add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void kws_reduce_slabs_batch_body(const SlabBatch& b, int bid, float4 (*red)[64]) {
  int m = 0;
  while (m + 1 < b.n && bid >= b.blk_end[m]) ++m;
  const int t = bid - (m ? b.blk_end[m - 1] : 0);
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t n4 = b.n4[m];
  const int S = b.S[m];
  const int64_t i = (int64_t)t * 64 + col;
  const float4* w = reinterpret_cast<const float4*>(b.ws[m]);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (i < n4 && b.order[m]) {
    float4 t2 = s0;
    int k = grp;
    for (; k + 4 < S; k += 8) {
      const float4 v = w[(int64_t)k * n4 + i], u = w[(int64_t)(k + 4) * n4 + i];
      kws_add4(s0, v); kws_add4(t2, u);
    }
    if (k < S) kws_add4(s0, w[(int64_t)k * n4 + i]);
    kws_add4(s0, t2);
  } else if (i < n4) {
    int k = grp;
    for (; k + 12 < S; k += 16) {
      const float4 v0 = w[(int64_t)k * n4 + i], v1 = w[(int64_t)(k + 4) * n4 + i];
      const float4 v2 = w[(int64_t)(k + 8) * n4 + i], v3 = w[(int64_t)(k + 12) * n4 + i];
      kws_add4(s0, v0); kws_add4(s1, v1); kws_add4(s2, v2); kws_add4(s3, v3);
    }
    for (; k < S; k += 4) kws_add4(s0, w[(int64_t)k * n4 + i]);
    kws_add4(s0, s1); kws_add4(s2, s3); kws_add4(s0, s2);
  }
  red[grp][col] = s0;
  __syncthreads();
  if (grp == 0 && i < n4) {
    float4 r = red[0][col];
    kws_add4(r, red[1][col]); kws_add4(r, red[2][col]); kws_add4(r, red[3][col]);
    reinterpret_cast<float4*>(b.out[m])[i] = r;
  }
}
#endif
// The weight-gradient GEMMs of one backward pass, their slab sums deferred: gemm() launches the GEMM into the next piece of
// `base` (cap floats: the sum of the calls' kws_gemm_tn_workspace_floats, each rounded up to 64) and flush() sums the slabs
// of up to KWS_SLAB_BATCH of them in one launch.  A gradient is final only after the flush that follows its gemm().
struct KwsSlabQueue {
  const float* ws[KWS_SLAB_BATCH];
  float* out[KWS_SLAB_BATCH];
  int64_t n[KWS_SLAB_BATCH];
  int S[KWS_SLAB_BATCH];
  int count = 0;
  float* base = nullptr;
  int64_t used = 0, cap = 0;
  bool allow_pair = true;      // false (gemm mode 1): pair() makes the two launches of rounds 1 - 3
  int flush(hipStream_t st) {
    if (count == 0) return 0;
    const int rc = kws_reduce_slabs_batch(ws, out, n, S, count, st);
    count = 0;
    used = 0;           // the pieces are free again: later GEMMs follow the sum in stream order
    return rc;
  }
  int gemm(const float* A, const float* G, float* dW, int64_t M, int K, int N, hipStream_t st) {
    const int64_t need = (kws_gemm_tn_workspace_floats(M, K, N) + 63) / 64 * 64;
    if (count == KWS_SLAB_BATCH || used + need > cap) {
      const int rc = flush(st);
      if (rc) return rc;
    }
    if (need > cap) return KWS_E_WORKSPACE;   // (the layout reserved the sum of all calls: unreachable)
    ws[count] = base + used; out[count] = dW; n[count] = (int64_t)K * N;
    const int rc = kws_gemm_tn_slabs_f32(A, G, M, K, N, base + used, &S[count], st);
    if (rc) return rc;
    used += need;
    ++count;
    return 0;
  }
  // a GATHERED weight-gradient GEMM (first / shortcut convolutions of the residual families): its slabs join the batch in the
  // summation order of its own slab-sum launch (negative count), which they replace - one small launch less per call
  int gemm_gather(const float* X, const kws_gather_t* g, const float* G, float* dW, int B, int N, hipStream_t st) {
    const int64_t M = (int64_t)B * g->L_out;
    const int K = g->taps * g->cin;
    const int64_t need = (kws_gemm_tn_workspace_floats(M, K, N) + 63) / 64 * 64;
    if (count == KWS_SLAB_BATCH || used + need > cap) {
      const int rc = flush(st);
      if (rc) return rc;
    }
    if (need > cap) return KWS_E_WORKSPACE;
    ws[count] = base + used; out[count] = dW; n[count] = (int64_t)K * N;
    int S_ = 0, lda = 0;
    if (kws_gather_strided_rows(g, &lda)) {     // every s-th row of a matrix: the wave-specialised kernel with a row pitch (round 6)
      const int rs = kws_gemm_tn_slabs_strided_f32(X, lda, G, M, K, N, base + used, &S_, st);
      if (rs < 0) return rs;
      if (rs == 0) {
        S[count] = S_;                          // that kernel's slabs: the plain summation order
        used += need;
        ++count;
        return 0;
      }
    }
    const int rc = kws_gemm_tn_gather_slabs_f32(X, g, G, B, N, base + used, &S_, st);
    if (rc) return rc;
    S[count] = -S_;
    used += need;
    ++count;
    return 0;
  }
  // the input-gradient GEMM dZ = dY WT and the weight-gradient slabs of dW = Z^T dY of one layer: ONE launch where the fused kernel
  // takes the shapes (kws_gemm_dgrad_wgrad_f32), the two launches otherwise
  int pair(const float* dY, const float* WT, float* dZ, const float* Z, float* dW, int64_t M, int cin, int cout, hipStream_t st) {
    const int64_t need = (kws_gemm_tn_workspace_floats(M, cin, cout) + 63) / 64 * 64;
    if (count == KWS_SLAB_BATCH || used + need > cap) {
      const int rc = flush(st);
      if (rc) return rc;
    }
    if (need > cap) return KWS_E_WORKSPACE;
    const int rc = allow_pair ? kws_gemm_dgrad_wgrad_f32(dY, WT, dZ, Z, M, cin, cout, base + used, &S[count], st) : 1;
    if (rc < 0) return rc;
    if (rc == 0) {
      ws[count] = base + used; out[count] = dW; n[count] = (int64_t)cin * cout;
      used += need;
      ++count;
      return 0;
    }
    const int rc2 = kws_gemm_nn_f32(dY, WT, dZ, M, cout, cin, nullptr, st);
    if (rc2) return rc2;
    return gemm(Z, dY, dW, M, cin, cout, st);
  }
};
// bn.hip: the depthwise weight gradients of up to KWS_DW_FIN_BATCH layers folded in one launch (their partial rows kept apart)
constexpr int KWS_DW_FIN_BATCH = 16;
int kws_dw_grad_finalize_batch(const float* const* part, const int* n_parts, const int* C, float* const* dw, int count,
                               hipStream_t stream);
// bn.hip: inference BatchNorm tables of up to KWS_BN_INFER_BATCH layers in one launch
constexpr int KWS_BN_INFER_BATCH = 16;
extern "C" int kws_bn_infer_prepare_batch(const float* const* gamma, const float* const* beta, const float* const* mm, const float* const* mv,
                               float eps, const int* C, float* const* bn, int count, hipStream_t stream);
constexpr int KWS_TRANSPOSE_BATCH = 16;
extern "C" int kws_transpose_batch_f32(const float* const* in, float* const* out, const int* rows, const int* cols, int n,
                                       hipStream_t stream);

// ---- STFT plan (device tables), shared by stft.hip (generic kernel, plan) and stft4.hip (feature kernel) ------
struct kws_stft_plan {
  int frame_len, frame_step, fft_len, n_bins, n_mel, n_out;
  float log_offset, log_floor;
  int n_w;            // CSR weights
  float* window;      // [512] zero padded
  float2* w256;       // [256] e^{-2 pi i j/256}
  float2* w512;       // [257] e^{-2 pi i k/512}
  int* band_start;    // [n_mel]
  int* band_cnt;      // [n_mel]
  int* band_ofs;      // [n_mel]
  float* band_w;      // [n_w]
  float* dct;         // [n_mel * n_out]
  float* dct64;       // [n_mel][64] zero padded
  // stft4 (first radix-16 pass on the matrix pipe): per-lane constants, columns in the order KPERM = 0..7, 9..15, 8
  float* b4;          // [64 lanes][8 k-chunks][2 column tiles] 0.5 * DFT16 entries of the MFMA B operand
  float2* tw4;        // [16 c][16 n2] W256^(n2 * KPERM[c])
  float2* w512p;      // [16 c][8 k2]  W512^(KPERM[c] + 16 k2)
  // stft4 mel stage: band m reads the mel_maxw taps starting at bin mel_ws[m] with weights mel_wpad[m][.] (zero outside
  // the band); mel_maxw = 4 * max(mel_mc) = row stride of mel_wpad (mel_mc[i] = four-tap blocks of the widest band of
  // bands 16 i .. 16 i + 15), 0 when the windows do not fit
  int mel_mc[8];
  int mel_maxw;
  int* mel_ws;        // [n_mel]
  float* mel_wpad;    // [n_mel][mel_maxw]
  float* img4;        // stft4: the constant part of a workgroup's LDS as one image (kws_stft4_prepare), or NULL
};
int kws_stft4_prepare(kws_stft_plan* pl);
int kws_stft4_lds_bytes(const kws_stft_plan* pl);
int kws_stft4_launch(const kws_stft_plan* pl, const float* x, int B, int L, int F, float* out, hipStream_t st);

