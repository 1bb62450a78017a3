/* libkws_hip.so - C ABI of the MI355X (gfx950) keyword-spotting hot path.
 *
 * The reference (see--/speech_recognition) has no FFI of its own: its hot path is Python that
 * lowers to TensorFlow-1.4 / Keras-2.1.2 ops.  Each entry point below replaces the
 * ops behind one reference call site, cited as file:line under the reference checkout.
 *
 * Conventions
 *  - plain C, no exceptions across the ABI; every function returns 0 (KWS_OK) or a negative
 *    KWS_E_* code; kws_last_error() returns a thread-local message for the last failure.
 *  - every tensor argument is a raw DEVICE pointer owned by the caller (e.g. the data_ptr() of a
 *    PyTorch-ROCm tensor); the library never frees or retains it.  Exceptions are the *_create
 *    functions, whose table arguments are HOST pointers copied once into a plan.
 *  - work is enqueued on `stream` (a hipStream_t passed as void*) and not synchronised.
 *  - layout is channels-last row-major fp32 [B, L, C] (= Keras channels_last), so Keras-named
 *    weights load without transposes: conv1d/kernel [k, Cin, Cout], depthwise_kernel [1,3,C,1],
 *    dense/kernel [in, out].
 *  - functions are re-entrant: no global mutable state besides the thread-local error string and the
 *    thread-local profiler attachment; switches (profiling, GEMM arithmetic arm) live on handles.
 */
#ifndef KWS_HIP_H_
#define KWS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libkws_hip.so is built with -fvisibility=hidden: exactly the declarations of this header are exported
 * (tests/test_abi.py compares `nm -D` with it) */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define KWS_OK 0
#define KWS_E_INVALID (-1)   /* bad argument / unsupported shape */
#define KWS_E_HIP (-2)       /* a HIP runtime call failed */
#define KWS_E_WORKSPACE (-3) /* caller workspace too small */

#define KWS_ABI_VERSION 5   /* round 6: round 5's measured loss (kws_wgrad_items_t, kws_dwconv_bwd_bn_wgrad_f32, kws_gemm_tn_items, kws_gemm_tn_ckpt_floats, gemm mode 3) retired from the ABI - kept as scripts/probes/wgrad_beside_dwbwd/mode3.patch; round 5 (4): a refused profiler_destroy changes nothing; round 4 (3): hidden visibility (exports = this header), the three |x|-maximum producers, gemm mode 1, profiler_destroy may refuse; round 3 (2): profiler / gemm-mode state moved onto handles */

int kws_abi_version(void);
const char* kws_last_error(void);
/* name of the device the calling thread is bound to; "" if no HIP device is usable */
int kws_device_name(char* buf, int cap);

/* A HIP stream in a scheduling class (cls: -1 lowest, 0 normal, +1 highest priority of the device's range).  The
 * batch generator runs on a LOW-priority stream so that its kernels fill the CUs the training stream leaves idle
 * instead of co-running with its MFMA kernels (PyTorch itself can only create normal / high priority streams). */
int kws_stream_create(int cls, void** stream);
int kws_stream_destroy(void* stream);

/* Optional per-kernel-family profiler (measurement only).  A profiler is a HANDLE; kws_profiler_attach(p) makes the
 * CALLING THREAD record into p (NULL detaches): while attached, every launcher called from that thread brackets its
 * launch with a hipEvent pair on the launch stream and books the algorithmic FLOPs/bytes of the call.  Several threads
 * may attach the same handle (the batch generator thread and the training thread of bench.py do).
 * kws_profiler_collect() waits for the recorded events and returns the number of families, kws_profiler_get() reads one
 * (summed device ms, launches, FLOPs, bytes).  No process-wide switch: a thread that never attaches never records.
 * The handle counts its attached threads (a thread that exits detaches itself): kws_profiler_destroy() returns KWS_E_INVALID
 * and changes NOTHING - the handle stays alive, the calling thread stays attached - while any OTHER thread is still attached;
 * when it succeeds it ends the calling thread's own attachment with the handle. */
typedef struct kws_profiler kws_profiler_t;
int kws_profiler_create(kws_profiler_t** out);
int kws_profiler_destroy(kws_profiler_t* p);
int kws_profiler_attach(kws_profiler_t* p);
int kws_profiler_collect(kws_profiler_t* p);
int kws_profiler_get(kws_profiler_t* p, int idx, char* name, int cap, double* ms, int64_t* count, double* flops,
                     double* bytes);

/* ------------------------------------------------------------------------------------------
 * a2  augment graph: decode_wav -> multiply -> tf_roll -> multiply/add -> reshape
 *     reference input_data.py:334-359, utils.py:56-73
 *   out[b,t] = bg_vol[b] * noise[noise_off[b] + t] + fg_vol[b] * bank[clip_idx[b], (t - shift[b]) mod L]
 * bank: [n_clips, L] resident clip bank (f32, or int16 PCM scaled by 1/32768 like DecodeWav).
 * noise: 1-D concatenation of the background recordings (input_data.py:274-309); noise_off are
 * absolute start samples (input_data.py:484-487); noise may be NULL when every bg_vol is 0.
 * ---------------------------------------------------------------------------------------- */
int kws_augment_f32(const float* bank, int64_t n_clips, int L, const int32_t* clip_idx,
                    const float* fg_vol, const int32_t* shift, const float* noise,
                    int64_t noise_len, const int64_t* noise_off, const float* bg_vol,
                    float* out, int B, void* stream);
int kws_augment_i16(const int16_t* bank, int64_t n_clips, int L, const int32_t* clip_idx,
                    const float* fg_vol, const int32_t* shift, const float* noise,
                    int64_t noise_len, const int64_t* noise_off, const float* bg_vol,
                    float* out, int B, void* stream);

/* a1  host-side sampler: the per-clip RNG draw loop of AudioProcessor.get_data, reference
 * input_data.py:457-514 (draw order: SURVEY Appendix C).  Pure host code (no GPU work): consumes the
 * NumPy legacy MT19937 stream passed in (key[624], pos as in np.random.get_state()) and advances it,
 * so seeded runs reproduce the reference's choices.  Outputs are the per-clip parameters of
 * kws_augment_* plus the label indices. */
typedef struct {
  const int32_t* rows;    /* clip-bank row of every entry of the partition */
  const int32_t* labels;  /* label index (word_to_index) */
  const uint8_t* silence; /* 1 where the entry is a _silence_ clip */
  int32_t n;
} kws_sampler_set_t;
typedef struct {
  int32_t deterministic;  /* 1: entries offset..offset+count-1 in order (how_many == -1 or mode != training) */
  int32_t offset, count;
  int32_t use_background; /* background data present and mode == training */
  int32_t n_bg;
  const int64_t* bg_len;   /* samples per background recording */
  const int64_t* bg_start; /* start of each recording inside the concatenated noise vector */
  int32_t desired_samples;
  int32_t shift_lo, shift_hi; /* time_shift_range (inclusive) */
  double background_frequency, background_volume_range, foreground_frequency, foreground_volume_range;
  double time_shift_frequency, pseudo_frequency, flip_frequency, silence_volume_range;
} kws_sampler_args_t;
int kws_sampler_draw(uint32_t* mt_key, int* mt_pos, const kws_sampler_set_t* cand,
                     const kws_sampler_set_t* pseudo, const kws_sampler_args_t* args,
                     int32_t* out_rows, int32_t* out_labels, int32_t* out_shift, int64_t* out_bg_off,
                     float* out_bg_vol, float* out_fg_vol);

/* a17 TTA transforms, reference make_submission.py:125-134.
 * kind: 0 copy, 1 np.roll(X,-1500,axis=1), 2 1.2*X, 3 clip(1.1*X,-1,1), 4 0.9*X */
int kws_tta_transform(const float* x, float* out, int B, int L, int kind, void* stream);
/* mean of n_terms probability tensors / divisor, argmax; make_submission.py:137-146 */
int kws_tta_combine(const float* const* probs, int n_terms, float divisor, float* out_probs,
                    int32_t* out_argmax, int B, int C, void* stream);

/* Speed-TTA time stretch: reference create_tta_set.py:9-22 -
 * librosa.effects.time_stretch(np.float32(pcm) / 32767, rate)[-keep:] -> np.int16(. * 32767) -> wav, read back
 * by make_submission.py:86-100 through DecodeWav (/ 32768) for the slow predict passes (:133-136).
 * librosa 0.5.x defaults (STFT 2048 / 512, periodic Hann, centred reflect padding, phase vocoder, ISTFT with
 * window-sum-square normalisation trimmed by 1024 each side); the stretched signal has
 * kws_stretch_out_samples() = 512 * (ceil((1 + n_samples / 512) / rate) - 1) samples.
 *   x   [B, n_samples] f32 (multiplied by in_scale on load) or int16 (divided by 32767 like the reference)
 *   out [B, keep]: the LAST `keep` stretched samples; zero padded at the end when fewer exist
 *   quantize != 0 reproduces the int16 wav round trip: (int16)(v * 32767) / 32768 (C-cast truncation)
 * rate <= 0 is KWS_E_INVALID (librosa raises ParameterError). */
typedef struct kws_stretch_plan kws_stretch_plan_t;
int kws_stretch_plan_create(int n_samples, double rate, kws_stretch_plan_t** plan);
int kws_stretch_plan_destroy(kws_stretch_plan_t* plan);
int kws_stretch_out_samples(const kws_stretch_plan_t* plan);
int kws_time_stretch_f32(const kws_stretch_plan_t* plan, const float* x, float in_scale, float* out, int B,
                         int keep, int quantize, void* stream);
int kws_time_stretch_i16(const kws_stretch_plan_t* plan, const int16_t* x, float* out, int B, int keep,
                         int quantize, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stand-alone forms of the classifier-tail ops (the network programs below run them fused in one
 * kernel; these bind ONE reference call site each).
 *
 * keras Dropout, reference model.py:819,828 (mask = floor(keep_prob + U[0,1)), x / keep_prob).
 * Counter-based: element i of row r keeps its value iff fmix32(((row_offset + r) * n + i) * 0x9E3779B1 + key) <
 * keep_prob * 2^32, key = f(seed, step, layer_id) - the masks the network programs and the oracle use
 * (layer_id 1 = dropout_1, 2 = dropout_2).  bwd = the same mask applied to the incoming gradient.
 * ---------------------------------------------------------------------------------------- */
int kws_dropout_fwd(const float* x, float* out, int B, int n, float keep_prob, uint64_t seed,
                    uint32_t step, uint32_t layer_id, int64_t row_offset, void* stream);
int kws_dropout_bwd(const float* dy, float* dx, int B, int n, float keep_prob, uint64_t seed,
                    uint32_t step, uint32_t layer_id, int64_t row_offset, void* stream);

/* a12 attention pooling, reference model.py:824-827:
 *   feat[b] = [ max_t(x[b,t,:] * att[b,t]) ; mean_t x[b,t,:] ]        x [B,T,C], att [B,T], feat [B,2C]
 * bwd: dfeat [B,2C] -> dx [B,T,C] (gradient wrt x through both branches) and datt [B,T]; reduce_max
 * splits its gradient equally among ties (TF _MinOrMaxGrad).  workspace: kws_attn_pool_bwd_workspace_floats. */
int kws_attn_pool_fwd(const float* x, const float* att, float* feat, int B, int T, int C, void* stream);
int64_t kws_attn_pool_bwd_workspace_floats(int B, int T, int C);
int kws_attn_pool_bwd(const float* x, const float* att, const float* dfeat, float* dx, float* datt,
                      float* workspace, int B, int T, int C, void* stream);

/* a13 smooth_categorical_crossentropy, reference utils.py:87-108 as used at model.py:835-836:
 *   loss[b] = softmax_cross_entropy(labels = y (1 - s) + s / NC, logits = log(clip(p, 1e-7, 1 - 1e-7)))
 * probs/labels [B,NC] (NC <= 64); per_correct (may be NULL) = 1 where argmax p == argmax y.
 * bwd: dprobs = dL/dp * inv_loss_batch (zero where the clip is active), dlogits = that gradient carried
 * through the softmax that produced p; either output may be NULL. */
int kws_softmax_xent_smooth_fwd(const float* probs, const float* labels, float* per_loss,
                                float* per_correct, int B, int NC, float label_smoothing, void* stream);
int kws_softmax_xent_smooth_bwd(const float* probs, const float* labels, float* dprobs, float* dlogits,
                                int B, int NC, float label_smoothing, float inv_loss_batch, void* stream);

/* Gradient exchange of the data-parallel step (the reference is single-session, train.py:24-26):
 * an RCCL communicator from a (rank, world, 128-byte unique id) triple - rank 0 calls kws_comm_unique_id and
 * hands the bytes to the other ranks by any side channel - and the in-place sum of the flat gradient buffer
 * over xGMI, enqueued on `stream`.  librccl is resolved at run time (dlopen); a process that never creates a
 * communicator never loads it. */
typedef struct kws_comm kws_comm_t;
int kws_comm_unique_id(void* id128);
int kws_comm_create(int rank, int world, const void* id128, kws_comm_t** comm);
int kws_comm_destroy(kws_comm_t* comm);
int kws_allreduce_grads(kws_comm_t* comm, float* grads, int64_t n, void* stream);

/* a18 32->12 head, reference freeze_graph_32_classes.py:55-69.
 * map[i] in [0,12): output slot of input class i (slot 1 = max over all classes mapped to 1). */
int kws_head32to12(const float* p_in, int C_in, const int32_t* map, int C_out, float* p_out,
                   int B, void* stream);

/* ------------------------------------------------------------------------------------------
 * a3-a5  STFT -> |X| -> mel -> log -> DCT   (one table-driven kernel for both feature paths)
 *     path B: reference input_data.py:361-381 (tf.contrib.signal.stft, abs, tensordot mel,
 *             log(+1e-6), mfccs_from_log_mel_spectrograms[..., :K])
 *     path A: reference audio.py:15-23 (audio_spectrogram magnitude_squared -> mfcc)
 * Host tables (copied into the plan): window[frame_len], mel[n_bins * n_mel] row-major
 * (n_bins = fft_len/2+1), dct[n_mel * n_out] row-major.  fft_len must be 512.
 * out_kind: 0 = DCT features [B,F,n_out] (mfcc_), 1 = magnitude spectrogram [B,F,257]
 * (spectrogram_, input_data.py:366), 2 = log-mel [B,F,n_mel].
 * ---------------------------------------------------------------------------------------- */
typedef struct kws_stft_plan kws_stft_plan_t;
int kws_stft_plan_create(int frame_len, int frame_step, int fft_len, int n_mel, int n_out,
                         const float* window, const float* mel, const float* dct,
                         float log_offset, float log_floor, kws_stft_plan_t** plan);
int kws_stft_plan_destroy(kws_stft_plan_t* plan);
int kws_stft_num_frames(const kws_stft_plan_t* plan, int L);
int kws_stft_mel_f32(const kws_stft_plan_t* plan, const float* x, int B, int L, float* out,
                     int out_kind, void* stream);

/* A/B arm, off by default (csrc/gemm_f16x2.hip; selected per net handle by kws_net_set_gemm_mode(net, 2)): the pointwise
 * GEMMs with every f32 operand scaled by a power of two and split into TWO fp16 parts, three f16 MFMA products
 * accumulated in f32 - as accurate as the f32 matrix pipe, not bit-identical to it.  The scale of an operand
 * comes from its |x| maximum, kept on the device in a "slot group" of 256 words (atomicMax of the bit patterns, so it
 * is the same in every run): kws_absmax_batch_f32 fills groups for arbitrary tensors; inside the network the kernels
 * that produce a GEMM operand leave its maximum behind.  The largest magnitude lands in [2^14, 2^15) (fp16 overflows at
 * 65504) and results are multiplied by the two inverse scales on the way out (exact: powers of two). */
int kws_absmax_batch_f32(const float* const* in, const int64_t* n, unsigned* slots, int count, void* stream);
/* The producers of a GEMM operand that leave the operand's |x| maximum behind on the way (what the network programs call for this
 * arm; `amax` = a slot group of 256 words zeroed by the caller, or NULL = the plain kernel): kws_dwconv_fwd_f32 / pass 2 of
 * kws_dwconv_bwd_bn_f32 / kws_bn_bwd_apply with one more argument.  The tensors they write are bit-identical to the plain calls'. */
int kws_dwconv_fwd_amax_f32(const float* y, const float* bn, const float* w, float* z, int B, int L_in, int L_out,
                            int C, int stride, int pad_l, unsigned* amax, void* stream);
int kws_dwconv_bwd_bn_amax_f32(const float* dz, const float* y, const float* bn, const float* w, const float* coef,
                               float* dy, float* part, int pass, int B, int L_in, int L_out, int C, int stride, int pad_l,
                               unsigned* amax, void* stream);
int kws_bn_bwd_apply_amax(float* g, const float* y, const float* bn, const float* gamma, const float* coef, int64_t rows,
                          int C, unsigned* amax, void* stream);
int kws_f16x2_split_batch(const float* const* in, void* const* out, const int* rows, const int* cols,
                          const int* transpose, const unsigned* const* slots, int count, void* stream);
/* shapes the arm's kernels take (K granule, 32-bit offsets inside a 2 GB buffer view); the network programs fall back
 * to the f32 kernels for anything else.  stats rows of the NN kernel: one per 128-row tile. */
int kws_gemm_nn_f16x2_supported(int64_t M, int K, int N);
int kws_gemm_tn_f16x2_supported(int64_t M, int K, int N);
int kws_gemm_nn_f16x2_stats_rows(int64_t M);
int kws_gemm_nn_f16x2_f32(const float* A, const void* Bp, float* C, int64_t M, int K, int N,
                          const unsigned* a_slots, const unsigned* b_slots, float* stats_part, void* stream);
int64_t kws_gemm_tn_f16x2_workspace_floats(int64_t M, int K, int N);
int kws_gemm_tn_f16x2_f32(const float* Z, const float* G, float* dW, int64_t M, int K, int N,
                          const unsigned* z_slots, const unsigned* g_slots, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * a7+a8, a10  GEMM family on f32 MFMA (v_mfma_f32_32x32x2_f32)
 *   C[M,N] = A[M,K] * W[K,N]          pointwise Conv1D(1x1)  reference model.py:48-49
 *   with a gathered A it is frame+Conv1D(k3,s2) (model.py:805-808) / Conv1D(64,3) (model.py:1450)
 *   / the stride-2 1x1 shortcut (model.py:1431-1432):
 *   A[(b,t), j*cin + c] = X[b*x_batch_stride + t*stride_t + j*stride_j + c + base_off], 0 outside
 *   [0, x_len) of clip b.
 * stats (optional, may be NULL): partial column sums [rows][2][N] (sum x, sum x^2) for BatchNorm, finalised
 * by kws_bn_stats_finalize (a11).  The buffer must hold 2 * kws_gemm_num_row_tiles(M) * N floats (an upper
 * bound); the number of rows a call actually writes is kws_gemm_nn_stats_rows(M, K, N) for kws_gemm_nn_f32
 * (one row per workgroup of the persistent kernel, <= 256) and kws_gemm_gather_stats_rows(M) for
 * kws_gemm_gather_f32 (one row per 128-row tile) - pass that count to kws_bn_stats_finalize.
 * ---------------------------------------------------------------------------------------- */
int kws_gemm_num_row_tiles(int64_t M);
int kws_gemm_nn_stats_rows(int64_t M, int K, int N);
int kws_gemm_gather_stats_rows(int64_t M);
int kws_gemm_nn_f32(const float* A, const float* W, float* C, int64_t M, int K, int N,
                    float* stats_part, void* stream);
typedef struct {
  int L_out;             /* rows per clip */
  int cin;               /* channels per tap */
  int taps;              /* K = taps * cin */
  int stride_t;          /* element stride between consecutive output rows */
  int stride_j;          /* element stride between taps */
  int base_off;          /* offset of (t=0, j=0, c=0), may be negative (SAME padding) */
  int x_len;             /* valid elements per clip */
  int64_t x_batch_stride;
} kws_gather_t;
int kws_gemm_gather_f32(const float* X, const kws_gather_t* g, const float* W, float* C, int B,
                        int N, float* stats_part, void* stream);
/* dW[K,N] = A^T[K,M] * G[M,N] (Conv2DBackpropFilter of the 1x1 conv); deterministic split-M:
 * workspace floats >= kws_gemm_tn_workspace_floats(M,K,N). */
int64_t kws_gemm_tn_workspace_floats(int64_t M, int K, int N);
int kws_gemm_tn_f32(const float* A, const float* G, float* dW, int64_t M, int K, int N,
                    float* workspace, void* stream);
int kws_gemm_tn_gather_f32(const float* X, const kws_gather_t* g, const float* G, float* dW, int B,
                           int N, float* workspace, void* stream);
int kws_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * a11  BatchNormalization (training: biased batch moments over (B,L); eps 1e-3; momentum .99)
 *      + Activation(relu6), reference model.py:46-51, 809-810; constants SURVEY D.2.
 * The normalise+ReLU6 is never materialised: it is applied on load by the consumer through the
 * per-channel (scale, shift) these functions produce.
 * bn layout: float[4*C] = scale | shift | mean | rstd.
 * ---------------------------------------------------------------------------------------- */
/* scratch (optional): KWS_REDUCE_SLICES * 2 * C floats (5 * C for kws_dw_bwd_finalize); when given,
 * long partial lists are folded in two fixed-order stages instead of one serial pass. */
#define KWS_REDUCE_SLICES 32
int kws_bn_stats_finalize(const float* stats_part, int n_tiles, int64_t count, int C,
                          const float* gamma, const float* beta, float eps, float momentum,
                          float* moving_mean, float* moving_var, float* bn, float* scratch,
                          void* stream);
int kws_bn_infer_prepare(const float* gamma, const float* beta, const float* moving_mean,
                         const float* moving_var, float eps, int C, float* bn, void* stream);
/* elementwise y -> relu6(scale*y+shift): only used by tests and by inference outputs */
int kws_bn_relu6_apply(const float* y, const float* bn, float* out, int64_t rows, int C, int relu6,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * a9  DepthwiseConv2D((1,3)) on [B,1,L,C], reference model.py:34-44, with the producer's
 *     BN+ReLU6 applied on load (bn may be NULL: input used as is).
 *   z[b,t,c] = sum_j w[j,c] * act(y[b, s*t + j - pad_l, c])      (0 outside [0,L_in))
 * bwd (DepthwiseConv2dNativeBackpropInput/Filter + ReluGrad + the BatchNorm reduction, a15):
 *   g[b,u,c]  = relu6'(.) * sum_j w[j,c] dz[b,(u+pad_l-j)/s,c]
 *   part      = per-block partial sums of (g, g*xhat, dw0, dw1, dw2), finalised by kws_dw_bwd_finalize.
 * ---------------------------------------------------------------------------------------- */
int kws_dwconv_fwd_f32(const float* y, const float* bn, const float* w, float* z, int B, int L_in,
                       int L_out, int C, int stride, int pad_l, void* stream);
int64_t kws_dwconv_bwd_part_floats(int B, int L_in, int C);
int kws_dwconv_bwd_f32(const float* dz, const float* y, const float* bn, const float* w, float* g,
                       float* part, int B, int L_in, int L_out, int C, int stride, int pad_l,
                       void* stream);
/* The same backward with the consumer's BatchNorm backward fused in, without materialising g
 * (reference: the BatchNormalization + relu6 + DepthwiseConv2D backward chain of model.py:34-51):
 *   pass 1: part only (fold with kws_dw_bwd_finalize -> dw, dgamma, dbeta and coef = [c1 | c2]);
 *   pass 2: dy[b,u,c] = scale[c] * (g - c1[c] - xhat * c2[c]), g recomputed from the same operands
 *           (bit-identical to kws_dwconv_bwd_f32 followed by kws_bn_bwd_apply).  bn must not be NULL. */
int kws_dwconv_bwd_bn_f32(const float* dz, const float* y, const float* bn, const float* w,
                          const float* coef, float* dy, float* part, int pass, int B, int L_in,
                          int L_out, int C, int stride, int pad_l, void* stream);
/* reduces part -> dw[3,C] (may be NULL), dgamma[C], dbeta[C], and coef[2*C] = (c1, c2) used by
 * kws_bn_bwd_apply; n_parts = part floats / (5*C) */
int kws_dw_bwd_finalize(const float* part, int n_parts, int64_t count, int C, float* dw,
                        float* dgamma, float* dbeta, float* coef, float* scratch, void* stream);
/* dy = gamma*rstd*(g - c1 - xhat*c2), in place on g (BatchNorm backward through batch stats) */
int kws_bn_bwd_apply(float* g, const float* y, const float* bn, const float* gamma,
                     const float* coef, int64_t rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * a14  optimizers on one flat parameter buffer, reference model.py:834 (RMSprop(lr=1e-3)) and
 *      model.py:96,110 (SGD momentum); constants SURVEY D.5.  g_eff = grad*grad_scale + 2*l2[i]*p
 *      (l2[i] = per-element kernel_regularizer coefficient, 0 for BN/bias).
 * ---------------------------------------------------------------------------------------- */
int kws_rmsprop_step(float* p, const float* grad, float* acc, const float* l2, int64_t n, float lr,
                     float rho, float eps, float grad_scale, void* stream);
int kws_sgd_momentum_step(float* p, const float* grad, float* vel, const float* l2, int64_t n,
                          float lr, float momentum, float grad_scale, void* stream);
/* out[0] = sum_i l2[i]*p[i]^2 (Keras regularisation loss) */
int kws_l2_loss(const float* p, const float* l2, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Network programs: the whole forward / forward+backward of one model as a sequence of the
 * kernels above, launched natively (no Python between layers).
 *   KWS_NET_TS_ATTENTION: conv_1d_time_sliced_with_attention_model, reference model.py:775-838
 *   KWS_NET_LOG_MFCC:     conv_1d_log_mfcc_model, reference model.py:1400-1479 (and conv_1d_spectrogram_model,
 *                         model.py:1482-1561: num_features = 257)
 *   KWS_NET_STEFFE:       steffeNet, reference model.py:1663-1726 (raw input; input_size and num_classes only)
 *   KWS_NET_RESIDUAL:     conv_1d_residual_model, reference model.py:841-908 (raw input; filter_mult honoured)
 *   KWS_NET_MFCC_AND_RAW: conv_1d_mfcc_and_raw_model, reference model.py:1563-1660; the two Keras inputs arrive as ONE
 *                         row [mfcc spectrogram_length*num_features | raw samples], input_size = the row length
 * The net handle holds only the host-side layer table.  Parameters live in ONE flat f32 buffer
 * (trainable, Keras layer order) + one flat state buffer (BN moving mean/variance), both owned by
 * the caller; kws_net_tensor_info enumerates the Keras-named tensors inside them.
 * ---------------------------------------------------------------------------------------- */
#define KWS_NET_TS_ATTENTION 1
#define KWS_NET_LOG_MFCC 2
#define KWS_NET_STEFFE 3
#define KWS_NET_RESIDUAL 4
#define KWS_NET_MFCC_AND_RAW 5
typedef struct kws_net kws_net_t;
typedef struct {
  int kind;
  int num_classes;
  int filter_mult;        /* TS_ATTENTION only */
  int input_size;         /* 16000 (raw) or spectrogram_length*num_features */
  int spectrogram_length; /* LOG_MFCC only */
  int num_features;       /* LOG_MFCC only */
} kws_net_config_t;
typedef struct {
  char name[64];
  int64_t offset; /* float offset inside the params (is_state=0) or state (is_state=1) buffer */
  int64_t size;
  int ndim;
  int64_t shape[4];
  int is_state;
  float l2;       /* kernel_regularizer coefficient (0 if none) */
  int fan_in, fan_out; /* Glorot fans, 0 for non-kernels */
  float init;     /* constant initial value for non-kernels */
} kws_tensor_info_t;

int kws_net_create(const kws_net_config_t* cfg, kws_net_t** net);
int kws_net_destroy(kws_net_t* net);
/* arithmetic / launch schedule of this handle's pointwise GEMMs: 0 = f32 MFMA (default, the product path; since round 4 a
 * layer's input-gradient and weight-gradient GEMMs go out as ONE launch), 1 = f32 MFMA with the two as separate launches (the
 * schedule of rounds 1 - 3, kept as the A/B reference: bit-identical results; every net kind), 2 = the fp16 x 2 A/B arm (raw-waveform
 * attention net; other kinds ignore it).  State of the handle, not of the process. */
int kws_net_get_gemm_mode(const kws_net_t* net);
int kws_net_set_gemm_mode(kws_net_t* net, int mode);
int64_t kws_net_num_params(const kws_net_t* net);
int64_t kws_net_num_state(const kws_net_t* net);
int kws_net_num_tensors(const kws_net_t* net);
int kws_net_tensor_info(const kws_net_t* net, int idx, kws_tensor_info_t* info);
int64_t kws_net_workspace_bytes(const kws_net_t* net, int max_batch, int training);
/* parity/debug view into a workspace laid out for (batch, training): what = 0 pre-BN conv output
 * y[index] (0 = conv1d_1 .. 11 = conv1d_12), 1 depthwise output z[index], 2 BN table of
 * batch_normalization_{index+1} (scale|shift|mean|rstd), 3 attention weights [B,T] (training). */
int kws_net_debug_view(const kws_net_t* net, int batch, int training, int what, int index,
                       int64_t* offset_floats, int64_t* count);
/* inference (K.learning_phase()=0): moving statistics, no dropout. probs [B, num_classes] */
int kws_net_predict(const kws_net_t* net, const float* params, const float* state, const float* x,
                    int B, float* probs, void* workspace, int64_t workspace_bytes, void* stream);
/* one train_on_batch minus the optimizer: forward with batch statistics + dropout, loss
 * (a13: utils.py:87-108 for TS_ATTENTION, categorical_crossentropy for LOG_MFCC), backward,
 * BN moving-average update of `state`.  grads: flat buffer laid out like params (data-loss
 * gradient only, scaled by 1/loss_batch; L2 is folded into the optimizer).
 * metrics (device float[4]): sum of per-sample data loss, number of correct argmax, 0, 0.
 * row_offset: global index of row 0 (dropout counter offset for data-parallel shards). */
int kws_net_train_fwd_bwd(const kws_net_t* net, const float* params, float* state, const float* x,
                          const float* y_onehot, int B, float* grads, float* probs, float* metrics,
                          uint64_t seed, uint32_t step, int64_t row_offset, int loss_batch,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* The same step in two calls, so that the caller can start the gradient all-reduce of the LATE layers while the early
 * layers' backward still runs ("one fused buffer, overlapped with the tail of backward"; raw-waveform
 * attention net only):
 *   part 1  forward, classifier tail, backward of blocks n_blocks-1 .. split_block;
 *   part 2  backward of blocks split_block-1 .. 0 and of the first convolution.
 * After part 1 every gradient from float offset kws_net_grad_ready_offset(net, split_block) to the end of the flat
 * buffer is final.  Parts 1 + 2 enqueue exactly the launches of kws_net_train_fwd_bwd in the same order: the result is
 * bit-identical.  1 <= split_block < kws_net_num_blocks(net). */
int kws_net_num_blocks(const kws_net_t* net);
int64_t kws_net_grad_ready_offset(const kws_net_t* net, int split_block);
int kws_net_train_fwd_bwd_part(const kws_net_t* net, const float* params, float* state, const float* x,
                               const float* y_onehot, int B, float* grads, float* probs, float* metrics,
                               uint64_t seed, uint32_t step, int64_t row_offset, int loss_batch,
                               void* workspace, int64_t workspace_bytes, int part, int split_block,
                               void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* KWS_HIP_H_ */
