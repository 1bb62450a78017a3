// Host-side planners of libkws_hip under AddressSanitizer + UBSan.  The library's sources are built
// HOST-ONLY (hipcc --offload-host-only: kernels become launch stubs that are never called here) with -fsanitize=address,undefined and
// this driver walks everything that does pointer / offset arithmetic on the host without touching a GPU:
//   * the layer tables of all five net kinds (kws_net_create, kws_net_tensor_info): tensors inside their buffers, no overlap;
//   * the workspace layouts behind kws_net_workspace_bytes for awkward batches (1, 3, 70, 384, 1024, 2048), both modes, and every
//     debug view into them (inside the reported size, pairwise disjoint);
//   * the GEMM planners over every pointwise shape of the nets and those batches: kws_gemm_nn_stats_rows, kws_gemm_num_row_tiles,
//     kws_gemm_gather_stats_rows, kws_gemm_tn_workspace_floats (tn_plan: a memoised std::map under a mutex - called from four
//     threads at once), the fp16 x 2 arm's counterparts, kws_dwconv_bwd_part_floats, kws_attn_pool_bwd_workspace_floats.
// Prints "planners ok" and returns 0; any sanitizer report aborts with a non-zero status.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <utility>
#include <vector>

#include "kws_hip.h"

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); \
                                       fprintf(stderr, " (last error: %s)\n", kws_last_error()); exit(1); } } while (0)

static const int BATCHES[] = {1, 3, 70, 384, 1024, 2048};

struct Span { int64_t lo, hi; int what, index; };

static void check_disjoint(std::vector<Span> v, const char* tag, int B) {
  std::sort(v.begin(), v.end(), [](const Span& a, const Span& b) { return a.lo < b.lo; });
  for (size_t i = 1; i < v.size(); ++i)
    CHECK(v[i - 1].hi <= v[i].lo, "%s B=%d: view %d/%d [%lld, %lld) overlaps view %d/%d [%lld, %lld)", tag, B, v[i - 1].what, v[i - 1].index,
          (long long)v[i - 1].lo, (long long)v[i - 1].hi, v[i].what, v[i].index, (long long)v[i].lo, (long long)v[i].hi);
}

static void gemm_planners(int64_t M, int K, int N) {
  const int rows = kws_gemm_nn_stats_rows(M, K, N);
  const int tiles = kws_gemm_num_row_tiles(M);
  CHECK(rows > 0 && rows <= tiles, "nn_stats_rows(%lld,%d,%d) = %d, num_row_tiles = %d", (long long)M, K, N, rows, tiles);
  CHECK(kws_gemm_gather_stats_rows(M) == (int)((M + 127) / 128), "gather_stats_rows(%lld)", (long long)M);
  const int64_t ws = kws_gemm_tn_workspace_floats(M, K, N);
  CHECK(ws >= (int64_t)K * N && ws % ((int64_t)K * N) == 0, "tn_workspace_floats(%lld,%d,%d) = %lld", (long long)M, K, N, (long long)ws);
  CHECK(ws / ((int64_t)K * N) <= 4096, "tn_workspace_floats: %lld slabs", (long long)(ws / ((int64_t)K * N)));
  const int64_t wsh = kws_gemm_tn_f16x2_workspace_floats(M, K, N);
  CHECK(wsh >= 0, "tn_f16x2_workspace_floats");
  CHECK(kws_gemm_nn_f16x2_stats_rows(M) > 0, "nn_f16x2_stats_rows");
}

static void walk_net(int kind, int classes, int fm, int input, int T, int F, const char* tag) {
  kws_net_config_t cfg;
  cfg.kind = kind; cfg.num_classes = classes; cfg.filter_mult = fm; cfg.input_size = input; cfg.spectrogram_length = T; cfg.num_features = F;
  kws_net_t* net = nullptr;
  CHECK(kws_net_create(&cfg, &net) == KWS_OK && net, "%s: net_create", tag);
  const int64_t np = kws_net_num_params(net), ns = kws_net_num_state(net);
  const int nt = kws_net_num_tensors(net);
  CHECK(np > 0 && ns > 0 && nt > 0, "%s: counts", tag);
  std::vector<Span> ps, ss;
  std::vector<std::pair<int, int> > pw;              // (cin, cout) of every 1 x 1 kernel
  for (int i = 0; i < nt; ++i) {
    kws_tensor_info_t ti;
    CHECK(kws_net_tensor_info(net, i, &ti) == KWS_OK, "%s: tensor_info %d", tag, i);
    CHECK(ti.offset >= 0 && ti.size > 0 && ti.offset + ti.size <= (ti.is_state ? ns : np), "%s: tensor %s outside its buffer", tag, ti.name);
    int64_t prod = 1;
    for (int d = 0; d < ti.ndim; ++d) prod *= ti.shape[d];
    CHECK(prod == ti.size && ti.ndim >= 1 && ti.ndim <= 4, "%s: tensor %s shape", tag, ti.name);
    (ti.is_state ? ss : ps).push_back(Span{ti.offset, ti.offset + ti.size, ti.is_state, i});
    if (!ti.is_state && ti.ndim == 3 && ti.shape[0] == 1 && ti.shape[1] % 4 == 0 && ti.shape[2] % 4 == 0)
      pw.push_back(std::make_pair((int)ti.shape[1], (int)ti.shape[2]));
  }
  kws_tensor_info_t bad;
  CHECK(kws_net_tensor_info(net, nt, &bad) != KWS_OK && kws_net_tensor_info(net, -1, &bad) != KWS_OK, "%s: tensor_info range check", tag);
  check_disjoint(ps, tag, 0);
  check_disjoint(ss, tag, 0);
  for (int B : BATCHES) {
    int64_t last = 0;
    for (int training = 0; training < 2; ++training) {
      const int64_t bytes = kws_net_workspace_bytes(net, B, training);
      CHECK(bytes > 0 && bytes % 4 == 0, "%s: workspace_bytes(%d,%d) = %lld", tag, B, training, (long long)bytes);
      CHECK(training == 0 || bytes >= last, "%s: the training workspace is smaller than the inference one", tag);
      last = bytes;
      std::vector<Span> views;
      for (int what = 0; what < 6; ++what)
        for (int index = 0; index < 64; ++index) {
          int64_t off = -1, cnt = -1;
          if (kws_net_debug_view(net, B, training, what, index, &off, &cnt) != KWS_OK) continue;   // (no such view: an error, not a crash)
          CHECK(off >= 0 && cnt > 0 && (off + cnt) * 4 <= bytes, "%s B=%d train=%d: view %d/%d = [%lld, +%lld) outside %lld bytes", tag, B, training,
                what, index, (long long)off, (long long)cnt, (long long)bytes);
          if (what == 0 || what == 1) views.push_back(Span{off, off + cnt, what, index});
        }
      CHECK(!views.empty(), "%s: no debug views", tag);
      if (!training) continue;                       // inference ping-pongs its activations through shared buffers: overlap is the design
      // training keeps every pre-BN / depthwise tensor for the backward pass: distinct tensors never share floats (a view listed
      // under two indices is the same span twice)
      std::sort(views.begin(), views.end(), [](const Span& a, const Span& b) { return a.lo < b.lo || (a.lo == b.lo && a.hi < b.hi); });
      views.erase(std::unique(views.begin(), views.end(), [](const Span& a, const Span& b) { return a.lo == b.lo && a.hi == b.hi; }), views.end());
      check_disjoint(views, tag, B);
    }
    CHECK(kws_net_workspace_bytes(net, 0, 1) == 0 && kws_net_workspace_bytes(nullptr, B, 1) == 0, "%s: workspace_bytes argument checks", tag);
    // the GEMM planners on this net's pointwise shapes at this batch (rows per clip: the chain's lengths for a 1 s clip)
    static const int LENS[] = {399, 397, 199, 197, 99, 97, 96, 49, 48, 47, 24, 22, 12, 11, 9};
    for (const auto& s : pw)
      for (int L : LENS) gemm_planners((int64_t)B * L, s.first, s.second);
  }
  CHECK(kws_net_destroy(net) == KWS_OK, "%s: destroy", tag);
}

int main() {
  CHECK(kws_abi_version() == KWS_ABI_VERSION, "abi");
  walk_net(KWS_NET_TS_ATTENTION, 12, 1, 16000, 0, 0, "ts_attention/12");
  walk_net(KWS_NET_TS_ATTENTION, 32, 2, 16000, 0, 0, "ts_attention/32 x2");
  walk_net(KWS_NET_LOG_MFCC, 32, 1, 98 * 40, 98, 40, "log_mfcc/32 98x40");
  walk_net(KWS_NET_LOG_MFCC, 12, 1, 65 * 40, 65, 40, "log_mfcc/12 65x40");
  walk_net(KWS_NET_STEFFE, 12, 1, 16000, 0, 0, "steffe");
  walk_net(KWS_NET_RESIDUAL, 12, 1, 16000, 0, 0, "residual");
  walk_net(KWS_NET_MFCC_AND_RAW, 12, 1, 98 * 40 + 16000, 98, 40, "mfcc_and_raw");
  // a config that must be refused, not walked
  kws_net_config_t cfg = {99, 12, 1, 16000, 0, 0};
  kws_net_t* net = nullptr;
  CHECK(kws_net_create(&cfg, &net) != KWS_OK && net == nullptr, "unknown kind accepted");
  // tn_plan's memo (std::map under a mutex) from four threads at once, fresh shapes
  std::vector<std::thread> th;
  for (int t = 0; t < 4; ++t)
    th.emplace_back([t]() {
      for (int i = 0; i < 200; ++i) {
        const int64_t M = 1000 + 37 * i + t;
        const int K = 64 * (1 + i % 8), N = 64 * (1 + (i / 8) % 8);
        const int64_t ws = kws_gemm_tn_workspace_floats(M, K, N);
        if (ws < (int64_t)K * N || ws % ((int64_t)K * N) != 0) { fprintf(stderr, "tn plan (threads) M=%lld K=%d N=%d\n", (long long)M, K, N); exit(1); }
      }
    });
  for (auto& t : th) t.join();
  for (int B : BATCHES)
    for (int C : {64, 128, 192, 256, 320, 384, 512, 1024})
      for (int L : {799, 399, 98, 49, 12}) {
        const int64_t pf = kws_dwconv_bwd_part_floats(B, L, C);
        CHECK(pf > 0 && pf % (5 * C) == 0, "dwconv_bwd_part_floats(%d,%d,%d) = %lld", B, L, C, (long long)pf);
      }
  CHECK(kws_attn_pool_bwd_workspace_floats(1024, 9, 512) > 0, "attn_pool_bwd_workspace_floats");
  printf("planners ok\n");
  return 0;
}
