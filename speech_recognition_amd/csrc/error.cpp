// Error reporting + library identity for libkws_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/kws_hip.h"

static thread_local char g_err[512] = "";

void kws_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int kws_abi_version(void) { return KWS_ABI_VERSION; }

const char* kws_last_error(void) { return g_err; }

int kws_device_name(char* buf, int cap) {
  if (!buf || cap <= 0) return KWS_E_INVALID;
  buf[0] = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    kws_set_error("no HIP device");
    return KWS_E_HIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    kws_set_error("hipGetDeviceProperties failed");
    return KWS_E_HIP;
  }
  snprintf(buf, cap, "%s", prop.gcnArchName);
  return KWS_OK;
}

// Streams with a scheduling class.  PyTorch can only make normal / high priority streams; the batch generator wants
// a LOW one, so that its augment / STFT kernels take the CUs the training stream leaves idle instead of competing
// with its MFMA kernels (they co-ran with conv1_* and lost both ways).  cls: -1 low, 0 normal, +1 high.
int kws_stream_create(int cls, void** stream) {
  if (!stream) {
    kws_set_error("kws_stream_create: NULL output");
    return KWS_E_INVALID;
  }
  int least = 0, greatest = 0;   // numerically: least priority is the LARGER number
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
    kws_set_error("hipDeviceGetStreamPriorityRange failed");
    return KWS_E_HIP;
  }
  const int prio = cls < 0 ? least : (cls > 0 ? greatest : (least + greatest) / 2);
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
  if (e != hipSuccess) {
    kws_set_error("hipStreamCreateWithPriority(%d) failed: %s", prio, hipGetErrorString(e));
    return KWS_E_HIP;
  }
  *stream = s;
  return KWS_OK;
}

int kws_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    kws_set_error("hipStreamDestroy failed");
    return KWS_E_HIP;
  }
  return KWS_OK;
}

}  // extern "C"

// ---- profiler ----------------------------------------------------------------------------------------
#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct ProfRec {
  hipEvent_t a, b;
  std::string name;
  double flops, bytes;
};
struct ProfAgg {
  double ms = 0, flops = 0, bytes = 0;
  long long count = 0;
};
}  // namespace
// A profiler is a handle (include/kws_hip.h); the calling thread's attachment is the only state outside it.
struct kws_profiler {
  std::mutex mu;
  std::atomic<int> attached{0};   // threads recording into this handle: destroy refuses while it is non-zero
  std::vector<ProfRec*> recs;
  std::vector<std::pair<std::string, ProfAgg>> out;
  void clear_locked() {
    for (ProfRec* r : recs) {
      (void)hipEventDestroy(r->a);
      (void)hipEventDestroy(r->b);
      delete r;
    }
    recs.clear();
  }
};
// the calling thread's attachment; a thread that ends while attached detaches itself (its count must not pin the handle)
struct ProfTls {
  kws_profiler* p = nullptr;
  ~ProfTls() {
    if (p) p->attached.fetch_sub(1);
  }
};
static thread_local ProfTls t_prof_tls;
#define t_prof (t_prof_tls.p)
static std::mutex g_roctx_mu;

bool kws_prof_on() { return t_prof != nullptr; }

// ---- roctx ranges (KWS_ROCTX=1) -----------------------------------------------------------------------------
#include <dlfcn.h>
#include <stdlib.h>
namespace {
int (*g_roctx_push)(const char*) = nullptr;
int (*g_roctx_pop)() = nullptr;
int g_roctx_state = -1;   // -1 unknown, 0 off, 1 on
}  // namespace

bool kws_roctx_on() {
  if (g_roctx_state < 0) {
    std::lock_guard<std::mutex> lk(g_roctx_mu);
    if (g_roctx_state < 0) {
      int st = 0;
      const char* e = getenv("KWS_ROCTX");
      if (e && e[0] && e[0] != '0') {
        const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        for (const char* n : names) {
          void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
          if (!h) continue;
          g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
          g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
          if (g_roctx_push && g_roctx_pop) {
            st = 1;
            break;
          }
        }
      }
      g_roctx_state = st;
    }
  }
  return g_roctx_state == 1;
}
void kws_roctx_push(const char* name) { (void)g_roctx_push(name); }
void kws_roctx_pop() { (void)g_roctx_pop(); }

void* kws_prof_begin(hipStream_t st) {
  ProfRec* r = new ProfRec();
  if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) {
    delete r;
    return nullptr;
  }
  (void)hipEventRecord(r->a, st);
  return r;
}

void kws_prof_end(void* token, const char* name, double flops, double bytes, hipStream_t st) {
  ProfRec* r = static_cast<ProfRec*>(token);
  (void)hipEventRecord(r->b, st);
  r->name = name;
  r->flops = flops;
  r->bytes = bytes;
  kws_profiler* p = t_prof;
  if (!p) {                       // detached between begin and end: drop the record
    (void)hipEventDestroy(r->a);
    (void)hipEventDestroy(r->b);
    delete r;
    return;
  }
  std::lock_guard<std::mutex> lk(p->mu);
  p->recs.push_back(r);
}

extern "C" {

int kws_profiler_create(kws_profiler_t** out) {
  if (!out) {
    kws_set_error("profiler_create: NULL output");
    return KWS_E_INVALID;
  }
  *out = new kws_profiler();
  return KWS_OK;
}

int kws_profiler_destroy(kws_profiler_t* p) {
  if (!p) return KWS_OK;
  const int others = p->attached.load() - (t_prof == p ? 1 : 0);
  if (others != 0) {                     // another thread would keep a dangling pointer (its next launch books into p);
    kws_set_error("profiler_destroy: %d other thread(s) still attached (kws_profiler_attach(NULL) from each first)", others);
    return KWS_E_INVALID;                // a refused destroy changes nothing: the caller stays attached and keeps recording
  }
  if (t_prof == p) {                     // the caller's own attachment ends with the handle
    t_prof = nullptr;
    p->attached.fetch_sub(1);
  }
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->clear_locked();
  }
  delete p;
  return KWS_OK;
}

// the CALLING THREAD records into p from now on (NULL: stops recording)
int kws_profiler_attach(kws_profiler_t* p) {
  if (t_prof == p) return KWS_OK;
  if (t_prof) t_prof->attached.fetch_sub(1);
  if (p) p->attached.fetch_add(1);
  t_prof = p;
  return KWS_OK;
}

// Waits for every recorded event, aggregates per family, clears the records; returns the number of families.
int kws_profiler_collect(kws_profiler_t* p) {
  if (!p) return 0;
  std::lock_guard<std::mutex> lk(p->mu);
  std::map<std::string, ProfAgg> agg;
  std::vector<std::string> order;
  for (ProfRec* r : p->recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r->b) != hipSuccess || hipEventElapsedTime(&ms, r->a, r->b) != hipSuccess) continue;
    if (!agg.count(r->name)) order.push_back(r->name);
    ProfAgg& a = agg[r->name];
    a.ms += ms;
    a.flops += r->flops;
    a.bytes += r->bytes;
    a.count += 1;
  }
  p->clear_locked();
  p->out.clear();
  for (const std::string& n : order) p->out.push_back({n, agg[n]});
  return (int)p->out.size();
}

int kws_profiler_get(kws_profiler_t* p, int idx, char* name, int cap, double* ms, int64_t* count, double* flops,
                     double* bytes) {
  if (!p || !name || cap <= 0) {
    kws_set_error("profiler_get: bad arguments");
    return KWS_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(p->mu);
  if (idx < 0 || idx >= (int)p->out.size()) {
    kws_set_error("profiler_get: bad index %d", idx);
    return KWS_E_INVALID;
  }
  snprintf(name, cap, "%s", p->out[idx].first.c_str());
  const ProfAgg& a = p->out[idx].second;
  if (ms) *ms = a.ms;
  if (count) *count = a.count;
  if (flops) *flops = a.flops;
  if (bytes) *bytes = a.bytes;
  return KWS_OK;
}

}  // extern "C"
