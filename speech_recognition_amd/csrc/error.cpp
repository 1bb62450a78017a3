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
std::mutex g_prof_mu;
bool g_prof_enabled = false;
std::vector<ProfRec*> g_prof_recs;
std::vector<std::pair<std::string, ProfAgg>> g_prof_out;

void prof_clear_locked() {
  for (ProfRec* r : g_prof_recs) {
    (void)hipEventDestroy(r->a);
    (void)hipEventDestroy(r->b);
    delete r;
  }
  g_prof_recs.clear();
}
}  // namespace

bool kws_prof_on() { return g_prof_enabled; }

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
    std::lock_guard<std::mutex> lk(g_prof_mu);
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
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(r);
}

extern "C" {

int kws_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on) {
    prof_clear_locked();
    g_prof_out.clear();
  }
  g_prof_enabled = on != 0;
  return KWS_OK;
}

// Waits for every recorded event, aggregates per family; returns the number of families.
int kws_profile_collect(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, ProfAgg> agg;
  std::vector<std::string> order;
  for (ProfRec* r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r->b) != hipSuccess || hipEventElapsedTime(&ms, r->a, r->b) != hipSuccess) continue;
    if (!agg.count(r->name)) order.push_back(r->name);
    ProfAgg& a = agg[r->name];
    a.ms += ms;
    a.flops += r->flops;
    a.bytes += r->bytes;
    a.count += 1;
  }
  prof_clear_locked();
  g_prof_out.clear();
  for (const std::string& n : order) g_prof_out.push_back({n, agg[n]});
  return (int)g_prof_out.size();
}

int kws_profile_get(int idx, char* name, int cap, double* ms, int64_t* count, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx < 0 || idx >= (int)g_prof_out.size() || !name || cap <= 0) {
    kws_set_error("profile_get: bad index %d", idx);
    return KWS_E_INVALID;
  }
  snprintf(name, cap, "%s", g_prof_out[idx].first.c_str());
  const ProfAgg& a = g_prof_out[idx].second;
  if (ms) *ms = a.ms;
  if (count) *count = a.count;
  if (flops) *flops = a.flops;
  if (bytes) *bytes = a.bytes;
  return KWS_OK;
}

}  // extern "C"
