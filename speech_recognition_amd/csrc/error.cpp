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

}  // extern "C"
