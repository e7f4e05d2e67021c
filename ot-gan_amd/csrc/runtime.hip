// runtime.hip -- error reporting, version, and the HIP-event profiler behind otgan_prof_*.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "../../include/otgan.h"

static thread_local char g_err[512] = "";

void otgan_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
struct ProfClass {
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t used = 0;             // events in use
  double flops = 0, bytes = 0;
  long launches = 0;
};
std::mutex g_mu;
bool g_on = false;
ProfClass g_cls[OTGAN_PROF_NCLASS];

hipEvent_t next_event(ProfClass& c) {
  if (c.used == c.ev.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    c.ev.push_back(e);
  }
  return c.ev[c.used++];
}
}  // namespace

void otgan_prof_begin(int cls, double flops, double bytes, hipStream_t s) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfClass& c = g_cls[cls];
  c.flops += flops;
  c.bytes += bytes;
  c.launches += 1;
  hipEventRecord(next_event(c), s);
}

void otgan_prof_end(int cls, hipStream_t s) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfClass& c = g_cls[cls];
  if (c.used & 1) hipEventRecord(next_event(c), s);
}

extern "C" {

int otgan_version(void) { return OTGAN_ABI_VERSION; }
const char* otgan_last_error(void) { return g_err; }

int otgan_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return OTGAN_OK;
}

int otgan_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& c : g_cls) {
    c.used = 0;
    c.flops = c.bytes = 0;
    c.launches = 0;
  }
  return OTGAN_OK;
}

int otgan_prof_collect(int cls, double* out4) {
  if (cls < 0 || cls >= OTGAN_PROF_NCLASS || !out4) {
    otgan_set_error("otgan_prof_collect: bad class %d", cls);
    return OTGAN_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  ProfClass& c = g_cls[cls];
  double ms = 0;
  for (size_t i = 0; i + 1 < c.used; i += 2) {
    hipEventSynchronize(c.ev[i + 1]);
    float t = 0;
    hipEventElapsedTime(&t, c.ev[i], c.ev[i + 1]);
    ms += t;
  }
  out4[0] = (double)c.launches;
  out4[1] = ms;
  out4[2] = c.flops;
  out4[3] = c.bytes;
  return OTGAN_OK;
}

}  // extern "C"
