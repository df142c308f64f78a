#include "common.h"
#include <cstdarg>

static thread_local char g_err[512] = "";

void yogo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* yogo_hip_last_error(void) { return g_err; }

extern "C" int yogo_hip_abi_version(void) { return 7; }

// ---- launch log: which kernel instantiation (and planner parameters) each entry point launched -------------------------------
// Off by default (one relaxed load per launch).  The parity tests switch it on to prove that the instantiations and tilings a
// production-size batch uses are the ones they compared with the oracle (tests/test_gpu_production_shapes.py).
#include <atomic>
#include <mutex>
#include <string>

static std::atomic<int> g_log_on{0};
static std::mutex g_log_mu;
static std::string g_log;

bool yogo_launch_log_enabled() { return g_log_on.load(std::memory_order_relaxed) != 0; }

void yogo_launch_log(const char* fmt, ...) {
  if (!yogo_launch_log_enabled()) return;
  char line[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(line, sizeof(line), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lk(g_log_mu);
  if (g_log.size() < (size_t)(8u << 20)) {  // bounded: a forgotten switch cannot grow without limit
    g_log += line;
    g_log += '\n';
  }
}

// enable = 1: clear and start recording; 0: stop (the text stays readable)
extern "C" int yogo_hip_launch_log(int enable) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  if (enable) g_log.clear();
  g_log_on.store(enable ? 1 : 0, std::memory_order_relaxed);
  return YOGO_OK;
}

// copies the recorded text (one line per launch, NUL terminated, truncated to cap) and reports the bytes it needs
extern "C" int yogo_hip_launch_log_read(char* buf, size_t cap, size_t* needed) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  if (needed) *needed = g_log.size() + 1;
  if (buf && cap > 0) {
    const size_t n = g_log.size() < cap - 1 ? g_log.size() : cap - 1;
    memcpy(buf, g_log.data(), n);
    buf[n] = 0;
  }
  return YOGO_OK;
}
