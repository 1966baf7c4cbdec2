// Error plumbing, version/arch queries and the optional GEMM timing hook of libldmae_hip.
#include "common.h"

#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void ldmae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ldmae_last_error(void) { return g_err; }
extern "C" const char* ldmae_version(void) { return "ldmae_hip 0.3 (round 3)"; }
extern "C" const char* ldmae_arch(void) { return "gfx950"; }

// ---------------------------------------------------------------- timing hook (bench.py roofline line)
namespace {
struct ProfRec { hipEvent_t a, b; double flops; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}  // namespace

bool ldmae_prof_is_on() { return g_prof_on; }

long ldmae_prof_begin(hipStream_t st, double flops) {
  ProfRec r;
  hipEventCreate(&r.a);
  hipEventCreate(&r.b);
  r.flops = flops;
  hipEventRecord(r.a, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
  return (long)g_prof.size() - 1;
}
void ldmae_prof_end(long idx, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx >= 0 && idx < (long)g_prof.size()) hipEventRecord(g_prof[idx].b, st);
}

extern "C" int ldmae_prof_enable(int on) {
  g_prof_on = on != 0;
  return LDMAE_OK;
}

extern "C" int ldmae_prof_collect(double* total_ms, double* total_flops, long* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0, fl = 0;
  for (auto& r : g_prof) {
    hipEventSynchronize(r.b);
    float t = 0;
    hipEventElapsedTime(&t, r.a, r.b);
    ms += t;
    fl += r.flops;
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = (long)g_prof.size();
  g_prof.clear();
  return LDMAE_OK;
}

// ---------------------------------------------------------------- launch counts by kernel family (include/ldmae_hip.h)
#include <atomic>
static std::atomic<long> g_counts[9];
void ldmae_count(int family) { g_counts[family].fetch_add(1, std::memory_order_relaxed); }
extern "C" int ldmae_launch_counts(long* counts, int n, int reset) {
  LDMAE_REQUIRE(counts && n >= 0 && n <= 9, "launch_counts: counts null or n outside 0..9");
  for (int i = 0; i < n; ++i) counts[i] = g_counts[i].load(std::memory_order_relaxed);
  if (reset)
    for (auto& c : g_counts) c.store(0, std::memory_order_relaxed);
  return LDMAE_OK;
}

#ifdef LDMAE_DIAG
// ---------------------------------------------------------------- tuning knobs (kernel variant selection; not part of the reference seam)
static int g_tune[32] = {0};
int ldmae_tune_get(int key) { return (key >= 0 && key < 32) ? g_tune[key] : 0; }
extern "C" int ldmae_tune_query(int key) { return ldmae_tune_get(key); }
extern "C" int ldmae_tune(int key, int value) {
  if (key < 0 || key >= 32) LDMAE_FAIL(LDMAE_ERR_INVALID, "tune: key %d out of range", key);
  g_tune[key] = value;
  return LDMAE_OK;
}
#endif
