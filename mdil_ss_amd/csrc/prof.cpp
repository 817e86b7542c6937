// Per-launch timing of the conv / weight-gradient entry points with HIP events recorded on the
// launch stream, inside the library: the measured launches are the ones the shipped path makes
// (block-level C ABI, deferred reductions), not a re-orchestrated copy of them.  Used by bench.py's
// roofline leg only.  The one place the library keeps process-global state, and only between
// mdil_profile_begin and mdil_profile_end; off, a scope costs one relaxed atomic load.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

struct Slot {
  hipEvent_t e0, e1;
  mdil_profile_record r;
};

std::atomic<bool> g_on{false};
std::mutex g_mu;
std::vector<Slot> g_slots;
int g_used = 0;

}  // namespace

MdilProfScope::MdilProfScope(hipStream_t st_, int kind, const mdil_geom* g, int cin, int cout)
    : idx(-1), path(0), st(st_) {
  if (!g_on.load(std::memory_order_relaxed) || g == nullptr) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on.load() || g_used >= (int)g_slots.size()) return;
  idx = g_used++;
  Slot& s = g_slots[idx];
  s.r.kind = kind;
  s.r.cin = cin;
  s.r.cout = cout;
  s.r.ntaps = g->ntaps;
  s.r.npix = (long long)g->N * g->HO * g->WO;
  s.r.path = 0;
  s.r.ms = -1.f;
  (void)hipEventRecord(s.e0, st);
}

MdilProfScope::~MdilProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx >= (int)g_slots.size()) return;
  g_slots[idx].r.path = path;
  (void)hipEventRecord(g_slots[idx].e1, st);
}

extern "C" int mdil_profile_begin(int capacity) {
  MDIL_CHECK_ARG(capacity > 0 && capacity <= (1 << 20), "profile_begin: capacity %d", capacity);
  std::lock_guard<std::mutex> lk(g_mu);
  MDIL_CHECK_ARG(!g_on.load(), "profile_begin: a profile is already running");
  g_slots.resize(capacity);
  for (Slot& s : g_slots) {
    if (hipEventCreate(&s.e0) != hipSuccess || hipEventCreate(&s.e1) != hipSuccess) {
      mdil_set_error("profile_begin: hipEventCreate failed");
      return MDIL_ERR_LAUNCH;
    }
  }
  g_used = 0;
  g_on.store(true);
  return MDIL_OK;
}

extern "C" int mdil_profile_end(mdil_profile_record* out, int max_records) {
  std::lock_guard<std::mutex> lk(g_mu);
  MDIL_CHECK_ARG(g_on.load(), "profile_end: no profile running");
  g_on.store(false);
  int n = 0;
  for (int i = 0; i < g_used; ++i) {
    Slot& s = g_slots[i];
    if (hipEventSynchronize(s.e1) == hipSuccess) (void)hipEventElapsedTime(&s.r.ms, s.e0, s.e1);
    if (out && n < max_records) out[n++] = s.r;
  }
  for (Slot& s : g_slots) {
    (void)hipEventDestroy(s.e0);
    (void)hipEventDestroy(s.e1);
  }
  g_slots.clear();
  g_used = 0;
  return n;
}
