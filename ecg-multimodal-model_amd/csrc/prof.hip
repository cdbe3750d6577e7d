// Optional in-library kernel timing with HIP events on the launch stream (used by bench.py for the
// roofline figure).  Disabled by default: when off, the hooks are a single branch.
#include "ops.h"

namespace {
constexpr int MAXEV = 8192;
struct Rec { int kind; double flops; double bytes; };
bool g_on = false;
unsigned g_kinds = ~0u;  // bit k set = kind k is timed
int g_n = 0;
hipEvent_t g_start[MAXEV], g_stop[MAXEV];
bool g_created = false;
Rec g_rec[MAXEV];
int g_open = -1;
}  // namespace

void ecg_prof_begin(int kind, double flops, double bytes, hipStream_t s) {
  if (!g_on || g_n >= MAXEV || !((g_kinds >> kind) & 1u)) { g_open = -1; return; }
  g_open = g_n++;
  g_rec[g_open].kind = kind;
  g_rec[g_open].flops = flops;
  g_rec[g_open].bytes = bytes;
  (void)hipEventRecord(g_start[g_open], s);
}
void ecg_prof_end(hipStream_t s) {
  if (!g_on || g_open < 0) return;
  (void)hipEventRecord(g_stop[g_open], s);
  g_open = -1;
}

extern "C" int ecgmm_prof_enable(int on) {
  if (on && !g_created) {
    for (int i = 0; i < MAXEV; ++i) {
      if (hipEventCreate(&g_start[i]) != hipSuccess || hipEventCreate(&g_stop[i]) != hipSuccess)
        ECG_FAIL(ECGMM_ERR_LAUNCH, "prof: hipEventCreate failed");
    }
    g_created = true;
  }
  // on = 1: every kind; on = 2: only the bf16 implicit-GEMM kernel class (kinds 0, 1); on = 3: only its exact-fp32
  // instantiation (kinds 5, 6) -- an event pair costs ~1 us of stream time, so the timed region of bench.py brackets
  // only the kernel its roofline line is about
  g_on = on != 0;
  g_kinds = on == 2 ? 0x3u : on == 3 ? 0x60u : ~0u;
  g_n = 0;
  g_open = -1;
  return 0;
}

// Suspends / resumes the bracketing without discarding what was recorded (bench.py samples every 4th step of its
// timed region, so the event pairs cost the headline number under 1 %).
extern "C" int ecgmm_prof_pause(int paused) {
  g_on = g_created && !paused;
  return 0;
}

// Synchronises the recorded events and accumulates per-kind totals; nkinds entries each.
extern "C" int ecgmm_prof_collect(int nkinds, double* ms, double* flops, double* bytes, int64_t* count) {
  for (int k = 0; k < nkinds; ++k) { ms[k] = 0; flops[k] = 0; bytes[k] = 0; count[k] = 0; }
  for (int i = 0; i < g_n; ++i) {
    if (hipEventSynchronize(g_stop[i]) != hipSuccess) ECG_FAIL(ECGMM_ERR_LAUNCH, "prof: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_start[i], g_stop[i]) != hipSuccess) ECG_FAIL(ECGMM_ERR_LAUNCH, "prof: elapsed failed");
    int k = g_rec[i].kind;
    if (k >= 0 && k < nkinds) { ms[k] += t; flops[k] += g_rec[i].flops; bytes[k] += g_rec[i].bytes; count[k] += 1; }
  }
  int n = g_n;
  g_n = 0;
  return n >= MAXEV ? ECGMM_ERR_WORKSPACE : 0;
}

// ---- step timeline (diagnostic): timestamped marks on the caller's stream at the plans' phase boundaries, read back as
// milliseconds since the first mark.  Off by default (one branch per mark).
namespace {
constexpr int TLMAX = 4096;
bool g_tl_on = false, g_tl_created = false;
int g_tl_n = 0;
hipEvent_t g_tl_ev[TLMAX];
int g_tl_id[TLMAX];
}  // namespace
void ecg_tl_mark(int id, hipStream_t s) {
  if (!g_tl_on || g_tl_n >= TLMAX) return;
  g_tl_id[g_tl_n] = id;
  (void)hipEventRecord(g_tl_ev[g_tl_n++], s);
}
extern "C" int ecgmm_tl_enable(int on) {
  if (on && !g_tl_created) {
    for (int i = 0; i < TLMAX; ++i)
      if (hipEventCreate(&g_tl_ev[i]) != hipSuccess) ECG_FAIL(ECGMM_ERR_LAUNCH, "timeline: hipEventCreate failed");
    g_tl_created = true;
  }
  g_tl_on = on != 0;
  g_tl_n = 0;
  return 0;
}
extern "C" int ecgmm_tl_mark(int id, void* stream) {
  ecg_tl_mark(id, (hipStream_t)stream);
  return 0;
}
// ids[i], ms[i] (since mark 0) for up to `cap` marks; returns the number of marks recorded
extern "C" int ecgmm_tl_collect(int cap, int* ids, float* ms) {
  int n = g_tl_n < cap ? g_tl_n : cap;
  for (int i = 0; i < n; ++i) {
    (void)hipEventSynchronize(g_tl_ev[i]);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_tl_ev[0], g_tl_ev[i]);
    ids[i] = g_tl_id[i];
    ms[i] = t;
  }
  g_tl_n = 0;
  return n;
}
