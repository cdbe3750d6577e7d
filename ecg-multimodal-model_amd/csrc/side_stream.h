// Library-owned side stream for the weight-gradient kernels of an encoder's backward.
// wgrad only feeds the optimizer, so it is taken off the critical path dgrad -> BN-backward -> dgrad: it runs on a
// HIP stream owned by the library, forked from / joined to the caller's stream with events (all asynchronous; a call joins
// before it returns unless the caller asked to defer that, so to the caller the work is still ordered on the stream it
// passed).  NOT supported under hipGraph stream capture: the plans' launch paths query device properties / set function
// attributes behind first-use guards and record events on this library-owned (non-capturing) stream; a whole-step capture
// attempted in round 2 faulted inside capture and was dropped (the step is not launch-bound: host enqueue 3.1 ms vs 7 ms GPU).  MFMA-bound wgrad overlaps the HBM-bound BatchNorm passes.  Each encoder plan has
// its own instance: the two encoders of the multimodal model run concurrently and must not couple through one stream.
#pragma once
#include <cstdlib>

#include "ops.h"

struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ev[64];
  int next = 0;
  bool ok = false, enabled = true, defer_join = false;
  hipEvent_t doneA = nullptr, doneB = nullptr, doneC = nullptr;  // last reader of dy / dy1 / dyd
  hipEvent_t done2[3][2] = {};  // the same for plans that ping-pong those buffers between blocks

  int init() {
    if (ok) return 0;
    const char* e = getenv("ECGMM_SIDE_WGRAD");
    enabled = !(e && e[0] == '0');
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
      ECG_FAIL(ECGMM_ERR_LAUNCH, "side stream creation failed");
    for (int i = 0; i < 64; ++i)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess)
        ECG_FAIL(ECGMM_ERR_LAUNCH, "side event creation failed");
    ok = true;
    return 0;
  }
  hipEvent_t next_ev() { return ev[next++ & 63]; }  // (a held event is waited on within ~12 later records)
  // everything enqueued on `main` so far happens-before later work on the side stream
  void fork(hipStream_t main) {
    hipEvent_t e = next_ev();
    (void)hipEventRecord(e, main);
    (void)hipStreamWaitEvent(s, e, 0);
  }
  // marks the side stream's tail (for a later main_wait)
  hipEvent_t mark() {
    hipEvent_t e = next_ev();
    (void)hipEventRecord(e, s);
    return e;
  }
  // everything the side stream has been given so far happens-before later work on `stream`
  int wait_on(hipStream_t stream) {
    hipEvent_t e = next_ev();
    if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(stream, e, 0) != hipSuccess)
      ECG_FAIL(ECGMM_ERR_LAUNCH, "side wait failed");
    return 0;
  }
};
inline void main_wait(hipStream_t main, hipEvent_t& e) {
  if (e) (void)hipStreamWaitEvent(main, e, 0);
  e = nullptr;
}
