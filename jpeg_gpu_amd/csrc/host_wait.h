/* host_wait.h — waiting for the device without holding a core.
 *
 * Measured on this stack (ROCm 7.2, MI355X; tools/wait_probe.hip, profiles/r2_host_waits.txt):
 * hipStreamSynchronize AND hipEventSynchronize on an event created with hipEventBlockingSync
 * both SPIN — thread CPU time equals wall time for the whole wait, whatever was queued last
 * and whatever HSA_ENABLE_INTERRUPT / ROC_ACTIVE_WAIT_TIMEOUT say.  Right for one frame's
 * latency; wrong for a pipeline whose lanes wait several milliseconds per group on a host
 * that grants fewer cores than lanes (a lane burnt ~6 ms of CPU per group of 32 frames
 * doing nothing).  So the throughput paths poll the event and sleep in between: a few
 * queries back to back (many waits are over within tens of microseconds), then naps of 20 to
 * 40 us (each ~50 us longer in practice: timer slack; longer naps measured 2 % slower and no
 * cheaper, JGA_WAIT_NAP_US).  Costs at most one nap of latency per wait. */
#ifndef JGA_HOST_WAIT_H
#define JGA_HOST_WAIT_H (1)
#include <hip/hip_runtime_api.h>
#include <stdlib.h>
#include <time.h>
#include "jga_tune.h"

static inline hipError_t jga_event_wait_sleeping(hipEvent_t ev) {
  hipError_t e = hipSuccess;
  for (int i = 0; i < 8; i++) {
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
  }
  static const long nap_cap = jga_tune("JGA_WAIT_NAP_US") ? atol(jga_tune("JGA_WAIT_NAP_US"))*1000 : 40000;   // tuning knob
  long nap_ns = 20000;
  for (;;) {
    timespec ts = {0, nap_ns};
    nanosleep(&ts, NULL);
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (nap_ns < nap_cap) nap_ns += nap_ns/2;
  }
}

/* Everything queued on `st` so far, through `ev` (any event of the caller's). */
static inline hipError_t jga_stream_wait_sleeping(hipStream_t st, hipEvent_t ev) {
  const hipError_t e = hipEventRecord(ev, st);
  return e != hipSuccess ? e : jga_event_wait_sleeping(ev);
}
#endif
