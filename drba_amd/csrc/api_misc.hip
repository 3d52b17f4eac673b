// ABI version and error strings of libdrba_hip.so.
#include "common.hpp"

namespace {
constexpr int kTimingSlots = 4096;
hipEvent_t g_ev[kTimingSlots][2];
bool g_ev_made[kTimingSlots];
int g_armed = -1;
}  // namespace

namespace drba {
TimedLaunch take_armed_timing() {
  TimedLaunch t{nullptr, nullptr};
  if (g_armed >= 0) {
    t.start = g_ev[g_armed][0];
    t.stop = g_ev[g_armed][1];
    g_armed = -1;
  }
  return t;
}
}  // namespace drba

extern "C" {

int drba_abi_version(void) { return 1; }

int drba_timing_slots(void) { return kTimingSlots; }

int drba_timing_arm(int slot) {
  if (slot < 0 || slot >= kTimingSlots) return DRBA_EINVAL;
  if (!g_ev_made[slot]) {
    if (hipEventCreate(&g_ev[slot][0]) != hipSuccess || hipEventCreate(&g_ev[slot][1]) != hipSuccess) return DRBA_ELAUNCH;
    g_ev_made[slot] = true;
  }
  g_armed = slot;
  return DRBA_OK;
}

int drba_timing_elapsed_ms(int slot, float *ms) {
  if (slot < 0 || slot >= kTimingSlots || !ms || !g_ev_made[slot]) return DRBA_EINVAL;
  if (hipEventSynchronize(g_ev[slot][1]) != hipSuccess) return DRBA_ELAUNCH;
  return hipEventElapsedTime(ms, g_ev[slot][0], g_ev[slot][1]) == hipSuccess ? DRBA_OK : DRBA_ELAUNCH;
}

const char *drba_error_string(int code) {
  switch (code) {
    case DRBA_OK: return "ok";
    case DRBA_EINVAL: return "invalid argument";
    case DRBA_EUNSUPPORTED: return "unsupported shape or configuration";
    case DRBA_ELAUNCH: return "HIP launch failure";
  }
  return "unknown error";
}

}  // extern "C"
