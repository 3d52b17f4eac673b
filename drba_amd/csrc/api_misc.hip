// ABI version and error strings of libdrba_hip.so.
#include "common.hpp"

#include <cxxabi.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

// Kernel trace (measurement only): between drba_trace_begin() and drba_trace_end() EVERY kernel launch of the library
// carries an event pair on its own dispatch packet (hipExtLaunchKernelGGL), so each record is the kernel's own
// execution time -- what rocprofv3's kernel trace reports -- and bench.py can rank the whole step by kernel symbol.
// Process-global state driven by ONE host thread (the launches of both HIP streams come from the same thread).
namespace {
constexpr int kTraceSlots = 16384;
struct Rec {
  const void *fn;        // host stub of the kernel: its symbol name is looked up when the record is READ (drba_trace_get),
  const char *fallback;  // never inside the traced region
  unsigned gx, gy, gz;
  hipStream_t stream;
};
hipEvent_t g_ev[kTraceSlots][2];
bool g_ev_made[kTraceSlots];
std::vector<Rec> g_recs;
std::unordered_map<const void *, std::string> g_names;

const std::string &kernel_name(const void *host_fn, const char *fallback) {
  hipStream_t stream = nullptr;
  auto it = g_names.find(host_fn);
  if (it != g_names.end()) return it->second;
  std::string nm;
  const char *mangled = hipKernelNameRefByPtr(host_fn, stream);
  if (mangled && *mangled) {
    int status = 1;
    char *dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    nm = (status == 0 && dem) ? dem : mangled;
    std::free(dem);
    // rocprofv3 prints "void ns::name<...>(args)"; keep "ns::name<...>" (no return type, anonymous-namespace tag or
    // argument list) so that records group like its kernel-name column
    if (nm.rfind("void ", 0) == 0) nm.erase(0, 5);
    const std::string anon = "(anonymous namespace)::";
    for (size_t p = nm.find(anon); p != std::string::npos; p = nm.find(anon)) nm.erase(p, anon.size());
    int depth = 0;
    for (size_t i = 0; i < nm.size(); ++i) {
      if (nm[i] == '<') ++depth;
      else if (nm[i] == '>') --depth;
      else if (nm[i] == '(' && depth == 0) {
        nm.resize(i);
        break;
      }
    }
  } else {
    nm = fallback;
  }
  (void)hipGetLastError();
  return g_names.emplace(host_fn, nm).first->second;
}
}  // namespace

namespace drba {
bool g_trace_on = false;

TimedLaunch trace_launch(const void *host_fn, const char *fallback, dim3 grid, hipStream_t stream) {
  TimedLaunch t{nullptr, nullptr};
  const size_t slot = g_recs.size();
  if (slot >= (size_t)kTraceSlots) return t;
  if (!g_ev_made[slot]) {
    if (hipEventCreate(&g_ev[slot][0]) != hipSuccess || hipEventCreate(&g_ev[slot][1]) != hipSuccess) return t;
    g_ev_made[slot] = true;
  }
  g_recs.push_back(Rec{host_fn, fallback, grid.x, grid.y, grid.z, stream});
  t.start = g_ev[slot][0];
  t.stop = g_ev[slot][1];
  return t;
}
hipError_t max_dynamic_lds(const void *kernel, int bytes) {
  // the LARGEST size applied so far per (kernel, device): a later caller asking for more re-issues the attribute (a second
  // caller with a bigger tile used to be ignored silently), and a failed call is not remembered (it may have been transient)
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, int> applied;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(kernel, dev);
  auto it = applied.find(key);
  if (it != applied.end() && it->second >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) applied[key] = bytes;
  return e;
}
bool g_range_check = false;

// Status words (drba_status_word): one 8-byte host-mapped allocation per device, made on request, never freed.  The table is
// read on every family-4 launch (an atomic pointer load indexed by the current device) and written once per device.
namespace {
constexpr int kMaxDevices = 64;
std::atomic<unsigned long long *> g_status_host[kMaxDevices];
std::atomic<unsigned char *> g_status_dev[kMaxDevices];
std::mutex g_status_mu;
}  // namespace
unsigned char *status_bytes() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return g_status_dev[dev].load(std::memory_order_acquire);
}

namespace {
__global__ void __launch_bounds__(256) nonfinite_kernel(const float *__restrict__ x, size_t n, int *__restrict__ flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned u = __float_as_uint(x[i]);
    bad |= (u & 0x7f800000u) == 0x7f800000u;  // inf or NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace

int range_scan(const float *out, size_t n, void *stream) {
  static std::mutex mu;
  static std::map<int, int *> flags;  // one device word per GPU, allocated on first use (debug mode only)
  int dev = 0;
  if (!out || n == 0) return DRBA_OK;
  if (hipGetDevice(&dev) != hipSuccess) return DRBA_ELAUNCH;
  int *flag = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = flags.find(dev);
    if (it == flags.end()) {
      if (hipMalloc((void **)&flag, sizeof(int)) != hipSuccess) return DRBA_ELAUNCH;
      flags[dev] = flag;
    } else {
      flag = it->second;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  int host = 0;
  if (hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess) return DRBA_ELAUNCH;
  hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, out, n, flag);
  if (hipGetLastError() != hipSuccess) return DRBA_ELAUNCH;
  if (hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return DRBA_ELAUNCH;
  return host ? DRBA_EUNSUPPORTED : DRBA_OK;
}
}  // namespace drba

extern "C" {

int drba_abi_version(void) { return DRBA_ABI_VERSION; }

int drba_set_range_check(int on) {
  const int was = drba::g_range_check ? 1 : 0;
  drba::g_range_check = on != 0;
  return was;
}

int drba_status_word(volatile unsigned long long **host_word) {
  if (!host_word) return DRBA_EINVAL;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= drba::kMaxDevices) return DRBA_EINVAL;
  std::lock_guard<std::mutex> lock(drba::g_status_mu);
  unsigned long long *h = drba::g_status_host[dev].load(std::memory_order_acquire);
  if (!h) {
    void *hp = nullptr, *dp = nullptr;
    // coherent (fine-grained) host memory: a kernel's store is visible to the host once the kernel has finished
    if (hipHostMalloc(&hp, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return DRBA_ELAUNCH;
    *static_cast<unsigned long long *>(hp) = 0;
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
      (void)hipHostFree(hp);
      return DRBA_ELAUNCH;
    }
    h = static_cast<unsigned long long *>(hp);
    drba::g_status_host[dev].store(h, std::memory_order_release);
    drba::g_status_dev[dev].store(static_cast<unsigned char *>(dp), std::memory_order_release);
  }
  *host_word = h;
  return DRBA_OK;
}

int drba_status_clear(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= drba::kMaxDevices) return DRBA_EINVAL;
  unsigned long long *h = drba::g_status_host[dev].load(std::memory_order_acquire);
  if (h) *reinterpret_cast<volatile unsigned long long *>(h) = 0;
  return DRBA_OK;
}

int drba_stream_create_cu_mask(const uint32_t *mask, int words, void **stream) {
  if (!mask || words <= 0 || words > 32 || !stream) return DRBA_EINVAL;
  bool any = false;
  for (int i = 0; i < words; ++i) any |= mask[i] != 0;
  if (!any) return DRBA_EINVAL;
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) {
    (void)hipGetLastError();
    return DRBA_ELAUNCH;
  }
  *stream = (void *)s;
  return DRBA_OK;
}

int drba_stream_destroy(void *stream) {
  if (!stream) return DRBA_EINVAL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? DRBA_OK : DRBA_ELAUNCH;
}

int drba_trace_begin(void) {
  // events are created here, outside any timed region (hipEventCreate costs tens of microseconds each)
  for (int i = 0; i < 2048; ++i) {
    if (g_ev_made[i]) continue;
    if (hipEventCreate(&g_ev[i][0]) != hipSuccess || hipEventCreate(&g_ev[i][1]) != hipSuccess) return DRBA_ELAUNCH;
    g_ev_made[i] = true;
  }
  g_recs.clear();
  drba::g_trace_on = true;
  return DRBA_OK;
}

int drba_trace_resume(void) {
  drba::g_trace_on = true;
  return DRBA_OK;
}

int drba_trace_end(void) {
  drba::g_trace_on = false;
  return DRBA_OK;
}

int drba_trace_count(void) { return (int)g_recs.size(); }

int drba_trace_get(int i, const char **name, unsigned *grid3, float *ms) {
  if (i < 0 || (size_t)i >= g_recs.size() || !name || !grid3 || !ms) return DRBA_EINVAL;
  if (hipEventSynchronize(g_ev[i][1]) != hipSuccess) return DRBA_ELAUNCH;
  if (hipEventElapsedTime(ms, g_ev[i][0], g_ev[i][1]) != hipSuccess) return DRBA_ELAUNCH;
  *name = kernel_name(g_recs[i].fn, g_recs[i].fallback).c_str();
  grid3[0] = g_recs[i].gx;
  grid3[1] = g_recs[i].gy;
  grid3[2] = g_recs[i].gz;
  return DRBA_OK;
}

int drba_trace_get_start(int i, float *ms_since_first, unsigned long long *stream) {
  if (i < 0 || (size_t)i >= g_recs.size() || !ms_since_first || !stream) return DRBA_EINVAL;
  if (hipEventSynchronize(g_ev[i][0]) != hipSuccess) return DRBA_ELAUNCH;
  *ms_since_first = 0.f;
  if (i > 0 && hipEventElapsedTime(ms_since_first, g_ev[0][0], g_ev[i][0]) != hipSuccess) return DRBA_ELAUNCH;
  *stream = (unsigned long long)(uintptr_t)g_recs[i].stream;
  return DRBA_OK;
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
