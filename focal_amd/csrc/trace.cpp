// Launch trace of libfocal_hip: per-kernel durations of eager steps, measured on the launches themselves.
//
// Every kernel of the library is launched through FOCAL_LAUNCH (common.hpp).  While a trace is open each launch goes out as
// hipExtLaunchKernelGGL with a start / stop event pair attached to the dispatch (mode FOCAL_TRACE_DISPATCH: the elapsed time is the
// dispatch's own begin -> end timestamps, what rocprofv3 --kernel-trace reports, independent of what the host does between launches)
// or, as a cross-check, between two hipEventRecord calls placed directly around the launch (mode FOCAL_TRACE_EVENTS).  Records are
// labelled with the (mangled) symbol name of the launched kernel -- one per template instantiation, so bench.py's groups are the rows
// of `rocprofv3 --stats` (bench.py and tools/rocprof_reference.py demangle both sides the same way).
// Measurement only: nothing here is on the product path while no trace is open (one predictable branch per launch).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/focal_hip.h"

void focal_set_error(const char* fmt, ...);

int g_focal_trace_on = 0;

namespace {
struct Rec {
  const void* kernel;
  hipStream_t stream;
  unsigned grid[3], block[3];
  hipEvent_t e0, e1;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;  // events of finished traces, reused
std::unordered_map<const void*, std::string> g_names;
int g_capacity = 0;
int g_mode = 0;

hipEvent_t take_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

const std::string& name_of(const Rec& r) {
  auto it = g_names.find(r.kernel);
  if (it != g_names.end()) return it->second;
  // The host-side handle of a kernel carries the kernel's own (mangled) symbol name and is exported with it: dladdr gives exactly the name
  // rocprofv3 -M prints.  (hipKernelNameRefByPtr demangles when it can -- and spells __bf16 as "bool _Accum" -- so it is the fallback only.)
  Dl_info info;
  if (dladdr(r.kernel, &info) && info.dli_sname && info.dli_saddr == r.kernel) return g_names.emplace(r.kernel, info.dli_sname).first->second;
  const char* nm = hipKernelNameRefByPtr(r.kernel, r.stream);
  return g_names.emplace(r.kernel, nm ? nm : "?").first->second;
}
}  // namespace

// Called by FOCAL_LAUNCH while a trace is open.  Returns the mode to launch with (0: plain launch, no record -- the stream is being
// captured into a graph or the trace is full) and the event pair of the new record.
int focal_trace_slot(const void* kernel, dim3 grid, dim3 block, hipStream_t st, hipEvent_t* e0, hipEvent_t* e1) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_focal_trace_on || (int)g_recs.size() >= g_capacity) return 0;
  Rec r;
  r.kernel = kernel;
  r.stream = st;
  r.grid[0] = grid.x, r.grid[1] = grid.y, r.grid[2] = grid.z;
  r.block[0] = block.x, r.block[1] = block.y, r.block[2] = block.z;
  r.e0 = take_event();
  r.e1 = take_event();
  if (!r.e0 || !r.e1) return 0;
  g_recs.push_back(r);
  *e0 = r.e0;
  *e1 = r.e1;
  return g_mode;
}

extern "C" int focal_trace_begin(int capacity, int mode) {
  if (capacity <= 0 || (mode != FOCAL_TRACE_DISPATCH && mode != FOCAL_TRACE_EVENTS)) {
    focal_set_error("focal_trace_begin: capacity %d, mode %d", capacity, mode);
    return FOCAL_EINVAL;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    g_pool.push_back(r.e0);
    g_pool.push_back(r.e1);
  }
  g_recs.clear();
  g_recs.reserve(capacity);
  g_capacity = capacity;
  g_mode = mode;
  g_focal_trace_on = 1;
  return FOCAL_OK;
}

extern "C" int focal_trace_end(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_focal_trace_on = 0;
  return FOCAL_OK;
}

extern "C" int focal_trace_count(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_recs.size();
}

extern "C" int focal_trace_read(int first, int n, focal_trace_record* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (first < 0 || n < 0 || first + n > (int)g_recs.size() || (n && !out)) {
    focal_set_error("focal_trace_read: [%d, %d) of %d records", first, first + n, (int)g_recs.size());
    return FOCAL_EINVAL;
  }
  for (int i = 0; i < n; ++i) {
    const Rec& r = g_recs[first + i];
    focal_trace_record& o = out[i];
    memset(&o, 0, sizeof(o));
    const std::string& nm = name_of(r);
    strncpy(o.kernel, nm.c_str(), sizeof(o.kernel) - 1);
    for (int k = 0; k < 3; ++k) o.grid[k] = r.grid[k], o.block[k] = r.block[k];
    o.stream = (void*)r.stream;
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
    if (e != hipSuccess) {
      focal_set_error("focal_trace_read: record %d (%s): %s", first + i, o.kernel, hipGetErrorString(e));
      (void)hipGetLastError();
      return FOCAL_EHIP;
    }
    o.us = ms * 1e3f;
  }
  return FOCAL_OK;
}
