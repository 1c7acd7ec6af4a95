// vf_trace.hip — roctx ranges around the library's work (SURVEY 5 "tracing": the reference has three torch.Timers,
// train_vid_weighted.lua:323-325; SURVEY 8(b) asks the boundary for timers / roctx range helpers).
//
// vf_range_push / vf_range_pop / vf_mark forward to roctx when a roctx library can be found (the one rocprofv3 injects,
// librocprofiler-sdk-roctx.so, else libroctx64.so) and are no-ops otherwise; `rocprofv3 --marker-trace` then shows the host
// ranges (closures, Adam steps — the host mirror pushes them — and, with vf_trace_enable(1) or VF_ROCTX=1, one range per
// library launch named like bench.py's kernel table) beside the kernel trace.  The vf_prof_* timers stay the tool for numbers.
#include <dlfcn.h>

#include <cstdlib>

#include "vf_common.h"

namespace {

struct Roctx {
  bool tried = false;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  void (*mark)(const char*) = nullptr;
};
Roctx g_roctx;
int g_trace_on = -1;      // -1: not decided (VF_ROCTX decides at first use)
int g_depth = 0;

void roctx_bind() {
  if (g_roctx.tried) return;
  g_roctx.tried = true;
  const char* names[] = {getenv("VF_ROCTX_LIB"), "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                         "libroctx64.so.4"};
  void* h = nullptr;
  for (const char* n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // a copy the profiler already injected comes first
  for (const char* n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return;
  *(void**)(&g_roctx.push) = dlsym(h, "roctxRangePushA");
  *(void**)(&g_roctx.pop) = dlsym(h, "roctxRangePop");
  *(void**)(&g_roctx.mark) = dlsym(h, "roctxMarkA");
}

}  // namespace

bool vf_trace_enabled() {
  if (g_trace_on < 0) g_trace_on = (getenv("VF_ROCTX") && atoi(getenv("VF_ROCTX"))) ? 1 : 0;
  return g_trace_on == 1;
}

// 1 if a roctx library is bound (ranges reach the profiler), 0 if the calls below are no-ops
VF_API int vf_trace_available(void) {
  roctx_bind();
  return g_roctx.push != nullptr && g_roctx.pop != nullptr;
}
// per-launch ranges inside the library on / off (the explicit push / pop / mark below always work)
VF_API int vf_trace_enable(int on) {
  g_trace_on = on ? 1 : 0;
  return 0;
}
VF_API int vf_range_push(const char* name) {
  roctx_bind();
  ++g_depth;
  if (g_roctx.push) g_roctx.push(name ? name : "");
  return 0;
}
VF_API int vf_range_pop(void) {
  if (g_depth <= 0) {
    vf_set_error("vf_range_pop: no range is open");
    return 2;
  }
  --g_depth;
  if (g_roctx.pop) g_roctx.pop();
  return 0;
}
VF_API int vf_mark(const char* message) {
  roctx_bind();
  if (g_roctx.mark) g_roctx.mark(message ? message : "");
  return 0;
}
// open ranges (tests: pushes and pops balance)
VF_API int vf_range_depth(void) { return g_depth; }
