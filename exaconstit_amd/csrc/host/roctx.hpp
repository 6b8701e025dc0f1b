// Named profiling regions, the counterpart of the reference's Caliper marks (reference src/mechanics_log.hpp:4-15, e.g.
// CALI_MARK_BEGIN("ecmech_kernel") in src/mechanics_ecmech.cpp:237-257): roctx ranges, visible to `rocprofv3 --marker-trace`.
// libroctx64 is looked up at run time; without it (or outside a profiler) the calls are no-ops.
#pragma once
#include <dlfcn.h>

namespace exa_host {

struct Roctx {
   int (*push)(const char*) = nullptr;
   int (*pop)() = nullptr;
   Roctx() {
      // rocprofv3 intercepts the SDK's roctx library; libroctx64 (roctracer) is the fall-back for the older tools
      void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
      if (h) { push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop"); }
      if (!push || !pop) { push = nullptr; pop = nullptr; }
   }
   static Roctx& get() { static Roctx r; return r; }
};

// RAII range
struct ProfRegion {
   bool on;
   explicit ProfRegion(const char* name) { Roctx& r = Roctx::get(); on = r.push != nullptr; if (on) r.push(name); }
   ~ProfRegion() { if (on) Roctx::get().pop(); }
   ProfRegion(const ProfRegion&) = delete; ProfRegion& operator=(const ProfRegion&) = delete;
};

}  // namespace exa_host
