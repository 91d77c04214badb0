// Trace ranges with the names of the reference's own trace points, as rocTX ranges (rocprofv3 --marker-trace shows them
// next to the kernels they enclose).  The reference's l1_tracer writes Chrome-trace events "process_pdsch"
// (R/lib/phy/upper/downlink_processor_single_executor_impl.cpp:116-125), "CB batch" and "process_dmrs"
// (R/lib/phy/upper/channel_processors/pdsch_processor_concurrent_impl.cpp:265,317,342,367), "process_pdcch", "process_ssb",
// "process_nzp_csi_rs" (downlink_processor_single_executor_impl.cpp:83,169,212), "cb_decode"
// (pusch/pusch_decoder_impl.cpp:369); its ru_tracer "downlink_baseband" (R/lib/phy/lower/lower_phy_baseband_processor.cpp:142).
//
// The rocTX library is looked up at run time (no link dependency): ranges are live when the process already has it loaded
// -- a profiler put it there -- or when NRPHY_TRACE=1 asks for it; otherwise a range is one predictable branch.
#pragma once

#include <cstdlib>
#include <dlfcn.h>

namespace nrphy {

class TraceApi
{
public:
  static const TraceApi& get()
  {
    static const TraceApi api;
    return api;
  }
  int (*push)(const char*) = nullptr;
  int (*pop)()             = nullptr;

private:
  TraceApi()
  {
    const char* env   = std::getenv("NRPHY_TRACE");
    const bool  force = env != nullptr && env[0] == '1';
    if (env != nullptr && env[0] == '0') {
      return;
    }
    static const char* const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4",
                                        "libroctx64.so"};
    for (const char* name : names) {
      void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (h == nullptr && force) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      }
      if (h != nullptr) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop  = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push != nullptr && pop != nullptr) {
          return;
        }
        push = nullptr;
        pop  = nullptr;
      }
    }
  }
};

// One range for the lifetime of the object (host side: it brackets the enqueue of the work, which is what rocTX records).
class TraceRange
{
public:
  explicit TraceRange(const char* name) : live(TraceApi::get().push != nullptr)
  {
    if (live) {
      TraceApi::get().push(name);
    }
  }
  ~TraceRange()
  {
    if (live) {
      TraceApi::get().pop();
    }
  }
  TraceRange(const TraceRange&)            = delete;
  TraceRange& operator=(const TraceRange&) = delete;

private:
  bool live;
};

// true when ranges are being recorded (tests)
inline bool trace_enabled()
{
  return TraceApi::get().push != nullptr;
}

} // namespace nrphy
