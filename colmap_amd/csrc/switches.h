// Development switches: kernel variants kept for A/B comparisons and the tests that prove them equal.
//
// The shipped launch path reads NO environment. A switch has its built-in default unless the process sets it through the
// C ABI (colmap_amd_set_switch, include/colmap_amd_pm.h) -- which is what tests/ and bench.py's A/B legs do. Only a
// library built with -DCOLMAP_AMD_ENV_SWITCHES (scripts/ profiling sessions: `python -m colmap_amd.build --dev`)
// also consults getenv() under the same names. Diagnostics whose results are garbage (PatchMatch ablations, fixed
// perturbation) are compiled in only with -DCOLMAP_AMD_DIAG_BUILD (pm_internal.h).
#pragma once

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace colmap_amd {

struct DevSwitchTable {
  std::mutex mu;
  std::map<std::string, std::string> values;
  std::atomic<int> count{0};  // entries of `values`: a process that never set a switch (the product) never takes the lock
  static DevSwitchTable& Get() {
    static DevSwitchTable* t = new DevSwitchTable();  // leaked on purpose: read from static destructors' launch paths
    return *t;
  }
};

// Value of a switch as a string; false when it is unset.
inline bool dev_switch(const char* name, std::string* out) {
  if (DevSwitchTable& t = DevSwitchTable::Get(); t.count.load(std::memory_order_acquire) != 0) {
    std::lock_guard<std::mutex> lock(t.mu);
    auto it = t.values.find(name);
    if (it != t.values.end()) {
      *out = it->second;
      return true;
    }
  }
#ifdef COLMAP_AMD_ENV_SWITCHES
  if (const char* e = std::getenv(name)) {
    *out = e;
    return true;
  }
#endif
  return false;
}

inline int dev_switch_int(const char* name, int dflt) {
  std::string v;
  return dev_switch(name, &v) && !v.empty() ? std::atoi(v.c_str()) : dflt;
}

inline double dev_switch_double(const char* name, double dflt) {
  std::string v;
  return dev_switch(name, &v) && !v.empty() ? std::atof(v.c_str()) : dflt;
}

}  // namespace colmap_amd

// One definition per shared object (weak: every translation unit that includes this header carries it).
extern "C" __attribute__((weak, visibility("default"))) void colmap_amd_set_switch(const char* name, const char* value) {
  if (!name) return;
  colmap_amd::DevSwitchTable& t = colmap_amd::DevSwitchTable::Get();
  std::lock_guard<std::mutex> lock(t.mu);
  if (value) t.values[name] = value;
  else t.values.erase(name);
  t.count.store((int)t.values.size(), std::memory_order_release);
}
