// hipcub.hpp of tests/hip_emul -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h): the two device-wide primitives
// colmap_amd/csrc/fusion.hip and ba_schur_explicit.hip use, on the CPU.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "../hip/hip_runtime.h"

namespace hipcub {
struct DeviceScan {
  template <typename In, typename Out>
  static hipError_t ExclusiveSum(void* tmp, size_t& bytes, In in, Out out, int n, hipStream_t = nullptr) {
    if (!tmp) { bytes = 1; return hipSuccess; }
    long long acc = 0;
    for (int i = 0; i < n; ++i) { const auto v = in[i]; out[i] = (decltype(v))acc; acc += v; }
    return hipSuccess;
  }
};
struct DeviceRadixSort {
  template <typename K, typename V>
  static hipError_t SortPairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n, int = 0,
                              int = 8 * (int)sizeof(K), hipStream_t = nullptr) {
    if (!tmp) { bytes = 1; return hipSuccess; }
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return kin[a] < kin[b]; });
    for (int i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
    return hipSuccess;
  }
};
}  // namespace hipcub
