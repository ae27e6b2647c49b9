// TEST INFRASTRUCTURE (oracle/ref_shim/README.md): only here so that mvs/patch_match.h parses.
#pragma once
#include <mutex>
namespace colmap {
class ThreadPool;
}  // namespace colmap
