// TEST INFRASTRUCTURE (oracle/ref_shim/README.md): only here so that mvs/patch_match.h parses.
#pragma once
namespace colmap {
class BaseController {
 public:
  virtual ~BaseController() = default;
};
}  // namespace colmap
