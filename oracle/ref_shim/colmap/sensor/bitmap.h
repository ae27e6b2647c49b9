// TEST INFRASTRUCTURE (oracle/ref_shim/README.md). Minimal grey-scale stand-in for the reference's
// OpenImageIO-backed Bitmap (src/colmap/sensor/bitmap.h): the PatchMatch path only reads
// Width / Height / IsEmpty / RowMajorData / NumBytes (patch_match_cuda.cu:1586,1619-1621,
// mvs/image.cc:59-75).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <vector>

namespace colmap {

class Bitmap {
 public:
  Bitmap() = default;
  Bitmap(int width, int height, const uint8_t* gray)
      : width_(width), height_(height), data_(gray, gray + static_cast<size_t>(width) * height) {}

  int Width() const { return width_; }
  int Height() const { return height_; }
  int Channels() const { return 1; }
  bool IsEmpty() const { return data_.empty(); }
  size_t NumBytes() const { return data_.size(); }
  std::vector<uint8_t>& RowMajorData() { return data_; }
  const std::vector<uint8_t>& RowMajorData() const { return data_; }
  void Rescale(int, int) { throw std::logic_error("ref_shim::Bitmap::Rescale is not available"); }

 private:
  int width_ = 0, height_ = 0;
  std::vector<uint8_t> data_;
};

}  // namespace colmap
