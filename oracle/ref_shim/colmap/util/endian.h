// TEST INFRASTRUCTURE (oracle/ref_shim/README.md). The reference's util/endian.h reaches Eigen
// through util/types.h; the only users on this path are GpuMat<T>::Read / Write, which the
// checker never calls. Host is little-endian: plain stream I/O.
#pragma once

#include <iostream>
#include <vector>

namespace colmap {

template <typename T>
void ReadBinaryLittleEndian(std::istream* stream, std::vector<T>* data) {
  stream->read(reinterpret_cast<char*>(data->data()), static_cast<std::streamsize>(data->size() * sizeof(T)));
}

template <typename T>
void WriteBinaryLittleEndian(std::ostream* stream, const std::vector<T>& data) {
  stream->write(reinterpret_cast<const char*>(data.data()), static_cast<std::streamsize>(data.size() * sizeof(T)));
}

}  // namespace colmap
