// MINIMAL compile-smoke stand-in for <opencv2/opencv.hpp> (OpenCV is not installed in the build container).
#pragma once
#include <cstddef>
namespace cv {
struct Mat {
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  const unsigned char* ptr(int r) const { return data + (size_t)r * step; }
};
}  // namespace cv
