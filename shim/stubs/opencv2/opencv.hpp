// mini-OpenCV: a stand-in for <opencv2/opencv.hpp> — OpenCV is not installed in the build container.
// Only what AirSLAM's front-end sources touch (see shim/stubs/Eigen/Core for why this exists): cv::Mat as a strided 8-bit
// image, Size, Point_, Vec, DMatch, KeyPoint, Scalar, cv::resize for CV_8UC1 / INTER_LINEAR, and a declared
// cv::findFundamentalMat (defined in shim/stubs/mini_opencv.cpp to fail loudly: F-RANSAC stays OpenCV's).
//
// cv::resize here is OUR restatement of OpenCV 4.x's 8-bit bilinear path (imgproc/src/resize.cpp: 11-bit horizontal
// coefficients with cvRound, vertical `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`) — the same
// statement as oracle/ref_post.py::resize_linear_u8; it is NOT OpenCV's code and therefore pins nothing about cv::resize.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5

namespace cv {
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
template <class T>
struct Point_ {
  T x = 0, y = 0;
  Point_() {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <class T>
struct Point3_ {
  T x = 0, y = 0, z = 0;
  Point3_() {}
  Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <class T, int N>
struct Vec {
  T val[N];
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<float, 4> Vec4f;
typedef Vec<float, 6> Vec6f;
struct Scalar {
  double val[4] = {0, 0, 0, 0};
  Scalar() {}
  Scalar(double a, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
};
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 3.402823466e+38f;
  DMatch() {}
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};

// 8-bit single-channel image, rows `step` bytes apart; owns its pixels unless built on a user pointer
struct Mat {
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  size_t step = 0;
  std::shared_ptr<std::vector<uchar>> own;
  Mat() {}
  Mat(int r, int c, int /*type*/) { create(r, c, CV_8UC1); }
  Mat(int r, int c, int /*type*/, void* p, size_t s = 0) : rows(r), cols(c), data((uchar*)p), step(s ? s : (size_t)c) {}
  void create(int r, int c, int /*type*/) {
    own = std::make_shared<std::vector<uchar>>((size_t)r * c);
    rows = r; cols = c; step = (size_t)c; data = own->data();
  }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  void release() { own.reset(); data = nullptr; rows = cols = 0; step = 0; }
  int type() const { return CV_8UC1; }
  int channels() const { return 1; }
  size_t elemSize() const { return 1; }
  bool isContinuous() const { return step == (size_t)cols; }
  uchar* ptr(int r = 0) { return data + (size_t)r * step; }
  const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
  Mat clone() const {
    Mat m(rows, cols, CV_8UC1);
    for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols);
    return m;
  }
};

enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { FM_7POINT = 1, FM_8POINT = 2, FM_LMEDS = 4, FM_RANSAC = 8 };

namespace mini {
inline void resize_coeffs(int dsize, int ssize, std::vector<int>& s0, std::vector<int>& s1, std::vector<int>& a0, std::vector<int>& a1) {
  const double scale = (double)ssize / (double)dsize;
  s0.resize(dsize); s1.resize(dsize); a0.resize(dsize); a1.resize(dsize);
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= (float)s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    a0[d] = (int)std::nearbyint((1.f - f) * 2048.f);        // cvRound: to nearest, ties to even
    a1[d] = (int)std::nearbyint(f * 2048.f);
    s0[d] = s;
    s1[d] = s + 1 < ssize ? s + 1 : ssize - 1;
  }
}
}  // namespace mini

inline void resize(const Mat& src, Mat& dst, Size dsize, double = 0, double = 0, int = INTER_LINEAR) {
  Mat out(dsize.height, dsize.width, CV_8UC1);
  if (src.cols == dsize.width && src.rows == dsize.height) {
    for (int r = 0; r < src.rows; ++r) std::memcpy(out.ptr(r), src.ptr(r), (size_t)src.cols);
    dst = out;
    return;
  }
  std::vector<int> xs, xs1, xa0, xa1, ys, ys1, yb0, yb1;
  mini::resize_coeffs(dsize.width, src.cols, xs, xs1, xa0, xa1);
  mini::resize_coeffs(dsize.height, src.rows, ys, ys1, yb0, yb1);
  std::vector<int> r0((size_t)dsize.width), r1((size_t)dsize.width);
  for (int y = 0; y < dsize.height; ++y) {
    const uchar *p0 = src.ptr(ys[y]), *p1 = src.ptr(ys1[y]);
    for (int x = 0; x < dsize.width; ++x) {
      r0[x] = p0[xs[x]] * xa0[x] + p0[xs1[x]] * xa1[x];
      r1[x] = p1[xs[x]] * xa0[x] + p1[xs1[x]] * xa1[x];
    }
    uchar* o = out.ptr(y);
    for (int x = 0; x < dsize.width; ++x)
      o[x] = (uchar)((((yb0[y] * (r0[x] >> 4)) >> 16) + ((yb1[y] * (r1[x] >> 4)) >> 16) + 2) >> 2);
  }
  dst = out;
}

// cv::FileStorage / cv::FileNode: declared so that DBoW2's TemplatedVocabulary.h (its virtual save / load) compiles unchanged; nothing is stored —
// isOpened() is false, so TemplatedVocabulary::load(filename) throws like it does for a missing file (oracle/ref_bow.cpp builds the tree in memory)
struct FileNode {
  FileNode operator[](const char*) const { return FileNode(); }
  FileNode operator[](const std::string&) const { return FileNode(); }
  FileNode operator[](int) const { return FileNode(); }
  size_t size() const { return 0; }
  operator int() const { return 0; }
  operator double() const { return 0.0; }
  operator std::string() const { return std::string(); }
};
struct FileStorage {
  enum { READ = 0, WRITE = 1 };
  FileStorage() {}
  FileStorage(const std::string&, int) {}
  bool isOpened() const { return false; }
  void release() {}
  FileNode operator[](const char*) const { return FileNode(); }
  FileNode operator[](const std::string&) const { return FileNode(); }
};
template <class T>
inline FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }
template <class M>
inline void eigen2cv(const M&, Mat& dst) { dst = Mat(); }     // FSuperpoint::toMat32F (k-means training helper): float matrices are not modelled

// declared like OpenCV's; the definition (shim/stubs/mini_opencv.cpp) aborts with a message — the F-RANSAC of
// PointMatcher::MatchingPoints(outlier_rejection = true) is OpenCV's and out of scope
Mat findFundamentalMat(const std::vector<Point>& points1, const std::vector<Point>& points2, int method, double ransacReprojThreshold,
                       double confidence, std::vector<uchar>& mask);
}  // namespace cv
