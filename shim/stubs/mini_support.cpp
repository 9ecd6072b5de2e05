// Definitions behind the stand-in headers of shim/stubs/ (test infrastructure; see Eigen/Core for why they exist).
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>

#include "utils.h"

// ---- the four path helpers of the reference's src/utils.cc:177-211 (behaviour restated: regular file / directory tests, '/'-joined path)
bool FileExists(const std::string& file) {
  struct stat st;
  return stat(file.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
bool PathExists(const std::string& path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
void ConcatenateFolderAndFileName(const std::string& folder, const std::string& file_name, std::string* path) {
  *path = folder;
  if (path->empty() || path->back() != '/') *path += '/';
  *path += file_name;
}
std::string ConcatenateFolderAndFileName(const std::string& folder, const std::string& file_name) {
  std::string p;
  ConcatenateFolderAndFileName(folder, file_name, &p);
  return p;
}

// ---- cv::findFundamentalMat: OpenCV's RANSAC is not restated; MatchingPoints(outlier_rejection = true) needs real OpenCV
namespace cv {
Mat findFundamentalMat(const std::vector<Point>&, const std::vector<Point>&, int, double, double, std::vector<uchar>&) {
  std::fprintf(stderr, "mini-OpenCV: cv::findFundamentalMat is not available (F-RANSAC stays OpenCV's; call MatchingPoints with outlier_rejection = false)\n");
  std::abort();
}
}  // namespace cv
