// Stand-in for the reference's include/utils.h (which needs Boost.Serialization, g2o and OpenCV's highgui — absent here,
// SURVEY.md Appendix E): only the declarations the front-end sources and include/read_configs.h use, with the
// reference's signatures (include/utils.h:39-45,71-79); definitions in shim/stubs/mini_utils.cpp follow src/utils.cc.
#ifndef UTILS_H_
#define UTILS_H_
#include <Eigen/Core>
#include <Eigen/StdVector>
#include <iostream>
#include <map>
#include <memory>
#include <opencv2/opencv.hpp>
#include <string>
#include <vector>

typedef Eigen::Matrix<double, 5, 1> Vector5d;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
typedef Eigen::Matrix<double, 8, 1> Vector8d;
typedef Eigen::Matrix<double, 9, 1> Vector9d;

bool FileExists(const std::string& file);
bool PathExists(const std::string& path);
void ConcatenateFolderAndFileName(const std::string& folder, const std::string& file_name, std::string* path);
std::string ConcatenateFolderAndFileName(const std::string& folder, const std::string& file_name);
#endif  // UTILS_H_
