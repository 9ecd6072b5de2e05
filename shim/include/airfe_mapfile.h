// AirSLAM's feature / line records without Boost: the exact bytes `SerializeFeatures`
// (include/utils.h:205-222) and `SerializeEigenVector4dList` (include/utils.h:184-202) put into a boost::archive::binary_oarchive —
// primitives and make_array() payloads are written raw, little-endian — so a record can be copied between a reference map file
// and the [N][259] float rows of include/airfe.h with one read.  Python twin: airslam_amd/mapfile.py.
#ifndef AIRFE_MAPFILE_H_
#define AIRFE_MAPFILE_H_

#include <cstdint>
#include <istream>
#include <ostream>
#include <vector>

namespace airfe_mapfile {

constexpr int32_t kRows = 259;

// n feature rows [n][259] (= Eigen::Matrix<float,259,Dynamic>::data() of a 259 x n matrix) -> one record
inline void write_features(std::ostream& os, const float* rows, int32_t n) {
  os.write(reinterpret_cast<const char*>(&n), 4);
  os.write(reinterpret_cast<const char*>(&kRows), 4);
  os.write(reinterpret_cast<const char*>(rows), (std::streamsize)n * kRows * 4);
}

// -> false when the stream does not hold a feature record here; rows is resized to n * 259
inline bool read_features(std::istream& is, std::vector<float>& rows, int32_t& n) {
  int32_t cols = 0, r = 0;
  if (!is.read(reinterpret_cast<char*>(&cols), 4) || !is.read(reinterpret_cast<char*>(&r), 4) || r != kRows || cols < 0) return false;
  rows.resize((size_t)cols * kRows);
  n = cols;
  return cols == 0 || (bool)is.read(reinterpret_cast<char*>(rows.data()), (std::streamsize)cols * kRows * 4);
}

// std::vector<Eigen::Vector4d> storage ([l][4] doubles) -> one record
inline void write_lines(std::ostream& os, const double* xyxy, int32_t l) {
  os.write(reinterpret_cast<const char*>(&l), 4);
  os.write(reinterpret_cast<const char*>(xyxy), (std::streamsize)l * 32);
}

inline bool read_lines(std::istream& is, std::vector<double>& xyxy, int32_t& l) {
  if (!is.read(reinterpret_cast<char*>(&l), 4) || l < 0) return false;
  xyxy.resize((size_t)l * 4);
  return l == 0 || (bool)is.read(reinterpret_cast<char*>(xyxy.data()), (std::streamsize)l * 32);
}

}  // namespace airfe_mapfile
#endif  // AIRFE_MAPFILE_H_
