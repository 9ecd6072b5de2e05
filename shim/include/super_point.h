// Drop-in replacement for AirSLAM include/super_point.h (public surface of :22-26 kept).
#ifndef SUPER_POINT_H_
#define SUPER_POINT_H_

#include <Eigen/Core>
#include <memory>
#include <opencv2/opencv.hpp>
#include <string>
#include <vector>

#include "airfe.h"
#include "read_configs.h"

class SuperPoint {
 public:
  explicit SuperPoint(const SuperPointConfig& super_point_config);
  ~SuperPoint();

  bool build();
  bool infer(const cv::Mat& image, Eigen::Matrix<float, 259, Eigen::Dynamic>& features);
  void save_engine() {}
  bool deserialize_engine() { return false; }

 private:
  SuperPointConfig super_point_config_;
  airfe_ctx* ctx_ = nullptr;
  std::vector<float> feat_;
};

typedef std::shared_ptr<SuperPoint> SuperPointPtr;
#endif  // SUPER_POINT_H_
