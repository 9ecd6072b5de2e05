// TensorRT-free SuperPoint wrapper: replaces src/super_point.cpp of the reference.
#include "super_point.h"

#include <cstring>
#include <iostream>

#include "airfe_shim_common.h"

SuperPoint::SuperPoint(const SuperPointConfig& super_point_config) : super_point_config_(super_point_config) {}

SuperPoint::~SuperPoint() { airfe_destroy(ctx_); }

bool SuperPoint::build() {
  airfe_cfg cfg;
  airfe_default_cfg(&cfg);
  cfg.max_batch = 1;
  cfg.enc_chunk = 1;
  cfg.max_keypoints = super_point_config_.max_keypoints;
  cfg.keypoint_threshold = super_point_config_.keypoint_threshold;
  cfg.remove_borders = super_point_config_.remove_borders;
  const std::string pack = airfe_shim::pack_path(super_point_config_.onnx_file);
  cfg.superpoint_pack = pack.c_str();
  if (airfe_create(&cfg, &ctx_) != 0) {
    std::cout << "SuperPoint build failed: " << airfe_last_error(nullptr) << std::endl;
    ctx_ = nullptr;
    return false;
  }
  feat_.resize((size_t)((cfg.max_keypoints + 63) / 64 * 64) * AIRFE_FEAT_DIM);
  return true;
}

bool SuperPoint::infer(const cv::Mat& image, Eigen::Matrix<float, 259, Eigen::Dynamic>& features) {
  if (!ctx_ || image.empty()) return false;
  int n = 0;
  if (airfe_detect_points(ctx_, image.data, image.rows, image.cols, (int)image.step, feat_.data(),
                          (int)(feat_.size() / AIRFE_FEAT_DIM), &n) != 0)
    return false;
  features.resize(259, n);
  if (n) std::memcpy(features.data(), feat_.data(), (size_t)n * AIRFE_FEAT_DIM * sizeof(float));
  return true;
}
