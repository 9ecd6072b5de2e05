// TensorRT-free LightGlue wrapper: replaces src/light_glue.cpp of the reference.  The repack loops of
// process_input (:172-212) and the O(N0*N1) host filter_matches (:214-266) are gone: the 258 x N column-major
// Eigen buffers ARE the [N][258] row layout the library consumes, and filtering happens on the device.
#include "light_glue.h"

#include <iostream>

#include "airfe_shim_common.h"

SuperPointLightGlue::SuperPointLightGlue(const PointMatcherConfig& lightglue_config) : lightglue_config_(lightglue_config) {}

SuperPointLightGlue::~SuperPointLightGlue() { airfe_destroy(ctx_); }

bool SuperPointLightGlue::build() {
  airfe_cfg cfg;
  airfe_default_cfg(&cfg);
  cfg.max_batch = 1;
  cfg.max_keypoints = 1024;                                   // TensorRT profile max of the reference (:50-64)
  cfg.matcher = 0;
  cfg.image_width = lightglue_config_.image_width;
  cfg.image_height = lightglue_config_.image_height;
  const std::string pack = airfe_shim::pack_path(lightglue_config_.onnx_file);
  cfg.lightglue_pack = pack.c_str();
  if (airfe_create(&cfg, &ctx_) != 0) {
    std::cout << "LightGlue build failed: " << airfe_last_error(nullptr) << std::endl;
    ctx_ = nullptr;
    return false;
  }
  idx_.resize(2 * 1024);
  score_.resize(1024);
  return true;
}

bool SuperPointLightGlue::infer(const Eigen::Matrix<float, 258, Eigen::Dynamic>& features0,
                                const Eigen::Matrix<float, 258, Eigen::Dynamic>& features1,
                                Eigen::Matrix<int, Eigen::Dynamic, 2>& matches_index,
                                Eigen::Matrix<float, Eigen::Dynamic, 1>& matches_score) {
  if (!ctx_) return false;
  int nm = 0;
  if (airfe_match_lightglue(ctx_, features0.data(), (int)features0.cols(), features1.data(), (int)features1.cols(), idx_.data(),
                            score_.data(), 1024, &nm) != 0)
    return false;
  matches_index.resize(nm, 2);
  matches_score.resize(nm, 1);
  for (int i = 0; i < nm; ++i) {
    matches_index(i, 0) = idx_[2 * i];
    matches_index(i, 1) = idx_[2 * i + 1];
    matches_score(i) = score_[i];
  }
  return true;
}
