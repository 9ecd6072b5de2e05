// TensorRT-free SuperGlue wrapper: replaces src/super_glue.cpp of the reference (Sinkhorn and decode run on the device).
#include "super_glue.h"

#include <iostream>
#include <vector>

#include "airfe_shim_common.h"

SuperGlue::SuperGlue(const PointMatcherConfig& superglue_config) : superglue_config_(superglue_config) {}

SuperGlue::~SuperGlue() { airfe_destroy(ctx_); }

bool SuperGlue::build() {
  airfe_cfg cfg;
  airfe_default_cfg(&cfg);
  cfg.max_batch = 1;
  cfg.max_keypoints = 1024;                                   // TensorRT profile max of the reference (:52-75)
  cfg.matcher = 1;
  cfg.image_width = superglue_config_.image_width;
  cfg.image_height = superglue_config_.image_height;
  const std::string pack = airfe_shim::pack_path(superglue_config_.onnx_file);
  cfg.superglue_pack = pack.c_str();
  if (airfe_create(&cfg, &ctx_) != 0) {
    std::cout << "SuperGlue build failed: " << airfe_last_error(nullptr) << std::endl;
    ctx_ = nullptr;
    return false;
  }
  return true;
}

bool SuperGlue::infer(const Eigen::Matrix<float, 259, Eigen::Dynamic>& features0,
                      const Eigen::Matrix<float, 259, Eigen::Dynamic>& features1, Eigen::VectorXi& indices0,
                      Eigen::VectorXi& indices1, Eigen::VectorXd& mscores0, Eigen::VectorXd& mscores1) {
  if (!ctx_) return false;
  const int n0 = (int)features0.cols(), n1 = (int)features1.cols();
  indices0.resize(n0);
  indices1.resize(n1);
  mscores0.resize(n0);
  mscores1.resize(n1);
  std::vector<int32_t> i0(n0), i1(n1);
  if (airfe_match_superglue(ctx_, features0.data(), n0, features1.data(), n1, i0.data(), i1.data(), mscores0.data(),
                            mscores1.data()) != 0)
    return false;
  for (int i = 0; i < n0; ++i) indices0(i) = i0[i];
  for (int j = 0; j < n1; ++j) indices1(j) = i1[j];
  return true;
}
