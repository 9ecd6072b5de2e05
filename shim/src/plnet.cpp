// TensorRT-free PLNet wrapper: replaces src/plnet.cpp of the reference with calls into libairfe.so.
// Everything the reference did on the host after a 15.5 MB D2H copy (src/plnet.cpp:237,450-585) now runs on the GPU;
// what crosses the boundary is the final 259 x N matrix, the line list and (optionally) the junction matrix.
#include "plnet.h"

#include <cstring>
#include <iostream>

#include "airfe_shim_common.h"

PLNet::PLNet(PLNetConfig& plnet_config) : plnet_config_(plnet_config) {}

PLNet::~PLNet() { airfe_destroy(ctx_); }

bool PLNet::build() {
  airfe_cfg cfg;
  airfe_default_cfg(&cfg);
  cfg.max_batch = 1;
  cfg.enc_chunk = 1;
  cfg.max_keypoints = plnet_config_.max_keypoints;
  cfg.keypoint_threshold = plnet_config_.keypoint_threshold;
  cfg.remove_borders = plnet_config_.remove_borders;
  cfg.line_threshold = plnet_config_.line_threshold;
  cfg.line_length_threshold = plnet_config_.line_length_threshold;
  const std::string s0 = airfe_shim::pack_path(plnet_config_.plnet_s0_onnx);   // stage 0: VGG trunk + point heads + line branch (line.*)
  const std::string s1 = airfe_shim::pack_path(plnet_config_.plnet_s1_onnx);   // LOI line-verification head
  cfg.superpoint_pack = s0.c_str();
  cfg.plnet_s1_pack = s1.c_str();
  if (airfe_create(&cfg, &ctx_) != 0) {
    std::cout << "PLNet build failed: " << airfe_last_error(nullptr) << std::endl;
    ctx_ = nullptr;
    return false;
  }
  // The stage-0 pack normally carries the line branch (tensors line.conv1.*, line.head.*) and infer() then needs nothing from the
  // host.  A point-only pack leaves PLNet::infer without lines and junctions: the reference's callers (feature_detector.cc:52-69,
  // map_builder.cc) would silently turn into a point-only SLAM, so say it LOUDLY, once, here.
  has_lines_ = airfe_has_line_branch(ctx_) != 0;
  if (!has_lines_)
    std::cout << "PLNet: " << s0 << " has no line branch (line.* tensors): infer() will return POINTS ONLY — no lines, no junctions — "
              << "unless set_stage0_provider() supplies the stage-0 line tensors." << std::endl;
  const int cap = (cfg.max_keypoints + 63) / 64 * 64;
  feat_.resize((size_t)cap * AIRFE_FEAT_DIM);
  junc_.resize((size_t)2048 * AIRFE_FEAT_DIM);
  lines_.resize((size_t)45056 * 4);                                 // 300 junctions: at most 44850 unique lines
  return true;
}

bool PLNet::infer(const cv::Mat& image, Eigen::Matrix<float, 259, Eigen::Dynamic>& features,
                  std::vector<Eigen::Vector4d>& lines, Eigen::Matrix<float, 259, Eigen::Dynamic>& junctions,
                  bool junction_detection) {
  if (!ctx_ || image.empty()) return false;                     // reference: process_image returns false on empty
  const airfe_plnet_stage0* s0 = s0_fn_ ? s0_fn_(image, s0_user_) : nullptr;
  int n = 0, nl = 0, nj = 0;
  const int cap = (int)(feat_.size() / AIRFE_FEAT_DIM);
  if (airfe_detect_plnet(ctx_, image.data, image.rows, image.cols, (int)image.step, s0, feat_.data(), cap, &n,
                         lines_.data(), (int)(lines_.size() / 4), &nl, junc_.data(), (int)(junc_.size() / AIRFE_FEAT_DIM), &nj,
                         junction_detection ? 1 : 0) != 0)
    return false;
  features.resize(259, n);                                        // column-major 259 x N == n rows of 259 floats
  if (n) std::memcpy(features.data(), feat_.data(), (size_t)n * AIRFE_FEAT_DIM * sizeof(float));
  for (int i = 0; i < nl; ++i)                                    // appended, never cleared (src/plnet.cpp:544)
    lines.emplace_back(lines_[4 * i], lines_[4 * i + 1], lines_[4 * i + 2], lines_[4 * i + 3]);
  if (junction_detection) {
    junctions.resize(259, nj);
    if (nj) std::memcpy(junctions.data(), junc_.data(), (size_t)nj * AIRFE_FEAT_DIM * sizeof(float));
  }
  return true;
}
