// One call per stereo keyframe for an AirSLAM tree that has taken INTEGRATION.md's swap: what src/map_builder.cc:85-86 does with
//     _feature_detector->Detect(image_left_rect, image_right_rect, left_features, right_features, left_lines, right_lines, junctions);
//     _point_matcher->MatchingPoints(left_features, right_features, stereo_matches, false);
// — three engine runs behind two facade calls (PLNet::infer on the left image with junctions, on the right one without, feature_detector.cc:97-108;
// LightGlue, point_matcher.cc:50-72) — as ONE queue of device work through airfe_stereo_keyframe (include/airfe.h): both images as one detector batch,
// the line path beside the matcher, two copies back.  Same outputs, bit for bit, as the two facade calls (tests/test_gpu_facade.py runs both).
// Needs ONE context that holds the detector, the stage-1 head and the matcher (the per-class wrappers of shim/src own a context each).
#ifndef AIRFE_KEYFRAME_H_
#define AIRFE_KEYFRAME_H_
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <opencv2/opencv.hpp>

#include "airfe.h"
#include "airfe_shim_common.h"
#include "read_configs.h"

class AirfeStereoKeyframe {
 public:
  AirfeStereoKeyframe(const PLNetConfig& plnet_config, const PointMatcherConfig& matcher_config) : plnet_(plnet_config), matcher_(matcher_config) {}
  ~AirfeStereoKeyframe() { airfe_destroy(ctx_); }
  AirfeStereoKeyframe(const AirfeStereoKeyframe&) = delete;
  AirfeStereoKeyframe& operator=(const AirfeStereoKeyframe&) = delete;

  bool build() {
    if (plnet_.use_superpoint || matcher_.matcher != 0) {
      std::cout << "AirfeStereoKeyframe: the one-call keyframe is PLNet + LightGlue (use_superpoint = 0, matcher = 0)" << std::endl;
      return false;
    }
    airfe_cfg cfg;
    airfe_default_cfg(&cfg);
    cfg.max_batch = 2;                                          // two images per detector pass, two pairs per LightGlue forward
    cfg.enc_chunk = 2;
    cfg.max_keypoints = plnet_.max_keypoints;
    cfg.keypoint_threshold = plnet_.keypoint_threshold;
    cfg.remove_borders = plnet_.remove_borders;
    cfg.line_threshold = plnet_.line_threshold;
    cfg.line_length_threshold = plnet_.line_length_threshold;
    cfg.matcher = 0;
    cfg.image_width = matcher_.image_width;
    cfg.image_height = matcher_.image_height;
    const std::string s0 = airfe_shim::pack_path(plnet_.plnet_s0_onnx), s1 = airfe_shim::pack_path(plnet_.plnet_s1_onnx);
    const std::string lg = airfe_shim::pack_path(matcher_.onnx_file);
    cfg.superpoint_pack = s0.c_str();
    cfg.plnet_s1_pack = s1.c_str();
    cfg.lightglue_pack = lg.c_str();
    if (airfe_create(&cfg, &ctx_) != 0) {
      std::cout << "AirfeStereoKeyframe build failed: " << airfe_last_error(nullptr) << std::endl;
      ctx_ = nullptr;
      return false;
    }
    if (!airfe_has_line_branch(ctx_)) {
      std::cout << "AirfeStereoKeyframe: " << s0 << " has no line branch (line.* tensors)" << std::endl;
      return false;
    }
    cap_ = (cfg.max_keypoints + 15) / 16 * 16;
    fl_.resize((size_t)cap_ * AIRFE_FEAT_DIM); fr_.resize((size_t)cap_ * AIRFE_FEAT_DIM);
    junc_.resize((size_t)2048 * AIRFE_FEAT_DIM);
    ll_.resize((size_t)4096 * 4); lr_.resize((size_t)4096 * 4);
    idx_.resize((size_t)cap_ * 2); score_.resize((size_t)cap_);
    return true;
  }

  // lines are APPENDED (plnet.cpp:544), features / junctions / matches replaced — like the two facade calls.
  // last_keyframe_features + temporal_matches (both or neither): ALSO map_builder.cc:96, MatchingPoints(features_last_keyframe, left_features, matches, ..)
  // up to its RANSAC, as the second pair of the same LightGlue forward; last_keyframe_features == nullptr with temporal_matches != nullptr re-uses the
  // features given last time (they stay on the device).
  bool Process(const cv::Mat& image_left, const cv::Mat& image_right, Eigen::Matrix<float, 259, Eigen::Dynamic>& left_features,
               Eigen::Matrix<float, 259, Eigen::Dynamic>& right_features, std::vector<Eigen::Vector4d>& left_lines,
               std::vector<Eigen::Vector4d>& right_lines, Eigen::Matrix<float, 259, Eigen::Dynamic>& junctions, std::vector<cv::DMatch>& matches,
               const Eigen::Matrix<float, 259, Eigen::Dynamic>* last_keyframe_features = nullptr, std::vector<cv::DMatch>* temporal_matches = nullptr) {
    matches.clear();
    if (temporal_matches) temporal_matches->clear();
    if (!ctx_ || image_left.empty() || image_right.empty() || image_left.rows != image_right.rows || image_left.cols != image_right.cols ||
        image_left.step != image_right.step) {
      std::cout << "Failed when extracting point features !" << std::endl;       // feature_detector.cc:104-106
      return false;
    }
    int nl = 0, nr = 0, nll = 0, nlr = 0, nj = 0, nm = 0, nt = 0;
    int rc;
    if (temporal_matches) {
      tidx_.resize((size_t)cap_ * 2); tscore_.resize((size_t)cap_);
      rc = airfe_stereo_keyframe_tracked(ctx_, image_left.data, image_right.data, image_left.rows, image_left.cols, (int)image_left.step, fl_.data(), fr_.data(),
                                         cap_, &nl, &nr, ll_.data(), lr_.data(), (int)(ll_.size() / 4), &nll, &nlr, junc_.data(),
                                         (int)(junc_.size() / AIRFE_FEAT_DIM), &nj, idx_.data(), score_.data(), cap_, &nm,
                                         last_keyframe_features ? last_keyframe_features->data() : nullptr,
                                         last_keyframe_features ? (int)last_keyframe_features->cols() : 0, tidx_.data(), tscore_.data(), &nt);
    } else {
      rc = airfe_stereo_keyframe(ctx_, image_left.data, image_right.data, image_left.rows, image_left.cols, (int)image_left.step, fl_.data(), fr_.data(), cap_,
                                 &nl, &nr, ll_.data(), lr_.data(), (int)(ll_.size() / 4), &nll, &nlr, junc_.data(), (int)(junc_.size() / AIRFE_FEAT_DIM), &nj,
                                 idx_.data(), score_.data(), cap_, &nm);
    }
    if (rc != 0) {
      std::cout << "Failed when extracting point features ! (" << airfe_last_error(ctx_) << ")" << std::endl;
      return false;
    }
    left_features.resize(259, nl);
    right_features.resize(259, nr);
    junctions.resize(259, nj);
    if (nl) std::memcpy(left_features.data(), fl_.data(), (size_t)nl * AIRFE_FEAT_DIM * sizeof(float));
    if (nr) std::memcpy(right_features.data(), fr_.data(), (size_t)nr * AIRFE_FEAT_DIM * sizeof(float));
    if (nj) std::memcpy(junctions.data(), junc_.data(), (size_t)nj * AIRFE_FEAT_DIM * sizeof(float));
    for (int i = 0; i < nll; ++i) left_lines.emplace_back(ll_[4 * i], ll_[4 * i + 1], ll_[4 * i + 2], ll_[4 * i + 3]);
    for (int i = 0; i < nlr; ++i) right_lines.emplace_back(lr_[4 * i], lr_[4 * i + 1], lr_[4 * i + 2], lr_[4 * i + 3]);
    if (temporal_matches)
      for (int i = 0; i < nt; ++i) temporal_matches->emplace_back(tidx_[2 * i], tidx_[2 * i + 1], 1.0 - tscore_[i]);
    if (nl < 1 || nr < 1) return true;                                                                   // point_matcher.cc:53-55: no matches
    for (int i = 0; i < nm; ++i) matches.emplace_back(idx_[2 * i], idx_[2 * i + 1], 1.0 - score_[i]);   // point_matcher.cc:70
    return true;
  }

 private:
  PLNetConfig plnet_;
  PointMatcherConfig matcher_;
  airfe_ctx* ctx_ = nullptr;
  int cap_ = 0;
  std::vector<float> fl_, fr_, junc_, score_, tscore_;
  std::vector<double> ll_, lr_;
  std::vector<int32_t> idx_, tidx_;
};
#endif
