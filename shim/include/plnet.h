// Drop-in replacement for AirSLAM include/plnet.h: same class name and public signatures
// (include/plnet.h:17-28 of the reference), no TensorRT / CUDA / tensorrtbuffer includes.
#ifndef PLNET_PLNET_H
#define PLNET_PLNET_H

#include <Eigen/Core>
#include <memory>
#include <opencv2/opencv.hpp>
#include <string>
#include <vector>

#include "airfe.h"
#include "read_configs.h"

class PLNet {
 public:
  PLNet(PLNetConfig& plnet_config);
  ~PLNet();

  bool build();

  bool infer(const cv::Mat& image, Eigen::Matrix<float, 259, Eigen::Dynamic>& features,
             std::vector<Eigen::Vector4d>& lines, Eigen::Matrix<float, 259, Eigen::Dynamic>& junctions,
             bool junction_detection = false);

  void save_engine() {}                         // weight packing is redone at build(); nothing to cache
  bool deserialize_engine() { return false; }

  // Optional HOST provider of the stage-0 line-branch tensors, overriding the on-device line branch
  // that a stage-0 pack with line.* tensors gives (known-answer tests of the downstream stages use it).
  void set_stage0_provider(const airfe_plnet_stage0* (*fn)(const cv::Mat&, void*), void* user) { s0_fn_ = fn; s0_user_ = user; }

 private:
  PLNetConfig plnet_config_;
  airfe_ctx* ctx_ = nullptr;
  bool has_lines_ = false;
  const airfe_plnet_stage0* (*s0_fn_)(const cv::Mat&, void*) = nullptr;
  void* s0_user_ = nullptr;
  std::vector<float> feat_, junc_;
  std::vector<double> lines_;
};

typedef std::shared_ptr<PLNet> PLNetPtr;
#endif  // PLNET_PLNET_H
