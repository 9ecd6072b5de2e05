// Drop-in replacement for AirSLAM include/light_glue.h (public surface of :23-31 kept).
#ifndef LIGHT_GLUE_H_
#define LIGHT_GLUE_H_

#include <Eigen/Core>
#include <memory>
#include <string>
#include <vector>

#include "airfe.h"
#include "read_configs.h"

class SuperPointLightGlue {
 public:
  explicit SuperPointLightGlue(const PointMatcherConfig& lightglue_config);
  ~SuperPointLightGlue();

  bool build();
  bool infer(const Eigen::Matrix<float, 258, Eigen::Dynamic>& features0, const Eigen::Matrix<float, 258, Eigen::Dynamic>& features1,
             Eigen::Matrix<int, Eigen::Dynamic, 2>& matches_index, Eigen::Matrix<float, Eigen::Dynamic, 1>& matches_score);
  void save_engine() {}
  bool deserialize_engine() { return false; }

 private:
  PointMatcherConfig lightglue_config_;
  airfe_ctx* ctx_ = nullptr;
  std::vector<int32_t> idx_;
  std::vector<float> score_;
};

typedef std::shared_ptr<SuperPointLightGlue> SuperPointLightGluePtr;
#endif  // LIGHT_GLUE_H_
