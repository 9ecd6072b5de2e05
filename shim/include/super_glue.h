// Drop-in replacement for AirSLAM include/super_glue.h (public surface of :24-33 kept).
#ifndef SUPER_GLUE_H_
#define SUPER_GLUE_H_

#include <Eigen/Core>
#include <memory>
#include <string>
#include <vector>

#include "airfe.h"
#include "read_configs.h"

class SuperGlue {
 public:
  explicit SuperGlue(const PointMatcherConfig& superglue_config);
  ~SuperGlue();

  bool build();
  bool infer(const Eigen::Matrix<float, 259, Eigen::Dynamic>& features0, const Eigen::Matrix<float, 259, Eigen::Dynamic>& features1,
             Eigen::VectorXi& indices0, Eigen::VectorXi& indices1, Eigen::VectorXd& mscores0, Eigen::VectorXd& mscores1);
  void save_engine() {}
  bool deserialize_engine() { return false; }

 private:
  PointMatcherConfig superglue_config_;
  airfe_ctx* ctx_ = nullptr;
};

typedef std::shared_ptr<SuperGlue> SuperGluePtr;
#endif  // SUPER_GLUE_H_
