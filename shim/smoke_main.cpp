// Compile/link smoke of the shim against the stub headers: constructs the four wrappers exactly as
// FeatureDetector / PointMatcher do (src/feature_detector.cc:7-34, src/point_matcher.cc:6-37) and checks that,
// without a GPU or weight packs, build() fails cleanly (returns false, prints, never throws).
#include <cstdio>

#include "light_glue.h"
#include "plnet.h"
#include "super_glue.h"
#include "super_point.h"

int main() {
  PLNetConfig pc;
  pc.plnet_s0_onnx = "/nonexistent/plnet_s0.onnx";
  pc.plnet_s1_onnx = "/nonexistent/plnet_s1.onnx";
  SuperPointConfig sc;
  sc.onnx_file = "/nonexistent/superpoint_v1_sim_int32.onnx";
  PointMatcherConfig mc;
  mc.onnx_file = "/nonexistent/superpoint_lightglue.onnx";
  PLNetPtr plnet(new PLNet(pc));
  SuperPointPtr sp(new SuperPoint(sc));
  SuperPointLightGluePtr lg(new SuperPointLightGlue(mc));
  SuperGluePtr sg(new SuperGlue(mc));
  const bool b = plnet->build() | sp->build() | lg->build() | sg->build();
  Eigen::Matrix<float, 259, Eigen::Dynamic> f, j;
  std::vector<Eigen::Vector4d> lines;
  cv::Mat empty;
  const bool r = plnet->infer(empty, f, lines, j, true) | sp->infer(empty, f);
  std::printf("shim smoke: build=%d infer=%d\n", (int)b, (int)r);
  return (b || r) ? 1 : 0;      // on a box with no packs everything must fail cleanly
}
