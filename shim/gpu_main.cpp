// The drop-in wrappers EXECUTED on a GPU box (tests/test_gpu_shim.py): constructs SuperPoint, PLNet, SuperPointLightGlue and SuperGlue
// exactly as FeatureDetector / PointMatcher do (src/feature_detector.cc:7-34, src/point_matcher.cc:6-37), runs infer() on a stereo
// pair read from raw files and dumps every output; the test compares the bytes with the ctypes path through the same C ABI.
//   gpu_main <model dir> <left.raw> <right.raw> <h> <w> <out dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "light_glue.h"
#include "plnet.h"
#include "super_glue.h"
#include "super_point.h"

static std::vector<unsigned char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T>
static void dump(const std::string& p, const T* d, size_t n) {
  std::ofstream f(p, std::ios::binary);
  f.write(reinterpret_cast<const char*>(d), (std::streamsize)(n * sizeof(T)));
}

// PointMatcher::NormalizeKeypoints (src/point_matcher.cc:39-48), statement for statement, on the 259 x N matrix; `skip_score` drops
// row 0 (the bottomRows(258) block LightGlue is handed at :67)
template <int ROWS>
static Eigen::Matrix<float, ROWS, Eigen::Dynamic> normalised(const Eigen::Matrix<float, 259, Eigen::Dynamic>& f, int width, int height, float scale) {
  Eigen::Matrix<float, ROWS, Eigen::Dynamic> o;
  o.resize(ROWS, f.cols());
  const float L_inv = 1.0 / std::max(width, height) * scale;
  const int skip = 259 - ROWS;
  for (long c = 0; c < f.cols(); ++c)
    for (int r = skip; r < 259; ++r) {
      float v = f(r, c);
      if (r == 1) v = (v - width / 2) * L_inv;
      if (r == 2) v = (v - height / 2) * L_inv;
      o(r - skip, c) = v;
    }
  return o;
}

int main(int argc, char** argv) {
  if (argc != 7) return 2;
  const std::string md = argv[1], od = argv[6];
  const int h = std::atoi(argv[4]), w = std::atoi(argv[5]);
  std::vector<unsigned char> L = slurp(argv[2]), R = slurp(argv[3]);
  if ((int)L.size() != h * w || (int)R.size() != h * w) return 3;
  cv::Mat left, right;
  left.rows = right.rows = h; left.cols = right.cols = w; left.step = right.step = (size_t)w;
  left.data = L.data(); right.data = R.data();

  SuperPointConfig sc;
  sc.onnx_file = md + "/superpoint_v1_sim_int32.onnx";
  SuperPointPtr sp(new SuperPoint(sc));
  if (!sp->build()) return 10;
  Eigen::Matrix<float, 259, Eigen::Dynamic> f0, f1;
  if (!sp->infer(left, f0) || !sp->infer(right, f1)) return 11;
  dump(od + "/sp_f0.bin", f0.data(), (size_t)f0.size());
  dump(od + "/sp_f1.bin", f1.data(), (size_t)f1.size());

  PLNetConfig pc;
  pc.plnet_s0_onnx = md + "/plnet_s0.onnx";
  pc.plnet_s1_onnx = md + "/plnet_s1.onnx";
  pc.line_threshold = 0.5f; pc.line_length_threshold = 4.f;
  PLNetPtr pl(new PLNet(pc));
  if (!pl->build()) return 20;
  Eigen::Matrix<float, 259, Eigen::Dynamic> pf, pj;
  std::vector<Eigen::Vector4d> lines;
  lines.emplace_back(1.0, 2.0, 3.0, 4.0);                        // infer() appends, never clears (src/plnet.cpp:544)
  if (!pl->infer(left, pf, lines, pj, true)) return 21;
  dump(od + "/pl_feat.bin", pf.data(), (size_t)pf.size());
  dump(od + "/pl_junc.bin", pj.data(), (size_t)pj.size());
  std::vector<double> lv;
  for (auto& l : lines) for (int k = 0; k < 4; ++k) lv.push_back(l(k));
  dump(od + "/pl_lines.bin", lv.data(), lv.size());

  PointMatcherConfig mc;
  mc.image_width = w; mc.image_height = h;
  mc.onnx_file = md + "/superpoint_lightglue.onnx";
  SuperPointLightGluePtr lg(new SuperPointLightGlue(mc));
  if (!lg->build()) return 30;
  Eigen::Matrix<int, Eigen::Dynamic, 2> midx;
  Eigen::Matrix<float, Eigen::Dynamic, 1> mscore;
  if (!lg->infer(normalised<258>(f0, w, h, 0.5f), normalised<258>(f1, w, h, 0.5f), midx, mscore)) return 31;
  std::vector<int> iv; std::vector<float> sv;
  for (long i = 0; i < midx.rows(); ++i) { iv.push_back(midx(i, 0)); iv.push_back(midx(i, 1)); sv.push_back(mscore(i)); }
  dump(od + "/lg_idx.bin", iv.data(), iv.size());
  dump(od + "/lg_score.bin", sv.data(), sv.size());

  mc.matcher = 1;
  mc.onnx_file = md + "/superglue_outdoor_sim_int32.onnx";
  SuperGluePtr sg(new SuperGlue(mc));
  if (!sg->build()) return 40;
  Eigen::VectorXi i0, i1; Eigen::VectorXd m0, m1;
  if (!sg->infer(normalised<259>(f0, w, h, 0.7f), normalised<259>(f1, w, h, 0.7f), i0, i1, m0, m1)) return 41;
  dump(od + "/sg_i0.bin", i0.data(), (size_t)i0.size());
  dump(od + "/sg_i1.bin", i1.data(), (size_t)i1.size());
  dump(od + "/sg_m0.bin", m0.data(), (size_t)m0.size());
  dump(od + "/sg_m1.bin", m1.data(), (size_t)m1.size());
  std::printf("shim gpu: %ld/%ld keypoints, %ld PLNet points, %zu lines, %ld junctions, %ld LightGlue matches\n", f0.cols(), f1.cols(),
              pf.cols(), lines.size(), pj.cols(), midx.rows());
  return 0;
}
