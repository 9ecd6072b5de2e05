// The reference's façade — src/feature_detector.cc and src/point_matcher.cc, compiled UNCHANGED from /root/reference (shim/Makefile) —
// on top of the TensorRT-free wrappers of shim/src/ and libairfe.so, EXECUTED on a GPU box (tests/test_gpu_facade.py):
// the configuration is read by the reference's own include/read_configs.h from a YAML file in the reference's format, all six
// FeatureDetector::Detect overloads (src/feature_detector.cc:36,52,62,71,83,97) and PointMatcher::MatchingPoints
// (src/point_matcher.cc:50-107, outlier_rejection = false) run on a stereo pair read from raw files, every output is dumped.
//   facade_gpu <config.yaml> <model dir> <left.raw> <right.raw> <h> <w> <out dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "airfe_keyframe.h"
#include "feature_detector.h"
#include "point_matcher.h"
#include "read_configs.h"

typedef Eigen::Matrix<float, 259, Eigen::Dynamic> Features;

static std::vector<unsigned char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T>
static void dump(const std::string& p, const T* d, size_t n) {
  std::ofstream f(p, std::ios::binary);
  f.write(reinterpret_cast<const char*>(d), (std::streamsize)(n * sizeof(T)));
}
static void dump_features(const std::string& p, const Features& f) { dump(p, f.data(), (size_t)f.size()); }
static void dump_lines(const std::string& p, const std::vector<Eigen::Vector4d>& l) {
  std::vector<double> v;
  for (const auto& x : l)
    for (int k = 0; k < 4; ++k) v.push_back(x(k));
  dump(p, v.data(), v.size());
}

int main(int argc, char** argv) {
  if (argc != 8) return 2;
  const std::string yaml = argv[1], md = argv[2], od = argv[7];
  const int h = std::atoi(argv[5]), w = std::atoi(argv[6]);
  std::vector<unsigned char> L = slurp(argv[3]), R = slurp(argv[4]);
  if ((int)L.size() != h * w || (int)R.size() != h * w) return 3;
  cv::Mat left(h, w, CV_8UC1, L.data(), (size_t)w), right(h, w, CV_8UC1, R.data(), (size_t)w);

  VisualOdometryConfigs cfgs(yaml, md);                       // the reference's loader: plnet block, SetModelPath, point_matcher block
  FeatureDetector det(cfgs.plnet_config);
  PointMatcher pm(cfgs.point_matcher_config);

  bool ok = true;
  {
    Features f;
    ok &= det.Detect(left, f);
    dump_features(od + "/d0_feat.bin", f);
  }
  {
    Features f;
    std::vector<Eigen::Vector4d> l;
    ok &= det.Detect(left, f, l);
    dump_features(od + "/d1_feat.bin", f);
    dump_lines(od + "/d1_lines.bin", l);
  }
  {
    Features f, j;
    std::vector<Eigen::Vector4d> l;
    ok &= det.Detect(left, f, l, j);
    dump_features(od + "/d2_feat.bin", f);
    dump_lines(od + "/d2_lines.bin", l);
    dump_features(od + "/d2_junc.bin", j);
  }
  Features fl, fr;
  {
    ok &= det.Detect(left, right, fl, fr);
    dump_features(od + "/d3_featl.bin", fl);
    dump_features(od + "/d3_featr.bin", fr);
  }
  {
    Features a, b;
    std::vector<Eigen::Vector4d> la, lb;
    ok &= det.Detect(left, right, a, b, la, lb);
    dump_features(od + "/d4_featl.bin", a);
    dump_features(od + "/d4_featr.bin", b);
    dump_lines(od + "/d4_linesl.bin", la);
    dump_lines(od + "/d4_linesr.bin", lb);
  }
  {
    Features a, b, j;
    std::vector<Eigen::Vector4d> la, lb;
    ok &= det.Detect(left, right, a, b, la, lb, j);
    dump_features(od + "/d5_featl.bin", a);
    dump_features(od + "/d5_featr.bin", b);
    dump_lines(od + "/d5_linesl.bin", la);
    dump_lines(od + "/d5_linesr.bin", lb);
    dump_features(od + "/d5_junc.bin", j);
  }
  if (!ok) return 10;
  cv::Mat none;
  Features fe;
  if (det.Detect(none, fe)) return 11;                        // empty image -> false (src/plnet.cpp:247, src/feature_detector.cc:46-48)

  std::vector<cv::DMatch> matches;
  const int n = pm.MatchingPoints(fl, fr, matches, false);
  if (n != (int)matches.size()) return 12;
  std::vector<int> q, t;
  std::vector<float> d;
  for (const auto& m : matches) { q.push_back(m.queryIdx); t.push_back(m.trainIdx); d.push_back(m.distance); }
  dump(od + "/m_query.bin", q.data(), q.size());
  dump(od + "/m_train.bin", t.data(), t.size());
  dump(od + "/m_dist.bin", d.data(), d.size());
  Features empty;
  empty.resize(259, 0);
  if (pm.MatchingPoints(empty, fr, matches, false) != 0) return 13;      // src/point_matcher.cc:53-55
  // the one-call keyframe (shim/include/airfe_keyframe.h) beside the two facade calls it stands for (overload 6 + MatchingPoints): PLNet + LightGlue only
  if (!cfgs.plnet_config.use_superpoint && cfgs.point_matcher_config.matcher == 0) {
    AirfeStereoKeyframe kf(cfgs.plnet_config, cfgs.point_matcher_config);
    if (!kf.build()) return 14;
    Features a, b, j;
    std::vector<Eigen::Vector4d> la, lb;
    std::vector<cv::DMatch> km;
    for (int rep = 0; rep < 2; ++rep) {
      la.clear(); lb.clear();
      if (!kf.Process(left, right, a, b, la, lb, j, km)) return 15;
    }
    dump_features(od + "/k_featl.bin", a);
    dump_features(od + "/k_featr.bin", b);
    dump_lines(od + "/k_linesl.bin", la);
    dump_lines(od + "/k_linesr.bin", lb);
    dump_features(od + "/k_junc.bin", j);
    std::vector<int> kq, kt;
    std::vector<float> kd;
    for (const auto& m : km) { kq.push_back(m.queryIdx); kt.push_back(m.trainIdx); kd.push_back(m.distance); }
    dump(od + "/k_query.bin", kq.data(), kq.size());
    dump(od + "/k_train.bin", kt.data(), kt.size());
    dump(od + "/k_dist.bin", kd.data(), kd.size());
    // ... and with the temporal match of map_builder.cc:96 in the same forward: here "the last keyframe" is the left image itself
    {
      Features a2, b2, j2;
      std::vector<Eigen::Vector4d> la2, lb2;
      std::vector<cv::DMatch> km2, tm, tm_ref;
      if (!kf.Process(left, right, a2, b2, la2, lb2, j2, km2, &fl, &tm)) return 17;
      if (km2.size() != km.size()) return 18;
      pm.MatchingPoints(fl, a2, tm_ref, false);
      std::vector<int> tq, tt, rq, rt;
      std::vector<float> td, rd;
      for (const auto& m : tm) { tq.push_back(m.queryIdx); tt.push_back(m.trainIdx); td.push_back(m.distance); }
      for (const auto& m : tm_ref) { rq.push_back(m.queryIdx); rt.push_back(m.trainIdx); rd.push_back(m.distance); }
      dump(od + "/t_query.bin", tq.data(), tq.size()); dump(od + "/t_train.bin", tt.data(), tt.size()); dump(od + "/t_dist.bin", td.data(), td.size());
      dump(od + "/tr_query.bin", rq.data(), rq.size()); dump(od + "/tr_train.bin", rt.data(), rt.size()); dump(od + "/tr_dist.bin", rd.data(), rd.size());
    }
    cv::Mat none2;
    if (kf.Process(none2, right, a, b, la, lb, j, km)) return 16;
  }
  std::printf("facade gpu: use_superpoint %d matcher %d: %ld/%ld keypoints, %d matches\n", cfgs.plnet_config.use_superpoint,
              cfgs.point_matcher_config.matcher, (long)fl.cols(), (long)fr.cols(), n);
  return 0;
}
