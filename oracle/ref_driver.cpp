// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points into the REFERENCE's own front-end code, compiled unchanged from
// /root/reference by oracle/Makefile into oracle/_ref/libairslam_ref.so:
//   src/feature_detector.cc, src/point_matcher.cc, src/plnet.cpp, src/super_point.cpp, src/light_glue.cpp, src/super_glue.cpp
//   and src/line_processor.cc:1-180 (PointLineDistance .. MatchLines; the rest of that file needs g2o / Camera),
//   behind the reference's own headers include/{feature_detector,point_matcher,plnet,super_point,light_glue,super_glue,read_configs}.h.
// Stand-ins (ours): Eigen / OpenCV / yaml-cpp / utils.h (shim/stubs/), TensorRT + BufferManager (oracle/ref_stubs/).
// The engines' outputs come from the callback installed with airslam_ref_set_engine (ref_stubs/ref_engine.h).
// Nothing of this file or of oracle/_ref is linked into, imported by or shipped with the product.
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "feature_detector.h"
#include "line_processor.h"
#include "point_matcher.h"
#include "ref_engine.h"

// free functions of the reference that its headers do not declare
void filter_matches(const Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic>& scores, Eigen::Matrix<int, Eigen::Dynamic, 2>& matches_index,
                    Eigen::Matrix<float, Eigen::Dynamic, 1>& matches_score, float threshold);               // src/light_glue.cpp:214
void decode(float* scores, int h, int w, std::vector<int>& indices0, std::vector<int>& indices1, std::vector<float>& mscores0,
            std::vector<float>& mscores1);                                                                   // src/super_glue.cpp:339
void log_optimal_transport(float* scores, float* Z, int m, int n, float alpha, int iters);                  // src/super_glue.cpp:400

namespace {
typedef Eigen::Matrix<float, 259, Eigen::Dynamic> Features;

void put_features(const Features& f, float* out, int cap, int* n) {
  *n = (int)f.cols();
  if (out && f.cols() <= cap && f.cols() > 0) std::memcpy(out, f.data(), (size_t)f.cols() * 259 * sizeof(float));   // column-major 259 x N
}
void get_features(const float* in, int n, Features& f) {
  f.resize(259, n);
  if (n > 0) std::memcpy(f.data(), in, (size_t)n * 259 * sizeof(float));
}
void put_lines(const std::vector<Eigen::Vector4d>& l, double* out, int cap, int* n) {
  *n = (int)l.size();
  if (out && (int)l.size() <= cap)
    for (size_t i = 0; i < l.size(); ++i)
      for (int k = 0; k < 4; ++k) out[4 * i + k] = l[i][k];
}
cv::Mat wrap(const uint8_t* p, int h, int w, int stride) { return p ? cv::Mat(h, w, CV_8UC1, (void*)p, (size_t)stride) : cv::Mat(); }
}  // namespace

extern "C" {

// ---------------------------------------------------------------- FeatureDetector (src/feature_detector.cc)
// model_dir: where SetModelPath (include/read_configs.h:39-49) puts the ONNX / engine names; the reference's save_engine() WRITES
// <model_dir>/*.engine there — pass a scratch directory.
void* airslam_ref_detector_create(const char* model_dir, int use_superpoint, int max_keypoints, float keypoint_threshold, int remove_borders,
                                  float line_threshold, float line_length_threshold) {
  PLNetConfig cfg;
  cfg.use_superpoint = use_superpoint;
  cfg.max_keypoints = max_keypoints;
  cfg.keypoint_threshold = keypoint_threshold;
  cfg.remove_borders = remove_borders;
  cfg.line_threshold = line_threshold;
  cfg.line_length_threshold = line_length_threshold;
  cfg.SetModelPath(model_dir);
  return new FeatureDetector(cfg);
}
// the reference's own config loader on one of its own YAML files (configs/visual_odometry/*.yaml): returns the parsed PLNet block too
void* airslam_ref_detector_create_from_yaml(const char* yaml_path, const char* model_dir, int* use_superpoint, int* max_keypoints,
                                            float* keypoint_threshold, int* remove_borders, float* line_threshold, float* line_length_threshold) {
  VisualOdometryConfigs cfgs(yaml_path, model_dir);
  const PLNetConfig& c = cfgs.plnet_config;
  *use_superpoint = c.use_superpoint; *max_keypoints = c.max_keypoints; *keypoint_threshold = c.keypoint_threshold;
  *remove_borders = c.remove_borders; *line_threshold = c.line_threshold; *line_length_threshold = c.line_length_threshold;
  return new FeatureDetector(c);
}
void airslam_ref_detector_destroy(void* d) { delete (FeatureDetector*)d; }

// The six Detect overloads (src/feature_detector.cc:36,52,62,71,83,97), `overload` = 0..5 in source order.  `n_lines_in` lines already in
// `lines_l` stay in front (the reference appends, src/plnet.cpp:544).  Returns the bool of Detect.
int airslam_ref_detect(void* d, int overload, const uint8_t* left, const uint8_t* right, int h, int w, int stride,
                       float* feat_l, int* n_l, float* feat_r, int* n_r, int cap,
                       double* lines_l, int n_lines_in, int* nl_l, double* lines_r, int* nl_r, int cap_lines,
                       float* junc, int* n_j, int cap_j) {
  FeatureDetector* det = (FeatureDetector*)d;
  cv::Mat il = wrap(left, h, w, stride), ir = wrap(right, h, w, stride);
  Features fl, fr, jn;
  std::vector<Eigen::Vector4d> ll, lr;
  for (int i = 0; i < n_lines_in; ++i) ll.emplace_back(lines_l[4 * i], lines_l[4 * i + 1], lines_l[4 * i + 2], lines_l[4 * i + 3]);
  bool ok = false;
  switch (overload) {
    case 0: ok = det->Detect(il, fl); break;
    case 1: ok = det->Detect(il, fl, ll); break;
    case 2: ok = det->Detect(il, fl, ll, jn); break;
    case 3: ok = det->Detect(il, ir, fl, fr); break;
    case 4: ok = det->Detect(il, ir, fl, fr, ll, lr); break;
    case 5: ok = det->Detect(il, ir, fl, fr, ll, lr, jn); break;
    default: return -1;
  }
  put_features(fl, feat_l, cap, n_l);
  put_features(fr, feat_r, cap, n_r);
  put_features(jn, junc, cap_j, n_j);
  put_lines(ll, lines_l, cap_lines, nl_l);
  put_lines(lr, lines_r, cap_lines, nl_r);
  return ok ? 1 : 0;
}

// ---------------------------------------------------------------- PointMatcher (src/point_matcher.cc)
void* airslam_ref_matcher_create(const char* model_dir, int matcher, int image_width, int image_height) {
  PointMatcherConfig cfg;
  cfg.matcher = matcher;
  cfg.image_width = image_width;
  cfg.image_height = image_height;
  cfg.onnx_file = ConcatenateFolderAndFileName(model_dir, matcher ? "superglue_outdoor_sim_int32.onnx" : "superpoint_lightglue.onnx");
  cfg.engine_file = ConcatenateFolderAndFileName(model_dir, matcher ? "superglue_outdoor_sim_int32.engine" : "superpoint_lightglue.engine");
  return new PointMatcher(cfg);
}
void airslam_ref_matcher_destroy(void* m) { delete (PointMatcher*)m; }

void airslam_ref_normalize_keypoints(void* m, const float* feat, int n, int width, int height, float scale, float* out) {
  Features f, o;
  get_features(feat, n, f);
  ((PointMatcher*)m)->NormalizeKeypoints(f, o, width, height, scale);
  int k;
  put_features(o, out, n, &k);
}
// MatchingPoints (src/point_matcher.cc:50-107): returns its return value; matches as (queryIdx, trainIdx, distance)
int airslam_ref_matching_points(void* m, const float* f0, int n0, const float* f1, int n1, int* query, int* train, float* distance, int cap,
                                int outlier_rejection) {
  Features a, b;
  get_features(f0, n0, a);
  get_features(f1, n1, b);
  std::vector<cv::DMatch> matches;
  const int r = ((PointMatcher*)m)->MatchingPoints(a, b, matches, outlier_rejection != 0);
  for (size_t i = 0; i < matches.size() && (int)i < cap; ++i) {
    query[i] = matches[i].queryIdx; train[i] = matches[i].trainIdx; distance[i] = matches[i].distance;
  }
  return r;
}

// ---------------------------------------------------------------- free functions of the matchers
// filter_matches (src/light_glue.cpp:214-266) on a row-major [n0][n1] score matrix, copied into the Eigen matrix exactly as
// process_output does (src/light_glue.cpp:268-281)
int airslam_ref_filter_matches(const float* scores, int n0, int n1, float threshold, int* idx /*[min(n0,n1)][2]*/, float* score) {
  Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> s;
  s.resize(n0, n1);
  for (int r = 0; r < n0; ++r)
    for (int c = 0; c < n1; ++c) s(r, c) = scores[(size_t)r * n1 + c];
  Eigen::Matrix<int, Eigen::Dynamic, 2> mi;
  Eigen::Matrix<float, Eigen::Dynamic, 1> ms;
  filter_matches(s, mi, ms, threshold);
  for (int i = 0; i < mi.rows(); ++i) { idx[2 * i] = mi(i, 0); idx[2 * i + 1] = mi(i, 1); score[i] = ms(i); }
  return (int)mi.rows();
}
// decode (src/super_glue.cpp:339-367) on the [h][w] score matrix INCLUDING the dustbin row / column; outputs of length h-1 / w-1
void airslam_ref_sg_decode(const float* scores, int h, int w, int* indices0, int* indices1, float* mscores0, float* mscores1) {
  std::vector<float> s(scores, scores + (size_t)h * w), m0, m1;
  std::vector<int> i0, i1;
  decode(s.data(), h, w, i0, i1, m0, m1);
  std::memcpy(indices0, i0.data(), i0.size() * sizeof(int));
  std::memcpy(indices1, i1.data(), i1.size() * sizeof(int));
  std::memcpy(mscores0, m0.data(), m0.size() * sizeof(float));
  std::memcpy(mscores1, m1.data(), m1.size() * sizeof(float));
}
// log_optimal_transport (src/super_glue.cpp:400-435; dead code in the reference, the Sinkhorn cross-check of SURVEY.md §8c)
void airslam_ref_log_optimal_transport(const float* scores, int m, int n, float alpha, int iters, float* Z /*[(m+1)(n+1)]*/) {
  std::vector<float> s(scores, scores + (size_t)m * n);
  log_optimal_transport(s.data(), Z, m, n, alpha, iters);
}

// ---------------------------------------------------------------- src/line_processor.cc:68-180
// AssignPointsToLines: relation as CSR (offsets [nl + 1], point index / distance in std::map order = ascending point index)
int airslam_ref_assign_points_to_lines(const double* lines, int nl, const float* feat, int n, int* offsets, int* pidx, double* pdist, int cap) {
  std::vector<Eigen::Vector4d> l;
  for (int i = 0; i < nl; ++i) l.emplace_back(lines[4 * i], lines[4 * i + 1], lines[4 * i + 2], lines[4 * i + 3]);
  Features f;
  get_features(feat, n, f);
  std::vector<std::map<int, double>> rel;
  if (nl > 0) AssignPointsToLines(l, f, rel);          // the reference dereferences lines[0]: its callers never pass an empty list
  int k = 0;
  offsets[0] = 0;
  for (int i = 0; i < nl; ++i) {
    for (const auto& kv : rel[(size_t)i]) {
      if (k < cap) { pidx[k] = kv.first; pdist[k] = kv.second; }
      ++k;
    }
    offsets[i + 1] = k;
  }
  return k;
}
void airslam_ref_match_lines(const int* off0, const int* pidx0, int nl0, const int* off1, const int* pidx1, int nl1, const int* query,
                             const int* train, int nmatch, int point_num0, int point_num1, int* line_matches /*[nl0]*/) {
  auto rel = [](const int* off, const int* pidx, int nl) {
    std::vector<std::map<int, double>> r((size_t)nl);
    for (int i = 0; i < nl; ++i)
      for (int k = off[i]; k < off[i + 1]; ++k) r[(size_t)i][pidx[k]] = 0.0;
    return r;
  };
  std::vector<cv::DMatch> pm;
  for (int i = 0; i < nmatch; ++i) pm.emplace_back(query[i], train[i], 0.f);
  std::vector<int> lm;
  MatchLines(rel(off0, pidx0, nl0), rel(off1, pidx1, nl1), pm, (size_t)point_num0, (size_t)point_num1, lm);
  for (size_t i = 0; i < lm.size(); ++i) line_matches[i] = lm[i];
}
float airslam_ref_point_line_distance(const float* line4, const float* point2) {
  return PointLineDistance(Eigen::Vector4f(line4[0], line4[1], line4[2], line4[3]), Eigen::Vector2f(point2[0], point2[1]));
}

const char* airslam_ref_sources() {
  return "src/feature_detector.cc src/point_matcher.cc src/plnet.cpp src/super_point.cpp src/light_glue.cpp src/super_glue.cpp "
         "src/line_processor.cc:1-180 3rdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h 3rdparty/DBoW2/src/{BowVector,FeatureVector,ScoringObject}.cpp "
         "src/bow/FSuperpoint.cc";
}
}  // extern "C"
