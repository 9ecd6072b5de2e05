// ORACLE — TEST INFRASTRUCTURE ONLY.  The fake TensorRT behind ref_stubs/NvInfer.h: binding tables of the five models
// (names and shapes from the reference's own call sites), output shapes from input shapes, and the
// engine callback.
#include <cstdio>
#include <cstdlib>

#include "3rdparty/tensorrtbuffer/include/buffers.h"
#include "NvInfer.h"
#include "NvOnnxParser.h"
#include "ref_engine.h"

namespace tensorrt_log {
Logger gLogger{Logger::Severity::kWARNING};
void setReportableSeverity(Logger::Severity severity) { gLogger.setReportableSeverity(severity); }
}  // namespace tensorrt_log

static airslam_ref_engine_fn g_engine_fn = nullptr;
static void* g_engine_user = nullptr;
extern "C" void airslam_ref_set_engine(airslam_ref_engine_fn fn, void* user) {
  g_engine_fn = fn;
  g_engine_user = user;
}

namespace nvinfer1 {
namespace fake {
static Dims D(std::initializer_list<int32_t> v) {
  Dims d;
  d.nbDims = (int32_t)v.size();
  int i = 0;
  for (int32_t x : v) d.d[i++] = x;
  return d;
}
static bool has(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

Model model_for_file(const std::string& path) {
  const std::string f = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
  Model m;
  if (has(f, "plnet_s0")) {                                   // src/plnet.cpp:453-462 fetches these by name
    m.kind = "plnet_s0";
    m.bindings = {{"input", true, D({1, 1, -1, -1})},
                  {"scores", false, D({1, -1, -1})},
                  {"descriptors", false, D({1, 256, -1, -1})},
                  {"juncs_pred", false, D({300, 2})},
                  {"lines_pred", false, D({-1, 4})},
                  {"iskeep", false, D({1, 3, -1, -1})},
                  {"idx_junc_to_end_min", false, D({1, 3, -1, -1})},
                  {"idx_junc_to_end_max", false, D({1, 3, -1, -1})},
                  {"loi_features", false, D({1, 128, -1, -1})},
                  {"loi_features_thin", false, D({1, 4, -1, -1})},
                  {"loi_features_aux", false, D({1, 4, -1, -1})}};
  } else if (has(f, "plnet_s1")) {                            // src/plnet.cpp:42-49,477-486
    m.kind = "plnet_s1";
    m.bindings = {{"juncs_pred", true, D({-1, 2})},
                  {"lines_pred", true, D({-1, 4})},
                  {"idx_lines_for_junctions", true, D({-1, 2})},
                  {"inverse", true, D({-1, 1})},
                  {"iskeep_index", true, D({-1, 1})},
                  {"loi_features", true, D({1, -1, -1, -1})},
                  {"loi_features_thin", true, D({1, 4, -1, -1})},
                  {"loi_features_aux", true, D({1, 4, -1, -1})},
                  {"lines_adjusted", false, D({-1, 4})},
                  {"scores_line", false, D({-1})}};
  } else if (has(f, "lightglue")) {                           // src/point_matcher.cc:9-13, src/light_glue.cpp:129-141
    m.kind = "lightglue";
    m.bindings = {{"keypoints_0", true, D({1, -1, 2})},
                  {"keypoints_1", true, D({1, -1, 2})},
                  {"descriptors_0", true, D({1, -1, 256})},
                  {"descriptors_1", true, D({1, -1, 256})},
                  {"scores", false, D({1, -1, -1})}};
  } else if (has(f, "superglue")) {                           // src/point_matcher.cc:21-27, src/super_glue.cpp:151-174
    m.kind = "superglue";
    m.bindings = {{"keypoints_0", true, D({1, -1, 2})},
                  {"scores_0", true, D({1, -1})},
                  {"descriptors_0", true, D({1, 256, -1})},
                  {"keypoints_1", true, D({1, -1, 2})},
                  {"scores_1", true, D({1, -1})},
                  {"descriptors_1", true, D({1, 256, -1})},
                  {"scores", false, D({1, -1, -1})}};
  } else if (has(f, "superpoint")) {                          // src/feature_detector.cc:15-17, src/super_point.cpp:76-83
    m.kind = "superpoint";
    m.bindings = {{"input", true, D({1, 1, -1, -1})}, {"scores", false, D({1, -1, -1})}, {"descriptors", false, D({1, 256, -1, -1})}};
  }
  return m;
}

static int idx(const Model& m, const char* name) {
  for (size_t i = 0; i < m.bindings.size(); ++i)
    if (m.bindings[i].name == name) return (int)i;
  return -1;
}

void infer_output_dims(const Model& m, std::vector<Dims>& d) {
  auto set = [&](const char* n, Dims v) { d[(size_t)idx(m, n)] = v; };
  if (m.kind == "plnet_s0" || m.kind == "superpoint") {
    const Dims& in = d[(size_t)idx(m, "input")];
    const int H = in.d[2], W = in.d[3];
    if (H < 0 || W < 0) return;
    set("scores", D({1, H, W}));
    set("descriptors", D({1, 256, H / 8, W / 8}));
    if (m.kind == "plnet_s0") {
      const int fh = H / 4, fw = W / 4;
      set("lines_pred", D({3 * fh * fw, 4}));
      for (const char* n : {"iskeep", "idx_junc_to_end_min", "idx_junc_to_end_max"}) set(n, D({1, 3, fh, fw}));
      set("loi_features", D({1, 128, fh, fw}));
      set("loi_features_thin", D({1, 4, fh, fw}));
      set("loi_features_aux", D({1, 4, fh, fw}));
    }
  } else if (m.kind == "plnet_s1") {
    const int m2 = d[(size_t)idx(m, "idx_lines_for_junctions")].d[0];
    if (m2 < 0) return;
    set("lines_adjusted", D({m2, 4}));
    set("scores_line", D({m2}));
  } else if (m.kind == "lightglue") {
    const int n0 = d[(size_t)idx(m, "keypoints_0")].d[1], n1 = d[(size_t)idx(m, "keypoints_1")].d[1];
    if (n0 >= 0 && n1 >= 0) set("scores", D({1, n0, n1}));
  } else if (m.kind == "superglue") {
    const int n0 = d[(size_t)idx(m, "keypoints_0")].d[1], n1 = d[(size_t)idx(m, "keypoints_1")].d[1];
    if (n0 >= 0 && n1 >= 0) set("scores", D({1, n0 + 1, n1 + 1}));
  }
}

bool execute(const Model& m, const std::vector<Dims>& dims, void* const* bindings) {
  if (!g_engine_fn) {
    std::fprintf(stderr, "fake-trt: no engine callback installed (airslam_ref_set_engine)\n");
    return false;
  }
  const int nb = (int)m.bindings.size();
  std::vector<const char*> names;
  std::vector<int> is_input, ndims, flat;
  for (int i = 0; i < nb; ++i) {
    names.push_back(m.bindings[(size_t)i].name.c_str());
    is_input.push_back(m.bindings[(size_t)i].input ? 1 : 0);
    ndims.push_back(dims[(size_t)i].nbDims);
    for (int k = 0; k < 8; ++k) flat.push_back(dims[(size_t)i].d[k]);
  }
  return g_engine_fn(g_engine_user, m.kind.c_str(), nb, names.data(), is_input.data(), ndims.data(), flat.data(), bindings) == 0;
}
}  // namespace fake

ICudaEngine* IRuntime::deserializeCudaEngine(const void* blob, std::size_t size) noexcept {
  const std::string s((const char*)blob, size);
  if (s.rfind("fake-trt:", 0) != 0) return nullptr;
  auto* e = new ICudaEngine;
  e->model = fake::model_for_file(s.substr(9) + ".onnx");
  if (e->model.kind.empty()) { delete e; return nullptr; }
  return e;
}
IBuilder* createInferBuilder(ILogger&) noexcept { return new IBuilder; }
IRuntime* createInferRuntime(ILogger&) noexcept { return new IRuntime; }
}  // namespace nvinfer1

namespace nvonnxparser {
bool IParser::parseFromFile(const char* onnx_path, int) noexcept {
  network->model = nvinfer1::fake::model_for_file(onnx_path);
  if (network->model.kind.empty()) return false;
  network->in.clear();
  network->out.clear();
  for (const auto& b : network->model.bindings) {
    nvinfer1::ITensor t;
    t.dims = b.dims;
    (b.input ? network->in : network->out).push_back(t);
  }
  return true;
}
IParser* createParser(nvinfer1::INetworkDefinition& network, nvinfer1::ILogger&) noexcept {
  auto* p = new IParser;
  p->network = &network;
  return p;
}
}  // namespace nvonnxparser
