// ORACLE — TEST INFRASTRUCTURE ONLY.  fake-TensorRT: a stand-in for <NvInfer.h> so that the reference's own
// src/{plnet,super_point,light_glue,super_glue}.cpp compile UNCHANGED with g++ (oracle/Makefile -> oracle/_ref/).
//
// TensorRT (NVIDIA-only) executes the ONNX graphs in the reference; here an "engine" knows only its
// binding table (names / shapes from SURVEY.md Appendix A, i.e. from the reference's own call sites) and
// IExecutionContext::executeV2 hands the bound buffers to a callback the test harness installs
// (airslam_ref_set_engine, oracle/ref_driver.cpp) — the harness fills the output bindings with the tensors of
// whatever network body it wants the reference's host code to post-process (the oracle's PyTorch bodies, the
// real plnet_s1.onnx through the ONNX interpreter, or maps read back from the HIP library).  Everything AROUND the engine
// call — process_image / process_input, BufferManager traffic, wireframe_matcher, the stage-1 feed, the line filter,
// detect_point, extract_descriptors, junction_detector, filter_matches, decode — is the reference's code, compiled as is.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

typedef void* cudaStream_t;

namespace nvinfer1 {
struct Dims {
  static constexpr int32_t MAX_DIMS = 8;
  int32_t nbDims = 0;
  int32_t d[MAX_DIMS] = {0, 0, 0, 0, 0, 0, 0, 0};
};
struct Dims2 : Dims { Dims2(int32_t a, int32_t b) { nbDims = 2; d[0] = a; d[1] = b; } };
struct Dims3 : Dims { Dims3(int32_t a, int32_t b, int32_t c) { nbDims = 3; d[0] = a; d[1] = b; d[2] = c; } };
struct Dims4 : Dims { Dims4(int32_t a, int32_t b, int32_t c, int32_t e) { nbDims = 4; d[0] = a; d[1] = b; d[2] = c; d[3] = e; } };
enum class DataType : int32_t { kFLOAT = 0, kHALF = 1, kINT8 = 2, kINT32 = 3, kBOOL = 4 };
enum class BuilderFlag : int32_t { kFP16 = 0, kINT8 = 1, kDEBUG = 2, kGPU_FALLBACK = 3, kSTRICT_TYPES = 4, kREFIT = 5, kTF32 = 7 };
enum class NetworkDefinitionCreationFlag : int32_t { kEXPLICIT_BATCH = 0 };
enum class OptProfileSelector : int32_t { kMIN = 0, kOPT = 1, kMAX = 2 };
enum class DeviceType : int32_t { kGPU = 0, kDLA = 1 };

class ILogger {
 public:
  enum class Severity : int32_t { kINTERNAL_ERROR = 0, kERROR = 1, kWARNING = 2, kINFO = 3, kVERBOSE = 4 };
  virtual void log(Severity severity, const char* msg) noexcept = 0;
  virtual ~ILogger() = default;
};

namespace fake {
struct Binding {
  std::string name;
  bool input;
  Dims dims;                      // -1 = dynamic
};
struct Model {
  std::string kind;               // "plnet_s0" | "plnet_s1" | "superpoint" | "lightglue" | "superglue"
  std::vector<Binding> bindings;  // inputs first, in the order the reference's assertions expect
};
Model model_for_file(const std::string& onnx_path);                     // fake_trt.cpp: by file name
void infer_output_dims(const Model& m, std::vector<Dims>& dims);      // fake_trt.cpp: output shapes from the input shapes
bool execute(const Model& m, const std::vector<Dims>& dims, void* const* bindings);   // fake_trt.cpp: the harness callback
}  // namespace fake

class IHostMemory {
 public:
  std::string blob;
  void* data() const noexcept { return (void*)blob.data(); }
  std::size_t size() const noexcept { return blob.size(); }
};
class ITensor {
 public:
  Dims dims;
  Dims getDimensions() const noexcept { return dims; }
};
class INetworkDefinition {
 public:
  fake::Model model;
  std::vector<ITensor> in, out;
  int32_t getNbInputs() const noexcept { return (int32_t)in.size(); }
  int32_t getNbOutputs() const noexcept { return (int32_t)out.size(); }
  ITensor* getInput(int32_t i) noexcept { return &in[(size_t)i]; }
  ITensor* getOutput(int32_t i) noexcept { return &out[(size_t)i]; }
};
class IOptimizationProfile {
 public:
  bool setDimensions(const char*, OptProfileSelector, Dims) noexcept { return true; }
};
class IBuilderConfig {
 public:
  IOptimizationProfile profile;
  void setFlag(BuilderFlag) noexcept {}
  int32_t addOptimizationProfile(const IOptimizationProfile*) noexcept { return 0; }
  void setProfileStream(const cudaStream_t) noexcept {}
  void setDefaultDeviceType(DeviceType) noexcept {}
  void setDLACore(int32_t) noexcept {}
};
class ICudaEngine;
class IExecutionContext {
 public:
  const ICudaEngine* engine = nullptr;
  std::vector<Dims> dims;
  bool setBindingDimensions(int32_t i, Dims d) noexcept;
  Dims getBindingDimensions(int32_t i) const noexcept { return dims[(size_t)i]; }
  bool executeV2(void* const* bindings) noexcept;
};
class ICudaEngine {
 public:
  fake::Model model;
  int32_t getNbBindings() const noexcept { return (int32_t)model.bindings.size(); }
  int32_t getBindingIndex(const char* name) const noexcept {
    for (size_t i = 0; i < model.bindings.size(); ++i)
      if (model.bindings[i].name == name) return (int32_t)i;
    return -1;
  }
  const char* getBindingName(int32_t i) const noexcept { return model.bindings[(size_t)i].name.c_str(); }
  bool bindingIsInput(int32_t i) const noexcept { return model.bindings[(size_t)i].input; }
  Dims getBindingDimensions(int32_t i) const noexcept { return model.bindings[(size_t)i].dims; }
  DataType getBindingDataType(int32_t) const noexcept { return DataType::kFLOAT; }
  int32_t getBindingVectorizedDim(int32_t) const noexcept { return -1; }
  int32_t getBindingComponentsPerElement(int32_t) const noexcept { return 1; }
  bool hasImplicitBatchDimension() const noexcept { return false; }
  IExecutionContext* createExecutionContext() noexcept {
    auto* c = new IExecutionContext;
    c->engine = this;
    for (const auto& b : model.bindings) c->dims.push_back(b.dims);
    return c;
  }
  IHostMemory* serialize() const noexcept { auto* m = new IHostMemory; m->blob = "fake-trt:" + model.kind; return m; }
};
inline bool IExecutionContext::setBindingDimensions(int32_t i, Dims d) noexcept {
  dims[(size_t)i] = d;
  fake::infer_output_dims(engine->model, dims);
  return true;
}
inline bool IExecutionContext::executeV2(void* const* bindings) noexcept { return fake::execute(engine->model, dims, bindings); }

class IBuilder {
 public:
  INetworkDefinition* createNetworkV2(uint32_t) noexcept { return new INetworkDefinition; }
  IBuilderConfig* createBuilderConfig() noexcept { return new IBuilderConfig; }
  IOptimizationProfile* createOptimizationProfile() noexcept { return &profile_; }
  IHostMemory* buildSerializedNetwork(INetworkDefinition& n, IBuilderConfig&) noexcept {
    if (n.model.kind.empty()) return nullptr;
    auto* m = new IHostMemory;
    m->blob = "fake-trt:" + n.model.kind;
    return m;
  }
  int32_t getNbDLACores() const noexcept { return 0; }
  bool platformHasFastFp16() const noexcept { return true; }

 private:
  IOptimizationProfile profile_;
};
class IRuntime {
 public:
  ICudaEngine* deserializeCudaEngine(const void* blob, std::size_t size) noexcept;       // fake_trt.cpp
};
IBuilder* createInferBuilder(ILogger& logger) noexcept;
IRuntime* createInferRuntime(ILogger& logger) noexcept;
}  // namespace nvinfer1
