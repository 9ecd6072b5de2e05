// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for the reference's 3rdparty/tensorrtbuffer/include/buffers.h (+ the
// pieces of its common.h / logger.h the front-end sources use), on the fake TensorRT of NvInfer.h — the vendored
// original needs the CUDA runtime and the whole TensorRT sample framework.
// Same surface and the same behaviour as the original where the reference relies on it (buffers.h:237-417): one host and
// one "device" buffer per binding, sized from the CONTEXT's binding dimensions at construction; getHostBuffer(name) =
// nullptr for unknown names; copyInputToDevice / copyOutputToHost copy input / output bindings only.
#ifndef TENSORRT_BUFFERS_H
#define TENSORRT_BUFFERS_H
#include <cassert>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "NvInfer.h"

namespace tensorrt_log {
class Logger : public nvinfer1::ILogger {
 public:
  explicit Logger(Severity severity = Severity::kWARNING) : sev_(severity) {}
  nvinfer1::ILogger& getTRTLogger() noexcept { return *this; }
  void log(Severity, const char*) noexcept override {}
  void setReportableSeverity(Severity s) noexcept { sev_ = s; }
  Severity getReportableSeverity() const { return sev_; }

 private:
  Severity sev_;
};
extern Logger gLogger;
void setReportableSeverity(Logger::Severity severity);
static std::ostream& gLogError = std::cerr;
}  // namespace tensorrt_log

#undef ASSERT
#define ASSERT(condition)                                                         \
  do {                                                                            \
    if (!(condition)) {                                                           \
      std::cerr << "Assertion failure: " << #condition << std::endl;             \
      abort();                                                                    \
    }                                                                             \
  } while (0)

namespace tensorrt_buffer {
struct InferDeleter {
  template <typename T>
  void operator()(T* obj) const { delete obj; }
};
template <typename T>
using TensorRTUniquePtr = std::unique_ptr<T, InferDeleter>;

inline std::unique_ptr<cudaStream_t> makeCudaStream() { return std::unique_ptr<cudaStream_t>(new cudaStream_t(nullptr)); }
inline void enableDLA(nvinfer1::IBuilder*, nvinfer1::IBuilderConfig*, int useDLACore, bool = true) { assert(useDLACore < 0); (void)useDLACore; }
inline int64_t volume(const nvinfer1::Dims& d) {
  return std::accumulate(d.d, d.d + d.nbDims, int64_t{1}, std::multiplies<int64_t>{});
}

class BufferManager {
 public:
  static const size_t kINVALID_SIZE_VALUE = ~size_t(0);
  BufferManager(std::shared_ptr<nvinfer1::ICudaEngine> engine, const int batchSize = 0, const nvinfer1::IExecutionContext* context = nullptr)
      : mEngine(engine) {
    (void)batchSize;
    for (int i = 0; i < mEngine->getNbBindings(); i++) {
      auto dims = context ? context->getBindingDimensions(i) : mEngine->getBindingDimensions(i);
      const int64_t vol = volume(dims);
      ASSERT(vol >= 0);
      mHost.emplace_back((size_t)vol, 0.f);
      mDevice.emplace_back((size_t)vol, 0.f);
      mDeviceBindings.emplace_back(mDevice.back().data());
    }
  }
  std::vector<void*>& getDeviceBindings() { return mDeviceBindings; }
  const std::vector<void*>& getDeviceBindings() const { return mDeviceBindings; }
  void* getDeviceBuffer(const std::string& tensorName) const { return getBuffer(false, tensorName); }
  void* getHostBuffer(const std::string& tensorName) const { return getBuffer(true, tensorName); }
  size_t size(const std::string& tensorName) const {
    int index = mEngine->getBindingIndex(tensorName.c_str());
    return index == -1 ? kINVALID_SIZE_VALUE : mHost[(size_t)index].size() * sizeof(float);
  }
  void copyInputToDevice() { memcpyBuffers(true, false); }
  void copyOutputToHost() { memcpyBuffers(false, true); }

 private:
  void* getBuffer(const bool isHost, const std::string& tensorName) const {
    int index = mEngine->getBindingIndex(tensorName.c_str());
    if (index == -1) return nullptr;
    return (void*)(isHost ? mHost[(size_t)index].data() : mDevice[(size_t)index].data());
  }
  void memcpyBuffers(const bool copyInput, const bool deviceToHost) {
    for (int i = 0; i < mEngine->getNbBindings(); i++)
      if ((copyInput && mEngine->bindingIsInput(i)) || (!copyInput && !mEngine->bindingIsInput(i))) {
        if (deviceToHost) mHost[(size_t)i] = mDevice[(size_t)i];
        else mDevice[(size_t)i] = mHost[(size_t)i];
      }
  }
  std::shared_ptr<nvinfer1::ICudaEngine> mEngine;
  std::vector<std::vector<float>> mHost, mDevice;
  std::vector<void*> mDeviceBindings;
};
}  // namespace tensorrt_buffer
#endif  // TENSORRT_BUFFERS_H
