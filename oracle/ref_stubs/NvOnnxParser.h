// ORACLE — TEST INFRASTRUCTURE ONLY.  fake-TensorRT ONNX parser (see NvInfer.h): parseFromFile() identifies the model by
// its file name and fills the network's binding table; it does not read the file (5 of the 6 ONNX files are absent upstream).
#pragma once
#include "NvInfer.h"
namespace nvonnxparser {
class IParser {
 public:
  nvinfer1::INetworkDefinition* network = nullptr;
  bool parseFromFile(const char* onnx_path, int verbosity) noexcept;    // fake_trt.cpp
};
IParser* createParser(nvinfer1::INetworkDefinition& network, nvinfer1::ILogger& logger) noexcept;
}  // namespace nvonnxparser
