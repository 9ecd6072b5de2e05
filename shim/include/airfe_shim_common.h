// Shared helpers of the TensorRT-free replacement wrappers (shim/src/*.cpp).
#ifndef AIRFE_SHIM_COMMON_H_
#define AIRFE_SHIM_COMMON_H_
#include <string>

#include "airfe.h"

namespace airfe_shim {
// "<dir>/plnet_s1.onnx" -> "<dir>/plnet_s1.airfe": packs sit next to where the reference keeps its ONNX/engine
// files (PLNetConfig::SetModelPath, include/read_configs.h:39-49); tools/onnx_to_pack.py writes them.
inline std::string pack_path(const std::string& onnx_path) {
  const std::string::size_type dot = onnx_path.rfind('.');
  return (dot == std::string::npos ? onnx_path : onnx_path.substr(0, dot)) + ".airfe";
}
}  // namespace airfe_shim
#endif
