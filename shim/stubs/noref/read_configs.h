// MINIMAL compile-smoke stand-in for the reference's include/read_configs.h (needs yaml-cpp, absent here):
// the three POD config structs with the fields the wrappers read (include/read_configs.h:9-103).
#pragma once
#include <string>
#include <vector>
struct PLNetConfig {
  std::string superpoint_onnx, superpoint_engine, plnet_s0_onnx, plnet_s0_engine, plnet_s1_onnx, plnet_s1_engine;
  int use_superpoint = 0, max_keypoints = 400;
  float keypoint_threshold = 0.004f;
  int remove_borders = 4;
  float line_threshold = 0.75f, line_length_threshold = 50.f;
};
struct SuperPointConfig {
  int max_keypoints = 400;
  float keypoint_threshold = 0.004f;
  int remove_borders = 4, dla_core = -1;
  std::vector<std::string> input_tensor_names, output_tensor_names;
  std::string onnx_file, engine_file;
};
struct PointMatcherConfig {
  int matcher = 0, image_width = 752, image_height = 480, dla_core = -1;
  std::vector<std::string> input_tensor_names, output_tensor_names;
  std::string onnx_file, engine_file;
};
