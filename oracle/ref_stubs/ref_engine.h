// ORACLE — TEST INFRASTRUCTURE ONLY.  The callback that stands where TensorRT executes an engine (see NvInfer.h).
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
// model: "plnet_s0" | "plnet_s1" | "superpoint" | "lightglue" | "superglue"; dims: nb rows of 8 ints; buffers: fp32, inputs filled by
// the reference's code, outputs to be filled by the callee.  Returns 0 on success.
typedef int (*airslam_ref_engine_fn)(void* user, const char* model, int nb, const char* const* names, const int* is_input, const int* ndims,
                                     const int* dims, void* const* buffers);
void airslam_ref_set_engine(airslam_ref_engine_fn fn, void* user);
#ifdef __cplusplus
}
#endif
