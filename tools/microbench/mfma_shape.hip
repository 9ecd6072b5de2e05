// Does the MFMA shape change what ONE wave per SIMD can sustain?  16x16x32 (16 KFLOP... 16384 FLOP) vs 32x32x16 (32768 FLOP) bf16,
// NACC independent accumulators per wave, back to back, 4 / 8 / 16 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BIG, int NACC>
__global__ void probe(float* sink, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(tid + e); b[e] = (__bf16)(float)(tid - e); }
  float s = 0.f;
  if constexpr (BIG) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
  }
  sink[blockIdx.x * blockDim.x + tid] = s;
}

template <int BIG, int NACC>
static void run(float* sink, int waves) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<BIG, NACC>), dim3(256), dim3(waves * 64), 0, 0, sink, 100);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<BIG, NACC>), dim3(256), dim3(waves * 64), 0, 0, sink, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)iters * NACC * waves * 256 * (BIG ? 32768.0 : 16384.0);
  printf("%s NACC=%2d waves/CU=%2d: %.3f ms, %.0f TFLOP/s chip\n", BIG ? "32x32x16" : "16x16x32", NACC, waves, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
  float* sink; hipMalloc(&sink, 256 * 1024 * 4);
  for (int waves : {4, 8, 16}) { run<0, 16>(sink, waves); run<1, 4>(sink, waves); run<1, 8>(sink, waves); }
  return 0;
}
