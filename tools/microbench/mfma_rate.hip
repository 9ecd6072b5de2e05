// SUPERSEDED by mfma_shape.hip: this probe times the FIRST launch of each configuration (module load included), which made one
// wave per SIMD look like 1.22 PFLOP/s; warmed up it is 1.94.
// MFMA issue-rate probe for gfx950: NACC independent v_mfma_f32_16x16x32_bf16 accumulators per wave, back to back.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(tid + e); b[e] = (__bf16)(float)(tid - e); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = s;
}

int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 8192); hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 4000;
  for (int waves : {4, 8, 16}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<16>, dim3(256), dim3(waves * 64), 0, 0, d, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto c : h) cyc += (double)c; cyc /= 256;
    const double n = (double)iters * 16;
    printf("waves/CU=%2d: %.1f ticks per MFMA per wave, kernel %.3f ms, %.0f ticks/us, %.1f TFLOP/s chip\n", waves, cyc / n, ms, cyc / (ms * 1e3),
           n * waves * 256 * 16384.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
