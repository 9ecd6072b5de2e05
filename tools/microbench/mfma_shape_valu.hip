// VERDICT r03 #9 / "next" #4: would the encoder's kernels gain from v_mfma_f32_32x32x16 instead of v_mfma_f32_16x16x32?
// Both shapes do 512 MAC / cycle / SIMD; what differs is the instruction count (one 32x32x16 = two 16x16x32) and the operand reads per MAC.
// The conv kernels are bound by ISSUE (every non-MFMA instruction costs the matrix pipe ~4 cycles), so the question this
// probe answers is: at a FIXED amount of vector work per FLOP (V independent v_fma_f32 per 32768 FLOP — what a kernel's address / epilogue /
// conv1a work amounts to), does the big shape finish sooner?  Two waves per SIMD (8 per CU), like conv64r / conv128r.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_shape_valu.hip -o /tmp/mfma_shape_valu && /tmp/mfma_shape_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BIG, int V>
__global__ __launch_bounds__(512, 1) void probe(float* sink, int iters) {
  const int tid = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((tid + e) & 7); b[e] = (_Float16)(float)((tid - e) & 3); }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = (float)(tid + e);
  const float k0 = 1.0001f, k1 = 0.5f;
  float s = 0.f;
  if constexpr (BIG) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < V; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(k0), "v"(k1));
      }
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  } else {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[2 * i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < V / 2; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(k0), "v"(k1));
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[2 * i + 1], 0, 0, 0);
#pragma unroll
        for (int q = V / 2; q < V; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(k0), "v"(k1));
      }
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0];
  }
  for (int e = 0; e < 8; ++e) s += v[e];
  sink[blockIdx.x * blockDim.x + tid] = s;
}

template <int BIG, int V>
static double run(float* sink) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<BIG, V>), dim3(256), dim3(512), 0, 0, sink, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<BIG, V>), dim3(256), dim3(512), 0, 0, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 8 * iters * 4 * 32768.0;          // workgroups x waves x iterations x units x FLOP per unit
  return flop / (ms * 1e-3) / 1e12;
}

template <int V>
static void row(float* sink) {
  const double small = run<0, V>(sink), big = run<1, V>(sink);
  std::printf("V = %2d v_fma per 32768 FLOP (VALU : MFMA = %.1f for 16x16x32, %.1f for 32x32x16):  16x16x32 %7.1f TFLOP/s   32x32x16 %7.1f TFLOP/s   ratio %.3f\n",
              V, V / 2.0, (double)V, small, big, big / small);
}

int main() {
  float* sink;
  hipMalloc(&sink, 256 * 512 * sizeof(float));
  row<0>(sink); row<2>(sink); row<4>(sink); row<8>(sink); row<12>(sink); row<16>(sink); row<24>(sink);
  return 0;
}
