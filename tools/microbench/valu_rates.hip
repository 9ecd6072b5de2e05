// Issue cost of the vector instructions the matcher's soft-max and epilogues are made of, on gfx950: N independent instructions of one kind per loop trip, one /
// two / three waves per SIMD, wall time over the whole chip -> cycles of a SIMD per wave-instruction (at the clock the chip holds under that load).  And the same
// instruction stream BESIDE a stream of 32x32x16 MFMAs in the same wave (R vector instructions per MFMA): does the vector time hide under the matrix time or add to it?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { K_EXP = 0, K_FMA = 1, K_PKFMA = 2, K_CVTPK = 3, K_PKADD = 4 };

// state of one independent chain: a float PAIR (the packed kinds work on it whole, the others on its halves)
template <int KIND>
__device__ __forceinline__ void op(f32x2& v, f32x2 cc) {
  if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v.x));
  if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v.x) : "v"(cc.x));
  if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(cc));
  if constexpr (KIND == K_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v.x) : "v"(v.y));
  if constexpr (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(cc));
}

// 16 independent register sets, one instruction on each per trip
template <int KIND>
__global__ void valu_only(float* sink, int iters) {
  f32x2 v[16];
  const f32x2 cc = {0.999f, 0.999f};
  for (int i = 0; i < 16; ++i) v[i] = f32x2{-0.001f * (threadIdx.x + i), 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) op<KIND>(v[i], cc);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// per trip: 4 independent 32x32x16 MFMAs (4 x 32 = 128 cycles of matrix pipe) with R vector instructions after each
template <int KIND, int R>
__global__ void beside_mfma(float* sink, int iters) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * e); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x16{};
  f32x2 v[16];
  const f32x2 cc = {0.999f, 0.999f};
  for (int i = 0; i < 16; ++i) v[i] = f32x2{-0.001f * (threadIdx.x + i), 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) op<KIND>(v[(m * R + r) & 15], cc);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  for (int m = 0; m < 4; ++m) s += acc[m][0];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();                         // warm-up (module load, clocks)
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  launch();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* sink;
  hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 20000;
  const double clk = 2.4e9;         // nominal; the printed cycles are at THIS clock (the chip may run lower under load)
  const char* names[5] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_f16_f32", "v_pk_add_f32"};
  printf("vector instructions alone: nominal-clock cycles of a SIMD per wave-instruction (256 workgroups, W waves per SIMD)\n");
  for (int wps : {1, 2, 3}) {
    double ms[5];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(valu_only<K_EXP>, dim3(256), dim3(wps * 256), 0, 0, sink, iters); });
    ms[1] = time_ms([&] { hipLaunchKernelGGL(valu_only<K_FMA>, dim3(256), dim3(wps * 256), 0, 0, sink, iters); });
    ms[2] = time_ms([&] { hipLaunchKernelGGL(valu_only<K_PKFMA>, dim3(256), dim3(wps * 256), 0, 0, sink, iters); });
    ms[3] = time_ms([&] { hipLaunchKernelGGL(valu_only<K_CVTPK>, dim3(256), dim3(wps * 256), 0, 0, sink, iters); });
    ms[4] = time_ms([&] { hipLaunchKernelGGL(valu_only<K_PKADD>, dim3(256), dim3(wps * 256), 0, 0, sink, iters); });
    for (int k = 0; k < 5; ++k) printf("  W=%d %-18s %6.2f cycles per wave-instruction\n", wps, names[k], ms[k] * 1e-3 * clk / ((double)iters * 16 * wps));
  }
  printf("\nbeside MFMAs (one wave's own stream: MFMA 32x32x16 followed by R vector instructions; W waves per SIMD): nominal-clock cycles per MFMA per SIMD (32 = the pipe's rate)\n");
  for (int wps : {1, 2, 3}) {
    auto row = [&](const char* what, double ms) { printf("  W=%d %-28s %6.1f cycles per MFMA\n", wps, what, ms * 1e-3 * clk / ((double)iters * 4 * wps)); };
    row("MFMA only", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_FMA, 0>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
    row("+ 2 v_fma_f32 per MFMA", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_FMA, 2>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
    row("+ 4 v_fma_f32 per MFMA", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_FMA, 4>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
    row("+ 8 v_fma_f32 per MFMA", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_FMA, 8>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
    row("+ 2 v_exp_f32 per MFMA", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_EXP, 2>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
    row("+ 4 v_exp_f32 per MFMA", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_EXP, 4>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
    row("+ 4 v_cvt_pk per MFMA", time_ms([&] { hipLaunchKernelGGL((beside_mfma<K_CVTPK, 4>), dim3(256), dim3(wps * 256), 0, 0, sink, iters); }));
  }
  return 0;
}
