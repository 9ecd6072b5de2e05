// Cross attention needs softmax(S) V1 (lane = a query of image 0) and softmax(S^T) V0 (lane = a query of image 1) from the SAME S = Q0 K1^T.  Today each direction
// computes its own product and its own exponentials.  "Shared S": one product, one exponential, and the second direction's P tile by TRANSPOSING the
// first one inside the registers.  This probe prices both per 32 x 32 tile (d_head = 64, v_mfma_f32_32x32x16_f16), three waves per SIMD like attention32_kernel:
//   A  two products:      4 more MFMAs + 16 more v_exp_f32 (+ 8 v_cvt_pk both ways)
//   B1 one product + transpose, portable form: four register-bit <-> lane-bit exchanges through __shfl_xor + one lane permutation (ds_bpermute)
//   B2 the same with the cheapest instruction forms gfx950 has: DPP quad_perm / row_ror for lane bits 0, 1, 3, v_permlane16_swap for bit 4, ds_bpermute for the
//      lane permutation (bits 5 <-> 2; v_permlane32_swap only exchanges a register bit with lane bit 5, the permutation would take three such rounds)
// Both B forms are CHECKED against the transposed tile before they are timed.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/p_transpose.hip -o tools/microbench/p_transpose && ./tools/microbench/p_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// layout of a 32 x 32 accumulator: lane l holds column n = l % 32, rows m = 8 (r / 4) + 4 (l / 32) + r % 4: register bits (3,2,1,0) = (m4,m3,m1,m0), lane bits (5..0) = (m2,n4,n3,n2,n1,n0)
template <int RB, int LB, int FORM>
__device__ __forceinline__ void xbit(float (&v)[16], int lane) {           // exchange register-index bit RB with lane bit LB
  const bool L = (lane >> LB) & 1;
#pragma unroll
  for (int lo = 0; lo < 16; ++lo) {
    if (lo & (1 << RB)) continue;
    const int hi = lo | (1 << RB);
    if constexpr (FORM == 2 && LB == 4) {                                   // odd rows of v[lo] <-> even rows of v[hi]: one instruction, no selects
      const u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v[lo]), __builtin_bit_cast(unsigned, v[hi]), false, false);
      v[lo] = __builtin_bit_cast(float, r[0]); v[hi] = __builtin_bit_cast(float, r[1]);
    } else {
      const float send = L ? v[lo] : v[hi];
      float recv;
      if constexpr (FORM == 2 && LB == 0) recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
      else if constexpr (FORM == 2 && LB == 1) recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
      else if constexpr (FORM == 2 && LB == 3) recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xF, 0xF, true));  // row_ror:8
      else recv = __shfl_xor(send, 1 << LB);
      if (L) v[lo] = recv; else v[hi] = recv;
    }
  }
}
template <int FORM>
__device__ __forceinline__ void transpose32(float (&v)[16], int lane) {
  xbit<3, 4, FORM>(v, lane); xbit<2, 3, FORM>(v, lane); xbit<1, 1, FORM>(v, lane); xbit<0, 0, FORM>(v, lane);      // (m4,m3,m1,m0) <-> (n4,n3,n1,n0)
  const int src = (lane & ~0x24) | ((lane & 4) << 3) | ((lane >> 3) & 4);                                             // lane bits 5 <-> 2: m2 <-> n2
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = __shfl(v[r], src);
}

template <int FORM>
__global__ void check(int* bad) {
  const int lane = threadIdx.x & 63;
  float v[16];
  for (int r = 0; r < 16; ++r) v[r] = (float)((8 * (r / 4) + 4 * (lane / 32) + r % 4) * 32 + lane % 32);           // P[m][n] = 32 m + n
  transpose32<FORM>(v, lane);
  int b = 0;
  for (int r = 0; r < 16; ++r) b += v[r] != (float)((lane % 32) * 32 + 8 * (r / 4) + 4 * (lane / 32) + r % 4);     // T[m'][n'] = P[n'][m']
  atomicAdd(bad, b);
}

// which exchange of the DPP / permlane form differs from the portable one (bit i of *bad: step i of transpose32)
__global__ void check_steps(int* bad) {
  const int lane = threadIdx.x & 63;
  float a[16], b[16];
  int w = 0;
#define STEP(I, RB, LB) for (int r = 0; r < 16; ++r) a[r] = b[r] = (float)(r * 64 + lane); xbit<RB, LB, 1>(a, lane); xbit<RB, LB, 2>(b, lane); for (int r = 0; r < 16; ++r) w |= (a[r] != b[r]) << I;
  STEP(0, 3, 4) STEP(1, 2, 3) STEP(2, 1, 1) STEP(3, 0, 0)
#undef STEP
  atomicOr(bad, w);
}

template <int MODE>            // 0 = A (two products), 1 = B1, 2 = B2
__global__ __launch_bounds__(256, 3) void probe(unsigned* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 q[4], k[4];
  for (int s = 0; s < 4; ++s) for (int e = 0; e < 8; ++e) { q[s][e] = (_Float16)(0.01f * ((lane + e + s) & 15)); k[s][e] = (_Float16)(0.02f * ((lane - e + s) & 7)); }
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(q[0]), "+v"(k[0]));                             // a new tile every iteration as far as the compiler knows
    f32x16 s0;
    for (int e = 0; e < 16; ++e) s0[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[s], q[s], s0, 0, 0, 0);
    float p[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(s0[e]);
#pragma unroll
    for (int e = 0; e < 16; e += 2) acc ^= __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[e], p[e + 1]));
    if constexpr (MODE == 0) {
      f32x16 s1;
      for (int e = 0; e < 16; ++e) s1[e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[s], k[s], s1, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(s1[e]);
    } else {
      transpose32<MODE>(p, lane);
    }
#pragma unroll
    for (int e = 0; e < 16; e += 2) acc ^= __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p[e], p[e + 1]));
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static double run(unsigned* sink) {
  const int iters = 4000, wgs = 768;                                       // 768 x 4 waves = 3 waves per SIMD on 256 CUs
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, sink, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(256), 0, 0, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 3);                                   // ns per tile and SIMD (three waves take turns)
}

int main() {
  unsigned* sink; int* bad; int h[2] = {0, 0};
  hipMalloc(&sink, 768 * 256 * 4); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(check<1>, dim3(1), dim3(64), 0, 0, bad);
  hipLaunchKernelGGL(check<2>, dim3(1), dim3(64), 0, 0, bad + 1);
  hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
  printf("transpose check: portable form %s (%d wrong of 1024), DPP / permlane form %s (%d wrong)\n", h[0] ? "WRONG" : "ok", h[0], h[1] ? "WRONG" : "ok", h[1]);
  if (h[1]) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check_steps, dim3(1), dim3(64), 0, 0, bad);
    hipMemcpy(h, bad, 4, hipMemcpyDeviceToHost);
    printf("  exchanges of the DPP / permlane form that differ from the portable one: %s%s%s%s (its time below is that of the instruction mix, not of a correct transpose)\n",
           (h[0] & 1) ? "[lane bit 4: v_permlane16_swap] " : "", (h[0] & 2) ? "[lane bit 3: row_ror:8] " : "", (h[0] & 4) ? "[lane bit 1: quad_perm 2301] " : "", (h[0] & 8) ? "[lane bit 0: quad_perm 1032] " : "");
  }
  const double a = run<0>(sink), b1 = run<1>(sink), b2 = run<2>(sink);
  printf("per 32x32 tile and SIMD (3 waves per SIMD, both directions' P packed to fp16):\n");
  printf("  A  two products (8 MFMA + 32 v_exp_f32 + 16 v_cvt_pk)                 %7.1f ns\n", a);
  printf("  B1 one product + register transpose, __shfl_xor form                  %7.1f ns   (%.2fx A)\n", b1, b1 / a);
  printf("  B2 one product + register transpose, DPP / v_permlane16_swap form     %7.1f ns   (%.2fx A)\n", b2, b2 / a);
  printf("the second product + its exponentials cost %.1f ns per tile (A minus the shared half = A / 2); the transpose costs %.1f (B1) / %.1f (B2) ns on top of that half\n", a / 2, b1 - a / 2, b2 - a / 2);
  return 0;
}
