// VERDICT r03 #12 / "next" #8: reproduce or retract the "erratum" of DESIGN.md §2.2.
// Round 3 traced an irreproducible LightGlue score (0.13 % of 64-pair steps, 4 % with another stream's kernels beside the matcher) to ONE
// instruction form in the rotary epilogue of gemmr_pair_kernel<ROT>:
//     v_pk_fma_f32 v[d:d+1], v[x0:x1], v[c0:c1], v[p0:p1] op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]      (d.lo = x0 * c1 - p0)
// whose LOW product came out as 0 in lanes 48-63.  This probe runs exactly that instruction 4 x 10^7 times per lane group on operands
//   A  held in plain registers,
//   B  x produced by an MFMA immediately before (as in the kernel: x = a 2-byte GEMM accumulator), tables in registers,
//   C  as B, tables read from LDS right before the instruction (ds_read_b64 + the kernel's hand-placed `s_waitcnt lgkmcnt(0)`),
//   D  as C, the LDS table itself filled by LDS-DMA (global_load_lds_dwordx4 + `s_waitcnt vmcnt(0)` + barrier) once per outer iteration,
//   E  as B with 32 wait states between the v_accvgpr_read of the MFMA result and the packed instruction,   F  as B without the cross-half select (op_sel:[0,0,0]),
//   G  with an MFMA in flight whose result is NOT an operand of the instruction,
// each ALONE and BESIDE a second stream that keeps every CU's matrix pipe busy, and counts results that differ from fmaf(x0, c1, -p0).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/pk_fma_opsel.hip -o /tmp/pk_fma_opsel && /tmp/pk_fma_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((address_space(3))) void* las_ptr;

__device__ __forceinline__ f32x2 pk_fma_opsel(f32x2 x, f32x2 c, f32x2 p) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(x), "v"(c), "v"(p));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma_hi_from_lo(f32x2 x, f32x2 c, f32x2 p) {   // the OTHER cross-half direction: d.hi = x1 * c0 - p1 (op_sel_hi clears src1's bit)
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(x), "v"(c), "v"(p));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul_opsel(f32x2 x, f32x2 c) {                   // v_pk_mul_f32 with the same cross-half select: d.lo = x0 * c1
  f32x2 d;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma_plain(f32x2 x, f32x2 c, f32x2 p) {      // the same instruction WITHOUT the cross-half select: d.lo = x0 * c0 - p0
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(x), "v"(c), "v"(p));
  return d;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* bad, unsigned long long* lanes_bad, const float* tab_g, int iters) {
  __shared__ __attribute__((aligned(16))) float tab[256 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  // tables: (cos, sin)-like pairs, per lane; x: values an fp16 GEMM accumulator would hold
  f32x2 c = {0.25f + 0.001f * lane, 0.75f - 0.002f * lane};
  f32x2 p = {0.125f * (lane & 7), 0.5f + 0.01f * lane};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.5f + 0.0625f * ((lane + e) & 15)); b[e] = (_Float16)(0.25f * ((lane * 3 + e) & 7)); }
  tab[tid * 4 + 0] = c[0]; tab[tid * 4 + 1] = c[1]; tab[tid * 4 + 2] = p[0]; tab[tid * 4 + 3] = p[1];
  __syncthreads();
  unsigned long long nbad = 0, nbad_lo = 0, nbad_hi = 0, nzero = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x2 x = {1.5f + 0.01f * lane, -0.75f + 0.02f * lane};
  const unsigned lds_base = (unsigned)(size_t)(las_ptr)tab;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 6) {             // an MFMA in flight whose result is NOT an operand (consumed after the loop); x from registers
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
      x[0] += 0.001f;
    } else if constexpr (MODE >= 1) {      // x comes out of the matrix pipe, as in the kernel's epilogue
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      if constexpr (MODE == 4) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));      // 32 wait states behind the v_accvgpr_reads (the "+v" constraint reads acc first)
      x = f32x2{acc[0] + 0.001f * (it & 15), acc[1]};
    } else {
      x[0] += 0.001f;
    }
    if constexpr (MODE == 3) {             // the table arrives by LDS-DMA with hand-placed waits (every 64 iterations)
      if ((it & 63) == 0) {
        __syncthreads();
        unsigned keep;
        const float* src = tab_g + (size_t)(tid & ~63) * 4 + lane * 4;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (tid >> 6) * 1024);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        __syncthreads();
      }
    }
    f32x2 cc = c, pp = p;
    if constexpr (MODE == 2 || MODE == 3) {             // tables from LDS immediately before the instruction, the kernel's own wait
      const unsigned addr = lds_base + tid * 16;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(cc), "=&v"(pp) : "v"(addr) : "memory");
    }
    const f32x2 d = MODE == 5 ? pk_fma_plain(x, cc, pp) : MODE == 7 ? pk_fma_hi_from_lo(x, cc, pp) : MODE == 8 ? pk_mul_opsel(x, cc) : pk_fma_opsel(x, cc, pp);
    float want_lo = __builtin_fmaf(x[0], MODE == 5 || MODE == 7 ? cc[0] : cc[1], -pp[0]), want_hi = __builtin_fmaf(x[1], MODE == 7 ? cc[0] : cc[1], -pp[1]);
    if (MODE == 8) { want_lo = x[0] * cc[1]; want_hi = x[1] * cc[1]; }
    if (d[0] != want_lo) ++nbad_lo;
    if (d[1] != want_hi) ++nbad_hi;
    if (d[0] != want_lo && d[0] == (MODE == 8 ? 0.f : -pp[0])) ++nzero;              // the signature of round 3: the low PRODUCT came out as 0
  }
  nbad = nbad_lo + nbad_hi;
  if (MODE == 6) nbad += (acc[0] != acc[0]);                     // keep the accumulator alive
  if (nbad) { atomicAdd(bad, nbad); atomicAdd(lanes_bad + (lane >> 4), 1ull); atomicAdd(lanes_bad + 4, nbad_lo); atomicAdd(lanes_bad + 5, nbad_hi); atomicAdd(lanes_bad + 6, nzero); }
}

__global__ __launch_bounds__(512) void mfma_noise(float* sink, int iters) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((threadIdx.x + e) & 7); b[e] = (_Float16)1.f; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* what, bool noise, unsigned long long* d_bad, const float* tab_g, float* sink, hipStream_t s0, hipStream_t s1) {
  const int iters = 40000;                                  // x 1024 workgroups x 4 waves = 1.6e8 wave-instructions = 1e10 lane results
  hipMemsetAsync(d_bad, 0, 8 * sizeof(unsigned long long), s0);
  hipStreamSynchronize(s0);
  if (noise) hipLaunchKernelGGL(mfma_noise, dim3(1024), dim3(512), 0, s1, sink, 60000);
  hipLaunchKernelGGL((probe<MODE>), dim3(1024), dim3(256), 0, s0, d_bad, d_bad + 1, tab_g, iters);
  hipStreamSynchronize(s0);
  hipStreamSynchronize(s1);
  unsigned long long h[8];
  hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost);
  std::printf("%-70s %-22s wrong: %llu of %.1e (low half %llu [of them exactly -p0, i.e. product = 0: %llu], high half %llu); waves with a wrong lane in lanes 0-15 / 16-31 / 32-47 / 48-63: %llu %llu %llu %llu\n",
              what, noise ? "beside an MFMA stream" : "alone", h[0], 1024.0 * 256 * iters, h[5], h[7], h[6], h[1], h[2], h[3], h[4]);
}

int main() {
  unsigned long long* d_bad;
  float *tab_g, *sink;
  hipMalloc(&d_bad, 8 * sizeof(unsigned long long));
  hipMalloc(&tab_g, 256 * 4 * sizeof(float));
  hipMalloc(&sink, 1024 * 512 * sizeof(float));
  float h[256 * 4];
  for (int t = 0; t < 256; ++t) {
    const int lane = t & 63;
    h[t * 4 + 0] = 0.25f + 0.001f * lane; h[t * 4 + 1] = 0.75f - 0.002f * lane; h[t * 4 + 2] = 0.125f * (lane & 7); h[t * 4 + 3] = 0.5f + 0.01f * lane;
  }
  hipMemcpy(tab_g, h, sizeof(h), hipMemcpyHostToDevice);
  hipStream_t s0, s1;
  hipStreamCreate(&s0); hipStreamCreate(&s1);
  for (int noise = 0; noise < 2; ++noise) {
    run<0>("A  operands in plain registers", noise, d_bad, tab_g, sink, s0, s1);
    run<1>("B  x from an MFMA just before, tables in registers", noise, d_bad, tab_g, sink, s0, s1);
    run<2>("C  B + tables by ds_read_b64 + s_waitcnt lgkmcnt(0)", noise, d_bad, tab_g, sink, s0, s1);
    run<3>("D  C + the LDS table filled by global_load_lds + vmcnt(0) + barrier", noise, d_bad, tab_g, sink, s0, s1);
    run<4>("E  B + 32 wait states (2 x s_nop 15) between v_accvgpr_read and the packed instruction", noise, d_bad, tab_g, sink, s0, s1);
    run<5>("F  B with op_sel:[0,0,0] (no cross-half select)", noise, d_bad, tab_g, sink, s0, s1);
    run<6>("G  an MFMA in flight whose result is not an operand; x in registers", noise, d_bad, tab_g, sink, s0, s1);
    run<7>("H  B with the other cross-half direction (op_sel_hi:[1,0,1]: d.hi = x1 * c0 - p1)", noise, d_bad, tab_g, sink, s0, s1);
    run<8>("I  B with v_pk_mul_f32 op_sel:[0,1] (d.lo = x0 * c1)", noise, d_bad, tab_g, sink, s0, s1);
  }
  return 0;
}
