// airfe — LightGlue post-attention block, FEATURE-SPLIT form:  msg = Wo·attn + bo;  h = GELU(LN(W1·cat(x, msg) + b1));
// x += W2·h + b2  for 128 tokens per workgroup, as ONE kernel.
//
// kernels_lgblock.hip keeps a token's whole chain inside one wave (msg and h never leave registers), which forces every wave
// to read every weight through LDS, one wave per SIMD, one workgroup-wide barrier per 32 KiB of weights.  This form is the
// other cut: the 8 waves (two per SIMD) split the OUTPUT FEATURES of each GEMM, every wave fetches only its own rows of the
// packed weight slabs straight from global memory into MFMA A-fragments (each weight byte is read once per workgroup, no LDS,
// no barrier), and the activations are the B operand shared through LDS: attn / x / msg tiles of [128][256] 2-byte (64 KiB)
// and the h tile of [128][512] (128 KiB, over the dead attn/x and msg tiles).  Five workgroup barriers per tile instead of 28.
//   LDS:  R0 [0, 64K) attn -> x -> h(lo)   R1 [64K, 128K) msg -> h(hi)   ST [128K, 136K) LayerNorm partial sums
//   rows are swizzled by XOR of the 16-byte piece index with (token & 15): conflict-free ds_read_b128 / ds_write_b128.
// Weights are the SAME packed slabs the separate launches use (LinW::w, [N/64][K/64][8 KiB]); the feature order inside a slab
// (slab_row_to_feature) is what makes a lane's accumulators 8 contiguous features of one token.
#include "common.h"
#include "kernels.h"

namespace airfe {

constexpr int LF_TM = 128;
constexpr int LF_R0 = 0, LF_R1 = 65536, LF_ST = 131072;
constexpr int LF_LDS = LF_ST + 8 * LF_TM * 8;
#ifndef LF_XDMA_SLAB
#define LF_XDMA_SLAB 3
#endif

__device__ __forceinline__ void lf_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

// GELU(y) = 0.5 y (1 + erf(y / sqrt 2)) = max(y, 0) - 0.5 |y| erfc(|y| / sqrt 2), with erfc(x) = 2^(-x^2 log2 e + R(x)),
// R = log2(erfcx) a degree-8 polynomial in u = x/3 - 1 on x in [0, 6] (Chebyshev fit, |dR| < 1e-5; beyond 6 erfc < 3e-17).
// One transcendental instead of the two (rcp, exp) of Abramowitz & Stegun 7.1.26 in ln_gelu_kernel, no cancellation on the
// negative side (max abs error 1.1e-6, relative 2e-4 in the far negative tail where 7.1.26 returns 0), and written on float
// PAIRS so that the 19 multiply-adds become v_pk_fma_f32 / v_pk_mul_f32: this epilogue is 128 elements per lane per tile and
// was as long as the tile's MFMA time.
__device__ __forceinline__ f32x2 lf_gelu2(f32x2 y) {
  const f32x2 a = __builtin_elementwise_abs(y);
  f32x2 x = a * 0.70710678118654752f;
  x = __builtin_elementwise_min(x, f32x2{6.0f, 6.0f});
  const f32x2 u = x * (1.0f / 3.0f) - 1.0f;
  f32x2 r = u * 0.008399954997003078f - 0.02763195149600506f;
  r = r * u + 0.04838801920413971f;
  r = r * u - 0.08357185125350952f;
  r = r * u + 0.1567264348268509f;
  r = r * u - 0.28920042514801025f;
  r = r * u + 0.5534077286720276f;
  r = r * u - 1.3146467208862305f;
  r = r * u - 2.481963634490967f;
  const f32x2 p = (x * x) * -1.4426950408889634f + r;
  const f32x2 q = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  return __builtin_elementwise_max(y, f32x2{0.f, 0.f}) - (x * 0.70710678118654752f) * q;
}

// [128 tokens][256] 2-byte rows of `src` -> LDS region (64 KiB) by LDS-DMA: 64 wave-instructions of 1 KiB (two rows each), eight
// per wave; lane i of an instruction lands at +16 i, so the swizzle is applied on the global side.
__device__ __forceinline__ void lf_stage_rows(const uint16_t* src, int m0, int region, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int inst = wave * 8 + i;
    const int r = inst * 2 + (lane >> 5), pp = lane & 31;
    lf_glds16(src + (size_t)(m0 + r) * 256 + ((pp ^ (r & 15)) << 3), (unsigned)(region + inst * 1024));
  }
}

template <class P>
__device__ __forceinline__ typename P::vec8 lf_ldg(const char* p) {
  return __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(p));
}

// The first slab's A fragments of a GEMM: issued ahead of the phase that precedes it (barrier, pack, GELU), so that a GEMM
// never starts by waiting one L2 round trip.
template <class P, int NT>
__device__ __forceinline__ void lf_first(typename P::vec8 (&cur)[NT][2], const char* w0, const char* w1) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    cur[t][0] = lf_ldg<P>(w0 + t * 2048);
    cur[t][1] = lf_ldg<P>(w1 + t * 2048);
  }
}

// acc[t][m] += W(tiles t of this wave, slabs 0..nslab-1) x B(token tiles m = 0..7): w0 / w1 = this lane's fragment address in
// slab 0, tile 0 for the two 32-wide halves of the slab's 64-wide K chunk; tile t at +2048 t, slab s at +8192 s.  `cur` holds
// slab 0 on entry; the next slab's fragments are fetched while the current one feeds 16 NT MFMAs, and during the last slab the
// fetch goes to n0 / n1 (the first slab of whatever this wave multiplies next), which `cur` holds on exit.
struct LfNoHook { __device__ __forceinline__ void operator()(int) const {} };

template <class P, int NT, class Hook = LfNoHook>
__device__ __forceinline__ void lf_mma(f32x4 (&acc)[NT][8], typename P::vec8 (&cur)[NT][2], const char* w0, const char* w1, int nslab,
                                       const char* n0, const char* n1, const char* breg, int pitch, int l15, int g, Hook hook = Hook()) {
  typename P::vec8 nxt[NT][2];
  const char* brow = breg + l15 * pitch;
#pragma unroll 1
  for (int s = 0; s < nslab; ++s) {
    const bool last = s + 1 == nslab;
    const char* p0 = last ? n0 : w0 + (s + 1) * SLAB_BYTES;
    const char* p1 = last ? n1 : w1 + (s + 1) * SLAB_BYTES;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      nxt[t][0] = lf_ldg<P>(p0 + t * 2048);
      nxt[t][1] = lf_ldg<P>(p1 + t * 2048);
    }
    hook(s);                                // vector-memory work that must queue BEHIND this trip's prefetch (vmcnt retires in order)
    __builtin_amdgcn_sched_barrier(0);      // keep the whole prefetch at the top of the trip (hipcc sinks loads towards their use)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int boff = (((s * 8 + h * 4 + g) ^ l15) << 4);
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        typename P::vec8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = lds_frag<P>(brow, boff + (mh * 4 + j) * 16 * pitch);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][mh * 4 + j] = P::mfma(cur[t][h], bf[j], acc[t][mh * 4 + j]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      cur[t][0] = nxt[t][0];
      cur[t][1] = nxt[t][1];
    }
  }
}

// The 32-feature GEMMs (out-proj, ffn.3): one slab is only 32 MFMAs per wave, less than an L2 round trip, so the A fragments
// run THREE slabs ahead through a four-buffer ring (fully unrolled: the ring index is a compile-time constant).  `first`
// holds slab 0 on entry.
template <class P, int NSLAB>
__device__ __forceinline__ void lf_mma2(f32x4 (&acc)[2][8], typename P::vec8 (&first)[2][2], const char* w0, const char* w1,
                                        const char* breg, int pitch, int l15, int g) {
  typename P::vec8 ring[4][2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    ring[0][t][0] = first[t][0];
    ring[0][t][1] = first[t][1];
  }
#pragma unroll
  for (int s = 1; s < 3 && s < NSLAB; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ring[s][t][0] = lf_ldg<P>(w0 + s * SLAB_BYTES + t * 2048);
      ring[s][t][1] = lf_ldg<P>(w1 + s * SLAB_BYTES + t * 2048);
    }
  const char* brow = breg + l15 * pitch;
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    if (s + 3 < NSLAB) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ring[(s + 3) & 3][t][0] = lf_ldg<P>(w0 + (s + 3) * SLAB_BYTES + t * 2048);
        ring[(s + 3) & 3][t][1] = lf_ldg<P>(w1 + (s + 3) * SLAB_BYTES + t * 2048);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int boff = (((s * 8 + h * 4 + g) ^ l15) << 4);
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        typename P::vec8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = lds_frag<P>(brow, boff + (mh * 4 + j) * 16 * pitch);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][mh * 4 + j] = P::mfma(ring[s & 3][t][h], bf[j], acc[t][mh * 4 + j]);
      }
    }
  }
}

#ifdef LF_TIMING
__device__ long long lf_dbg[512 * 8 * 12];
#define LF_STAMP(i) if (lane == 0 && blockIdx.x < 512) lf_dbg[(blockIdx.x * 8 + wave) * 12 + (i)] = wall_clock64();
#else
#define LF_STAMP(i)
#endif

template <class P>
__global__ __launch_bounds__(512, 1) void lg_blockf_kernel(LgBlockFArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * LF_TM;
  const int sw = (l15 >> 1) & 7;                                  // swz128 of this lane's slab row (same for every tile)
  const int fo0 = l15 * 128 + ((g ^ sw) << 4), fo1 = l15 * 128 + (((4 + g) ^ sw) << 4);
  // Which feature block a wave owns rotates with the workgroup (among the workgroups of one XCD: blockIdx / 8), so that the
  // 32 CUs of an XCD do not all ask its L2 for the same weight lines at the same moment.  Numerics do not depend on it.
  const int wf = (wave + (blockIdx.x >> 3)) & 7;
  const int cb = wf >> 1, tp = wf & 1;                            // 256-feature GEMMs: wave = 32 features (one tile pair)

  LF_STAMP(0)
  lf_stage_rows(a.attn, m0, LF_R0, wave, lane);
  const char* wob = reinterpret_cast<const char*>(a.wo) + (size_t)cb * 4 * SLAB_BYTES + 2 * tp * 2048;
  const char* w1b = reinterpret_cast<const char*>(a.w1) + (size_t)wf * 8 * SLAB_BYTES;
  const char* w2b = reinterpret_cast<const char*>(a.w2) + (size_t)cb * 8 * SLAB_BYTES + 2 * tp * 2048;
  typename P::vec8 c2[2][2], c4[4][2];                            // A fragments in flight: 32-feature GEMMs / the 64-feature one
  lf_first<P, 2>(c2, wob + fo0, wob + fo1);
  f32x4 bo2[2], b14[4];                                           // biases of the first two GEMMs: fetched with the attn tile
#pragma unroll
  for (int u = 0; u < 2; ++u) bo2[u] = *reinterpret_cast<const f32x4*>(a.bo + cb * 64 + tp * 32 + g * 8 + u * 4);
#pragma unroll
  for (int t = 0; t < 4; ++t) b14[t] = *reinterpret_cast<const f32x4*>(a.b1 + wf * 64 + (t >> 1) * 32 + g * 8 + (t & 1) * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LF_STAMP(1)

  // ---- msg = Wo attn + bo -> R1
  {
    f32x4 acc[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[u][m] = bo2[u];
    lf_mma2<P, 4>(acc, c2, wob + fo0, wob + fo1, smem + LF_R0, 512, l15, g);
    lf_first<P, 4>(c4, w1b + 4 * SLAB_BYTES + fo0, w1b + 4 * SLAB_BYTES + fo1);      // lands while msg is packed and the barrier drains
    const int piece = cb * 8 + tp * 4 + g;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[0][m][e];
        v[4 + e] = acc[1][m][e];
      }
      *reinterpret_cast<uint4*>(smem + LF_R1 + (m * 16 + l15) * 512 + ((piece ^ l15) << 4)) = pack8<P>(v);
    }
  }
  __syncthreads();                                                // msg complete, attn dead
  LF_STAMP(2)

  // ---- h = W1 cat(x, msg) + b1: the msg half first; the x tile's DMA into R0 is issued behind the msg half's LAST weight
  // prefetch — vmcnt retires in order, so issued any earlier every wait for weights would also wait for the HBM-latency DMA
  f32x4 h[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < 8; ++m) h[t][m] = b14[t];
  lf_mma<P, 4>(h, c4, w1b + 4 * SLAB_BYTES + fo0, w1b + 4 * SLAB_BYTES + fo1, 4, w1b + fo0, w1b + fo1, smem + LF_R1, 512, l15, g,
               [&](int s) { if (s == LF_XDMA_SLAB) lf_stage_rows(a.xb, m0, LF_R0, wave, lane); });
  LF_STAMP(3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LF_STAMP(4)
  lf_mma<P, 4>(h, c4, w1b + fo0, w1b + fo1, 4, w1b + fo0, w1b + fo1, smem + LF_R0, 512, l15, g);
  LF_STAMP(5)

  // LayerNorm scale / shift of this wave's 64 features: fetched now, used after the next barrier
  f32x4 gam[2][2], bet[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      gam[q][u] = *reinterpret_cast<const f32x4*>(a.gamma + wf * 64 + q * 32 + g * 8 + u * 4);
      bet[q][u] = *reinterpret_cast<const f32x4*>(a.beta + wf * 64 + q * 32 + g * 8 + u * 4);
    }
  // ---- LayerNorm(512): per-wave partial sums over its 64 features, exchanged through ST
  {
    float2* st = reinterpret_cast<float2*>(smem + LF_ST);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1 += h[t][m][e];
          s2 = fmaf(h[t][m][e], h[t][m][e], s2);
        }
      s1 = rows_sum(s1);
      s2 = rows_sum(s2);
      if (g == 0) st[wf * LF_TM + m * 16 + l15] = make_float2(s1, s2);
    }
  }
  __syncthreads();                                                // sums visible; x and msg tiles dead
  LF_STAMP(6)
  // ffn.3's first weight fragments are fetched now, the fp32 residual rows half way: their latency hides under the GELU arithmetic
  const int co = cb * 64 + tp * 32 + g * 8;
  float* xr0 = a.x32 + (size_t)(m0 + l15) * 256 + co;
  float4 r0[8], r1[8];
  lf_first<P, 2>(c2, w2b + fo0, w2b + fo1);
  f32x4 b22[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) b22[u] = *reinterpret_cast<const f32x4*>(a.b2 + co + u * 4);
  __builtin_amdgcn_sched_barrier(0);
  {
    const float2* st = reinterpret_cast<const float2*>(smem + LF_ST);
    float nmr[8], rstd[8];                                        // (h - mean) rstd = h rstd + nmr
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float2 p = st[w * LF_TM + m * 16 + l15];
        s1 += p.x;
        s2 += p.y;
      }
      const float mean = s1 * (1.0f / 512.0f);
      const float var = fmaxf(s2 * (1.0f / 512.0f) - mean * mean, 0.f);
      rstd[m] = 1.0f / sqrtf(var + 1e-5f);
      nmr[m] = -mean * rstd[m];
    }
    // ---- GELU(LN(h)) -> h tile [128][512] 2-byte over R0 + R1
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 g0 = gam[q][0], g1 = gam[q][1], be0 = bet[q][0], be1 = bet[q][1];
      const int piece = wf * 8 + q * 4 + g;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const f32x2 rs = {rstd[m], rstd[m]}, nm = {nmr[m], nmr[m]};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 y0 = (f32x2{h[2 * q][m][e], h[2 * q][m][e + 1]} * rs + nm) * f32x2{g0[e], g0[e + 1]} + f32x2{be0[e], be0[e + 1]};
          const f32x2 y1 = (f32x2{h[2 * q + 1][m][e], h[2 * q + 1][m][e + 1]} * rs + nm) * f32x2{g1[e], g1[e + 1]} + f32x2{be1[e], be1[e + 1]};
#ifdef LF_NOGELU
          const f32x2 o0 = y0, o1 = y1;
#else
          const f32x2 o0 = lf_gelu2(y0), o1 = lf_gelu2(y1);
#endif
          v[e] = o0.x; v[e + 1] = o0.y;
          v[4 + e] = o1.x; v[5 + e] = o1.y;
        }
        *reinterpret_cast<uint4*>(smem + (m * 16 + l15) * 1024 + ((piece ^ l15) << 4)) = pack8<P>(v);
      }
      if (q == 0) {           // half of h's registers are free now: the residual rows take them
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          r0[m] = *reinterpret_cast<const float4*>(xr0 + (size_t)m * 16 * 256);
          r1[m] = *reinterpret_cast<const float4*>(xr0 + (size_t)m * 16 * 256 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  LF_STAMP(7)

  // ---- x += W2 h + b2
  {
    f32x4 acc[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[u][m] = b22[u];
    lf_mma2<P, 8>(acc, c2, w2b + fo0, w2b + fo1, smem, 1024, l15, g);
    LF_STAMP(8)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const size_t row = (size_t)(m0 + m * 16 + l15);
      float* xr = xr0 + (size_t)m * 16 * 256;
      float v[8] = {acc[0][m][0] + r0[m].x, acc[0][m][1] + r0[m].y, acc[0][m][2] + r0[m].z, acc[0][m][3] + r0[m].w,
                    acc[1][m][0] + r1[m].x, acc[1][m][1] + r1[m].y, acc[1][m][2] + r1[m].z, acc[1][m][3] + r1[m].w};
      *reinterpret_cast<float4*>(xr) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(xr + 4) = make_float4(v[4], v[5], v[6], v[7]);
      *reinterpret_cast<uint4*>(a.xb + row * 256 + co) = pack8<P>(v);
    }
  }
  LF_STAMP(9)
}

#ifdef LF_TIMING
extern "C" int airfe_dbg_lf(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lf_dbg), sizeof(long long) * 512 * 8 * 12); }
#endif

template <class P>
static void launch_f(const LgBlockFArgs& a, hipStream_t st) {
  static bool attr_done = false;
  auto kfn = lg_blockf_kernel<P>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)(a.M / LF_TM)), dim3(512), LF_LDS, st, a);
}

void launch_lg_blockf(int prec, const LgBlockFArgs& a, hipStream_t st) {
  if (prec == 1) launch_f<PF16>(a, st); else launch_f<PBF16>(a, st);
}

}  // namespace airfe
