// airfe — large-M dense GEMM: 256 rows x 256 features per accumulation, 8 waves (2 per SIMD, each 64 rows x 128
// features), K streamed in 64-wide chunks through a two-stage LDS ring filled by LDS-DMA (global_load_lds: no staging
// VGPRs, no ds_write pass), 64 MFMAs per wave per barrier.
//
// Why this shape: the activations of the matcher (29-59 MB per tensor at 64 pairs) do not fit the 4 MiB per-XCD L2, so
// every pass over the feature dimension re-reads X from HBM/Infinity Cache.  With 128-feature passes the LightGlue
// linears were plainly bandwidth-bound (176 MB moved for a 15 GFLOP GEMM = the measured 44 us).  256-feature passes
// halve the X re-reads (one pass for N = 256, two for N = 512) and raise the MFMA : ds_read ratio to 8 : 3.
// The DMA wait sits before this iteration's output stores, so stores never sit between a DMA and its wait.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr8;

__device__ __forceinline__ void g8_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int G8_STAGE = 256 * 128 + 4 * SLAB_BYTES;     // X chunk [256 rows][64 k] + four weight slabs = 64 KiB

// epilogue helper shared with kernels_mm.hip's kernel (same accumulator mapping): defined there, declared here
template <class P>
__device__ __forceinline__ void g8_store_run(const GemmArgs& a, int row, int co0, float* v) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += a.bias[co0 + e];
  if (a.act == ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  switch (a.epi) {
    case EPI_STORE: {
      if (co0 < a.ldo)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + (size_t)row * a.ldo + co0) = pack8<P>(v);
      break;
    }
    case EPI_STORE_F32: {
      if (co0 < a.ldo) {
        float* o = reinterpret_cast<float*>(a.out) + (size_t)row * a.ldo + co0;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      break;
    }
    case EPI_RESID: {
      float* xr = a.x32 + (size_t)row * a.ldo + co0;
      const float4 r0 = *reinterpret_cast<float4*>(xr), r1 = *reinterpret_cast<float4*>(xr + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
      v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      *reinterpret_cast<float4*>(xr) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(xr + 4) = make_float4(v[4], v[5], v[6], v[7]);
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.out) + (size_t)row * a.ldo + co0) = pack8<P>(v);
      break;
    }
    case EPI_HEADS: {
      const int s = row / a.Np, n = row - s * a.Np;
      const int sel = co0 >> 8, cw = co0 & 255, h = cw >> 6, d = cw & 63;
      if (a.rot_cos) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(a.rot_cos + (size_t)row * 32 + (d >> 1));
        const f32x4 sn = *reinterpret_cast<const f32x4*>(a.rot_sin + (size_t)row * 32 + (d >> 1));
        rotate_pairs(v, c, sn);
      }
      uint16_t* o = reinterpret_cast<uint16_t*>(sel ? a.out2 : a.out) + (((size_t)s * a.H + h) * a.Np + n) * 64 + d;
      *reinterpret_cast<uint4*>(o) = pack8<P>(v);
      break;
    }
    default: break;
  }
}

template <class P, bool TRANS>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs a, int K, int quads_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves, each 64 rows x 128 features
  const int m0 = blockIdx.x * 256;
  const int NS = K >> 6;
  const int quad0 = blockIdx.y * quads_per_block;
  const int total = quads_per_block * NS;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr8)smem);

  // DMA coordinates: X piece q = tid + 512 i (i < 4) -> row q>>3, LDS slot q&7 holds channel chunk (q&7) ^ swz(row)
  const int xr = tid >> 3;                              // + 64 i ; rows 64 apart share the swizzle
  const int xcs = ((tid & 7) ^ swz128(xr)) * 8;         // source element offset inside the 64-wide chunk
  const int K1 = a.K1;
  const uint16_t* xp1 = a.X1 + (size_t)(m0 + xr) * a.ld1 + xcs;
  const uint16_t* xp2 = a.X2 ? a.X2 + (size_t)(m0 + xr) * a.ld2 + xcs : xp1;
  const size_t rstep1 = (size_t)64 * a.ld1, rstep2 = (size_t)64 * a.ld2;
  // gather form (a.rowidx): this thread's four rows come from X1 rows rowidx[m0 + xr + 64 i]
  constexpr bool GATHER = !TRANS;                        // (the transposed form has no registers to spare and no caller that gathers)
  const uint16_t* xg[GATHER ? 4 : 1];
  if constexpr (GATHER) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xg[i] = a.rowidx ? a.X1 + (size_t)a.rowidx[m0 + xr + 64 * i] * a.ld1 + xcs : xp1 + i * rstep1;
  }
  const char* wbase = reinterpret_cast<const char*>(a.Wp) + (size_t)tid * 16;

  auto dma = [&](int it) {
    const int qd = quad0 + it / NS, s = it % NS, k0 = s * 64;
    const bool first = k0 < K1;                         // K1 % 64 == 0: a chunk lies entirely on one side of cat(x, msg)
    const uint16_t* src = first ? xp1 + k0 : xp2 + (k0 - K1);
    const size_t rs = first ? rstep1 : rstep2;
    const unsigned dst = lds_base + (it & 1) * G8_STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (GATHER) g8_glds16(first ? xg[i] + k0 : src + i * rs, dst + i * 8192);
      else g8_glds16(src + i * rs, dst + i * 8192);
    }
    const char* w0 = wbase + ((size_t)(4 * qd) * NS + s) * SLAB_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) g8_glds16(w0 + (size_t)i * NS * SLAB_BYTES, dst + 32768 + i * 8192);
  };

  if (0 < total) dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // fragment offsets (loop invariant)
  int xoff[4], woff[8];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int r = wm * 64 + m * 16 + l15;
    xoff[m] = r * 128 + ((g ^ swz128(r)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int rr = (t & 3) * 16 + l15;
    woff[t] = 32768 + (wn * 2 + (t >> 2)) * SLAB_BYTES + rr * 128 + ((g ^ swz128(rr)) << 4);
  }

  f32x4 acc[4][8];
  for (int it = 0; it < total; ++it) {
    const int s = it % NS;
    if (s == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (it + 1 < total) dma(it + 1);                    // lands in the stage that was read during iteration it-1
    const char* st = smem + (it & 1) * G8_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename P::vec8 xf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) xf[m] = lds_frag<P>(st, xoff[m] ^ (ks << 6));
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {          // two 64-feature halves: keeps only 4 weight fragments live at a time
        typename P::vec8 wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = lds_frag<P>(st, woff[hb * 4 + t] ^ (ks << 6));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if constexpr (TRANS) acc[m][hb * 4 + t] = P::mfma(xf[m], wf[t], acc[m][hb * 4 + t]);
            else acc[m][hb * 4 + t] = P::mfma(wf[t], xf[m], acc[m][hb * 4 + t]);
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // chunk it+1 has landed (and last iteration's stores retired)
    __syncthreads();
    if (s == NS - 1) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int cb = 4 * (quad0 + it / NS) + wn * 2 + hb;
        if (cb < a.cb_total) {
          if constexpr (!TRANS) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int row = m0 + wm * 64 + m * 16 + l15;
#pragma unroll
              for (int tp = 0; tp < 2; ++tp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v[e] = acc[m][hb * 4 + 2 * tp][e];
                  v[4 + e] = acc[m][hb * 4 + 2 * tp + 1][e];
                }
                g8_store_run<P>(a, row, cb * 64 + tp * 32 + g * 8, v);
              }
            }
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int co = cb * 64 + slab_row_to_feature(t * 16 + l15);
              const float bv = a.bias[co];
              const int h = co >> 6, d = co & 63;
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                const int row0 = m0 + wm * 64 + m * 16 + g * 4;
                const int sq = row0 / a.Np, n = row0 - sq * a.Np;
                uint16_t* o = reinterpret_cast<uint16_t*>(a.out) + (((size_t)sq * a.H + h) * 64 + d) * a.Np + n;
                const f32x4 c = acc[m][hb * 4 + t];
                *reinterpret_cast<uint2*>(o) = pack4<P>(c[0] + bv, c[1] + bv, c[2] + bv, c[3] + bv);
              }
            }
          }
        }
      }
    }
  }
}

template <class P, bool TRANS>
static void gemm8_launch_t(int K, const GemmArgs& a, hipStream_t st) {
  constexpr int LDS = 2 * G8_STAGE;
  static PerDeviceOnce attr_once;
  auto kfn = gemm8_kernel<P, TRANS>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  const int mb = a.M / 256;
  const int nquads = (a.cb_total + 3) / 4;       // the weight buffer is packed with a multiple of 4 feature blocks
  int gy = 1;
  while (gy < nquads && (mb * gy < 224 || nquads % gy)) ++gy;      // one 8-wave workgroup per CU
  hipLaunchKernelGGL(kfn, dim3((unsigned)mb, (unsigned)gy), dim3(512), LDS, st, a, K, nquads / gy);
}

// ================================================================================ detector head as a STREAMING kernel (round 3)
// convPb (1x1, 256 -> 65) + soft-max over the 65 logits of a cell + 8x8 depth-to-space: 512 B in and 256 B out per cell for 33 kFLOP — HBM
// work.  As a 256 x 128 tile GEMM with a three-stage ring (gemm8_kernel with a soft-max epilogue, rounds 1-2) it ran at 2.1 TB/s, one workgroup per CU and four K chunks per
// tile being all fill and drain (190 us per 128 images, MFMA-busy 0.12).  Here the 65 x 256 head sits in LDS as MFMA A fragments
// (fragment-major, 40 KB, read conflict-free), every wave streams tiles of 16 cells straight from global memory into B fragments, one tile
// ahead, and writes the score map from its accumulators: no staging of activations, no barrier in the loop, three waves per SIMD.
// Same fragments, same k order (eight 32-wide steps, ascending), bias after the sum and the soft-max expressions of the tiled kernel's former
// epilogue: the same bits (profiles/r03_probe11_*: md5 of the score maps and features equal; stage 0.38 -> 0.31 ms per step).
template <class P>
__global__ __launch_bounds__(256, 3) void head_softmax_d2s_kernel(const uint16_t* __restrict__ X /*[ncell][256]*/, const uint16_t* __restrict__ Wp,
                                                                  const float* __restrict__ bias /*[65..]*/, float* __restrict__ heat, int ntiles,
                                                                  int hc, int wc, int* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) char wl[5 * 8 * 1024];      // [tile u][k-step][lane] 16-byte fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  {
    // tile u = 0..3: rows 16 u + l15 of the first 64-feature slab group, u = 4: row l15 of the second (feature 64 = the dustbin, lane row 0)
    const char* wbase = reinterpret_cast<const char*>(Wp);
    for (int f = wave; f < 40; f += 4) {
      const int u = f >> 3, ks = f & 7;
      const int cb = u >> 2, rr = (u & 3) * 16 + l15;
      const uint4 v = *reinterpret_cast<const uint4*>(wbase + ((size_t)cb * 4 + (ks >> 1)) * SLAB_BYTES + rr * 128 + ((((ks & 1) * 4 + g) ^ swz128(rr)) << 4));
      *reinterpret_cast<uint4*>(wl + (f * 64 + lane) * 16) = v;
    }
  }
  float bv[2][8];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[tp][e] = bias[tp * 32 + g * 8 + e];
  const float bdust = bias[64];
  __syncthreads();

  const int nw = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  typename P::vec8 xn[8];
  auto fetch = [&](int t) {
    const uint16_t* xr = X + ((size_t)t * 16 + l15) * 256 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xn[ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(xr + ks * 32));
  };
  if (tile < ntiles) fetch(tile);
  const int per = hc * wc;
  for (; tile < ntiles; tile += nw) {
    typename P::vec8 xf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xf[ks] = xn[ks];
    if (tile + nw < ntiles) fetch(tile + nw);
    f32x4 acc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int woff = lane * 16;                                // opaque per tile: otherwise hipcc hoists the 40 loop-invariant fragment reads out of
    asm volatile("" : "+v"(woff));                       // the loop (160 registers: 428 bytes of scratch per lane at three waves per SIMD)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int u = 0; u < 5; ++u) acc[u] = P::mfma(lds_frag<P>(wl, (u * 8 + ks) * 1024 + woff), xf[ks], acc[u]);
    // ---- the epilogue of gemm8's EPI_SOFTMAX_D2S, expression by expression
    const int row = tile * 16 + l15;
    float v[2][8];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[tp][e] = acc[2 * tp][e] + bv[tp][e];
        v[tp][4 + e] = acc[2 * tp + 1][e] + bv[tp][4 + e];
      }
    const float dust = (g == 0) ? acc[4][0] + bdust : -INFINITY;
    float mx = dust;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[tp][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = expf(dust - mx);
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[tp][e] = expf(v[tp][e] - mx);
        sum += v[tp][e];
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // a cell whose logits are inf / NaN (its sum of exponentials is then not a finite number): the 2-byte activations upstream overflowed — the host entries turn
    // the flag into an ERROR instead of handing out keypoints of a poisoned score map (one compare per cell; airfe.h "activation range")
    if (flag && g == 0 && !(sum <= 3.0e38f)) *reinterpret_cast<volatile int*>(flag) = 1;      // (host-mapped word: written only when it happens)
    const float inv = 1.0f / sum;
    const int b = row / per, rem = row - b * per, cy = rem / wc, cx = rem - cy * wc;
    float* o = heat + ((size_t)b * hc * 8 + (size_t)cy * 8) * (wc * 8) + cx * 8;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      float* r = o + (size_t)(tp * 4 + g) * (wc * 8);
      *reinterpret_cast<float4*>(r) = make_float4(v[tp][0] * inv, v[tp][1] * inv, v[tp][2] * inv, v[tp][3] * inv);
      *reinterpret_cast<float4*>(r + 4) = make_float4(v[tp][4] * inv, v[tp][5] * inv, v[tp][6] * inv, v[tp][7] * inv);
    }
  }
}

// requires M % 256 == 0, K % 64 == 0, K1 % 64 == 0
void launch_gemm8(int prec, int K, bool trans, const GemmArgs& a, hipStream_t st) {
  if (a.epi == EPI_SOFTMAX_D2S) {                          // the detector head: its own streaming kernel (above); K = 256, N = 65, rows = cells
    if (trans || K != 256 || a.N != 65 || a.X2 || a.rowidx || a.ld1 != 256 || a.M % 16 != 0) {           // (airfe_detect.hip builds no other form)
      fprintf(stderr, "airfe: EPI_SOFTMAX_D2S is the detector head only (K = 256, N = 65, dense rows)\n");
      abort();
    }
    const int ntiles = a.M / 16;
    const int wgs = std::min((ntiles + 3) / 4, 256 * 4);      // persistent: four 4-wave workgroups per CU (40 KB of LDS each)
    if (prec == 1) hipLaunchKernelGGL(head_softmax_d2s_kernel<PF16>, dim3(wgs), dim3(256), 0, st, a.X1, a.Wp, a.bias, reinterpret_cast<float*>(a.out), ntiles, a.d2s_hc, a.d2s_wc, a.flag);
    else hipLaunchKernelGGL(head_softmax_d2s_kernel<PBF16>, dim3(wgs), dim3(256), 0, st, a.X1, a.Wp, a.bias, reinterpret_cast<float*>(a.out), ntiles, a.d2s_hc, a.d2s_wc, a.flag);
    return;
  }
  if (prec == 1) {
    if (trans) gemm8_launch_t<PF16, true>(K, a, st); else gemm8_launch_t<PF16, false>(K, a, st);
  } else {
    if (trans) gemm8_launch_t<PBF16, true>(K, a, st); else gemm8_launch_t<PBF16, false>(K, a, st);
  }
}

}  // namespace airfe
