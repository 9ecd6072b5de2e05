// airfe — MFMA kernels: implicit-GEMM 3x3 convolution (NHWC, LDS-staged halo tile) and the dense
// "X-stationary" GEMM used by 1x1 convs and every Linear/Conv1d of the matchers.
//
// Both kernels use the same scheme (gfx950, wave64, v_mfma_f32_16x16x32_{bf16,f16}):
//   * the activation tile (halo'ed pixels x CIN, or BM rows x K) is staged ONCE into LDS with a
//     16-byte-chunk XOR swizzle so every ds_read_b128 fragment read is bank-conflict free;
//   * weights stream through a double-buffered 8 KiB slab ([64 features][64 k], pre-swizzled on
//     the host so staging is a linear copy), one barrier per slab, next slab prefetched in VGPRs;
//   * operands are SWAPPED (A = weight rows, B = pixels/tokens): the accumulator of lane
//     (j = lane&15, g = lane>>4) then holds 8 contiguous output features of pixel j per tile pair,
//     i.e. the epilogue (bias, ReLU, 2x2 max-pool, rotary, residual, ...) works on whole feature
//     runs of one pixel and stores 16-byte vectors.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace airfe {

template <int CIN>
__device__ __forceinline__ int swz_pix(int p) {
  if constexpr (CIN == 64) return swz128(p);
  else return swz256(p);
}

// ================================================================================== conv 3x3
template <class P, int CIN, int MR, bool POOL>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a, int tiles_x, int tiles_y, int cb_per_block) {
  constexpr int TH = 4 * MR, TW = 16, PH = TH + 2, PW = TW + 2;
  constexpr int PIXB = CIN * 2, CPP = CIN / 8, NS = 9 * (CIN / 64);
  constexpr int TILE_BYTES = PH * PW * PIXB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* ws = smem + TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int bid = blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
  const int H = a.H, W = a.W, COUT = a.COUT;

  // ---- stage the halo tile (input has a zero border, so no bounds tests)
  const size_t in_row = (size_t)(W + 2) * PIXB;
  const char* xin = (const char*)a.X + ((size_t)b * (H + 2) + (size_t)ty * TH) * in_row + (size_t)tx * TW * PIXB;
  for (int q = tid; q < PH * PW * CPP; q += 256) {
    const int p = q / CPP, c = q % CPP;
    const int pr = p / PW, pc = p % PW;
    const uint4 v = *reinterpret_cast<const uint4*>(xin + (size_t)pr * in_row + pc * PIXB + c * 16);
    *reinterpret_cast<uint4*>(xs + p * PIXB + ((c ^ swz_pix<CIN>(p)) << 4)) = v;
  }

  const int cb0 = blockIdx.y * cb_per_block;
  const int total = cb_per_block * NS;
  const uint4* wsrc = reinterpret_cast<const uint4*>(a.Wp) + (size_t)cb0 * NS * 512;
  uint4 pre0 = wsrc[tid], pre1 = wsrc[256 + tid];
  reinterpret_cast<uint4*>(ws)[tid] = pre0;
  reinterpret_cast<uint4*>(ws)[256 + tid] = pre1;
  __syncthreads();

  f32x4 acc[MR][4];
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int opad = a.out_pad;

  for (int it = 0; it < total; ++it) {
    const int s = it % NS;
    if (s == 0) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (it + 1 < total) {
      pre0 = wsrc[(size_t)(it + 1) * 512 + tid];
      pre1 = wsrc[(size_t)(it + 1) * 512 + 256 + tid];
    }
    const int tap = s / (CIN / 64), cc = s % (CIN / 64);
    const int dy = tap / 3, dx = tap % 3;
    const char* wb = ws + (it & 1) * SLAB_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename P::vec8 af[4], bf[MR];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rr = t * 16 + l15;
        af[t] = lds_frag<P>(wb, rr * 128 + (((ks * 4 + g) ^ swz128(rr)) << 4));
      }
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const int p = (wave * MR + m + dy) * PW + l15 + dx;
        const int c = cc * 8 + ks * 4 + g;
        bf[m] = lds_frag<P>(xs, p * PIXB + ((c ^ swz_pix<CIN>(p)) << 4));
      }
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = P::mfma(af[t], bf[m], acc[m][t]);
    }
    if (it + 1 < total) {
      char* wn = ws + ((it + 1) & 1) * SLAB_BYTES;
      reinterpret_cast<uint4*>(wn)[tid] = pre0;
      reinterpret_cast<uint4*>(wn)[256 + tid] = pre1;
    }
    if (s == NS - 1) {
      // ---- epilogue for cout block cb: bias, ReLU, optional 2x2 max-pool, 16-byte stores
      const int cb = cb0 + it / NS;
      const size_t orow = (size_t)(Wo + 2 * opad) * COUT;
      uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow;
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int co0 = cb * 64 + tp * 32 + g * 8;
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = a.bias[co0 + e];
        if constexpr (!POOL) {
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[m][2 * tp][e] + bias[e];
              v[4 + e] = acc[m][2 * tp + 1][e] + bias[4 + e];
            }
            if (a.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const int y = ty * TH + wave * MR + m, x = tx * TW + l15;
            *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT + co0) = pack8<P>(v);
          }
        } else {
#pragma unroll
          for (int mp = 0; mp < MR / 2; ++mp) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = fmaxf(acc[2 * mp][2 * tp][e], acc[2 * mp + 1][2 * tp][e]);
              v[4 + e] = fmaxf(acc[2 * mp][2 * tp + 1][e], acc[2 * mp + 1][2 * tp + 1][e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] = fmaxf(v[e], __shfl_xor(v[e], 1));
              v[e] += bias[e];
              if (a.relu) v[e] = fmaxf(v[e], 0.f);
            }
            if ((l15 & 1) == 0) {
              const int y = (ty * TH + wave * MR) / 2 + mp, x = (tx * TW + l15) / 2;
              *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT + co0) = pack8<P>(v);
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

template <class P, int CIN, int MR, bool POOL>
static void conv_launch_t(const ConvArgs& a, hipStream_t st) {
  constexpr int TH = 4 * MR;
  constexpr int LDS = (TH + 2) * 18 * CIN * 2 + 2 * SLAB_BYTES;
  static PerDeviceOnce attr_once;
  auto kfn = conv3x3_kernel<P, CIN, MR, POOL>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / TH;
  const int cbt = a.COUT / 64;
  const long blocks = (long)tiles_x * tiles_y * a.B;
  const int gy = (blocks >= 2048) ? 1 : cbt;
  dim3 grid((unsigned)blocks, (unsigned)gy);
  hipLaunchKernelGGL(kfn, grid, dim3(256), LDS, st, a, tiles_x, tiles_y, cbt / gy);
}

template <class P>
static void conv_launch_p(const ConvArgs& a, hipStream_t st) {
  if (a.CIN == 64) {
    if (a.pool) conv_launch_t<P, 64, 4, true>(a, st); else conv_launch_t<P, 64, 4, false>(a, st);
  } else {
    if (a.pool) conv_launch_t<P, 128, 2, true>(a, st); else conv_launch_t<P, 128, 2, false>(a, st);
  }
}

void launch_conv3x3(int prec, const ConvArgs& a, hipStream_t st) {
  if (a.CIN == 64 && (a.COUT % 64) == 0 && a.relu && (a.H % 16) == 0 && (a.W % 16) == 0) {
    launch_conv64r(prec, a, st);       // persistent kernel, filters resident in registers (kernels_conv64r.hip)
    return;
  }
  if (a.CIN == 128 && (a.COUT % 128) == 0 && a.relu && (a.H % 8) == 0 && (a.W % 16) == 0) {
    launch_conv128r(prec, a, st);      // persistent kernel, filters resident in registers (kernels_conv128r.hip)
    return;
  }
  if (prec == 1) conv_launch_p<PF16>(a, st); else conv_launch_p<PBF16>(a, st);
}

// ================================================================================== dense GEMM
template <class P>
__device__ __forceinline__ void gemm_store_run(const GemmArgs& a, int row, int co0, float* v) {
  // v[0..7] = features co0..co0+7 of row `row`, bias not yet added
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
      float4 r0 = *reinterpret_cast<float4*>(xr), r1 = *reinterpret_cast<float4*>(xr + 4);
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

// Tile: 128 rows x 128 features per accumulation, K streamed in 64-wide chunks.  Both operands go through a
// double-buffered LDS stage (X chunk [128][64] + two weight slabs = 32 KiB per stage) with the next chunk prefetched
// into VGPRs while the current one feeds 32 MFMAs per wave.  A workgroup walks ALL its feature-block pairs for one row
// tile in a single continuous (pair, k-chunk) stream, so the pipeline never drains between output tiles; X chunks are
// re-read from L2 per pair (the 64-128 KiB row tile is L2-resident), which keeps every load overlapped with MFMAs —
// the earlier "X-stationary" version staged the whole row tile up front and spent 70 % of its wave cycles waiting.
template <class P, bool TRANS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs a, int K, int pairs_per_block) {
  constexpr int STAGE = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 128;
  const int NS = K >> 6;
  const int pair0 = blockIdx.y * pairs_per_block;
  const int total = pairs_per_block * NS;

  // per-thread staging coordinates: X chunk = 1024 16-byte pieces; thread t owns piece (t & 7) of rows (t >> 3) + 32 i
  const int xr = tid >> 3, xc = tid & 7;
  const int xdst0 = xr * 128 + ((xc ^ swz128(xr)) << 4);       // rows 32 apart share the swizzle (32 % 16 == 0)
  const uint16_t* xp1 = a.X1 + (size_t)(m0 + xr) * a.ld1 + xc * 8;
  const uint16_t* xp2 = a.X2 ? a.X2 + (size_t)(m0 + xr) * a.ld2 + xc * 8 : xp1;
  uint4 px0, px1, px2, px3, pw0, pw1, pw2, pw3;   // named, not arrays: hipcc put the arrays in scratch
  const int K1 = a.K1, ld1 = a.ld1, ld2 = a.ld2;
  const uint4* wbase = reinterpret_cast<const uint4*>(a.Wp) + tid;
#define AIRFE_LOAD_CHUNK(IT)                                                                                  \
  {                                                                                                           \
    const int pr_ = pair0 + (IT) / NS, s_ = (IT) % NS, k0_ = s_ * 64;                                         \
    const bool first_ = k0_ < K1; /* K1 is a multiple of 64: the whole chunk is on one side */                \
    const uint16_t* src_ = first_ ? xp1 + k0_ : xp2 + (k0_ - K1);                                             \
    const size_t rstep_ = (size_t)32 * (first_ ? ld1 : ld2);                                                  \
    px0 = *reinterpret_cast<const uint4*>(src_);                                                              \
    px1 = *reinterpret_cast<const uint4*>(src_ + rstep_);                                                     \
    px2 = *reinterpret_cast<const uint4*>(src_ + 2 * rstep_);                                                 \
    px3 = *reinterpret_cast<const uint4*>(src_ + 3 * rstep_);                                                 \
    const uint4* w0_ = wbase + ((size_t)(2 * pr_) * NS + s_) * 512;                                           \
    pw0 = w0_[0]; pw1 = w0_[256]; pw2 = w0_[(size_t)NS * 512]; pw3 = w0_[(size_t)NS * 512 + 256];             \
  }
#define AIRFE_STORE_CHUNK(BUF)                                                                                \
  {                                                                                                           \
    char* xb_ = smem + (BUF) * STAGE;                                                                         \
    uint4* wb_ = reinterpret_cast<uint4*>(xb_ + 16384);                                                       \
    *reinterpret_cast<uint4*>(xb_ + xdst0) = px0;                                                             \
    *reinterpret_cast<uint4*>(xb_ + xdst0 + 4096) = px1;                                                      \
    *reinterpret_cast<uint4*>(xb_ + xdst0 + 8192) = px2;                                                      \
    *reinterpret_cast<uint4*>(xb_ + xdst0 + 12288) = px3;                                                     \
    wb_[tid] = pw0; wb_[256 + tid] = pw1; wb_[512 + tid] = pw2; wb_[768 + tid] = pw3;                         \
  }
  AIRFE_LOAD_CHUNK(0)
  AIRFE_STORE_CHUNK(0)
  __syncthreads();

  f32x4 acc[4][4];
  for (int it = 0; it < total; ++it) {
    const int s = it % NS;
    if (s == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (it + 1 < total) AIRFE_LOAD_CHUNK(it + 1)
    const char* xb = smem + (it & 1) * STAGE;
    const char* wb = xb + 16384 + wn * SLAB_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename P::vec8 wf[4], xf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rr = t * 16 + l15;
        wf[t] = lds_frag<P>(wb, rr * 128 + (((ks * 4 + g) ^ swz128(rr)) << 4));
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int r = wm * 64 + m * 16 + l15;
        xf[m] = lds_frag<P>(xb, r * 128 + (((ks * 4 + g) ^ swz128(r)) << 4));
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if constexpr (TRANS) acc[m][t] = P::mfma(xf[m], wf[t], acc[m][t]);
          else acc[m][t] = P::mfma(wf[t], xf[m], acc[m][t]);
        }
    }
    if (it + 1 < total) AIRFE_STORE_CHUNK((it + 1) & 1)
    if (s == NS - 1) {
      const int cb = 2 * (pair0 + it / NS) + wn;
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
                v[e] = acc[m][2 * tp][e];
                v[4 + e] = acc[m][2 * tp + 1][e];
              }
              gemm_store_run<P>(a, row, cb * 64 + tp * 32 + g * 8, v);
            }
          }
        } else {
          // transposed store: lane (feature row l15 of tile t, g) holds tokens g*4..g*4+3 of m-tile m
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
              *reinterpret_cast<uint2*>(o) = pack4<P>(acc[m][t][0] + bv, acc[m][t][1] + bv, acc[m][t][2] + bv, acc[m][t][3] + bv);
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

template <class P, bool TRANS>
static void gemm_launch_t(int K, const GemmArgs& a, hipStream_t st) {
  constexpr int LDS = 2 * 32768;
  static PerDeviceOnce attr_once;
  auto kfn = gemm_kernel<P, TRANS>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  const int mb = a.M / 128;
  const int npairs = (a.cb_total + 1) / 2;      // the weight buffer is packed with an even number of feature blocks
  // split the feature-block pairs over grid.y only as far as needed to keep two workgroups per CU resident
  int gy = 1;
  while (gy < npairs && (mb * gy < 512 || npairs % gy)) ++gy;
  dim3 grid((unsigned)mb, (unsigned)gy);
  hipLaunchKernelGGL(kfn, grid, dim3(256), LDS, st, a, K, npairs / gy);
}

// Small-M GEMM (the matcher on one or two pairs: 800-1600 tokens).  32 rows x 64 features per workgroup, no LDS and no K loop:
// every lane fetches its MFMA fragments for the WHOLE K extent straight from global memory (the packed weight slabs are
// stored in fragment order, an activation fragment is 16 contiguous bytes of a row), so a launch costs one L2 round trip plus
// 2*K/32 MFMAs per wave instead of K/64 load -> LDS -> barrier rounds; at M = 896 the staged kernel above is latency-bound at
// ~11 us per launch with 14-28 workgroups, this one runs 112-224 workgroups.  Accumulation order over K is the same (ascending
// 32-wide steps), so results are bit-identical to gemm_kernel's.
template <class P, bool TRANS, int NK>   // NK = K / 32
__global__ __launch_bounds__(256, 1) void gemm_small_kernel(GemmArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int tp = wave >> 1;                       // feature tile pair (32 features) of the 64-feature block
  const int cb = blockIdx.y, m0 = blockIdx.x * 32 + (wave & 1) * 16;
  constexpr int NS = NK / 2;
  const char* wbase = reinterpret_cast<const char*>(a.Wp) + (size_t)cb * NS * SLAB_BYTES;
  typename P::vec8 wf[2][NK], xf[NK];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rr = (2 * tp + u) * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
      wf[u][ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>(
          wbase + (ks >> 1) * SLAB_BYTES + rr * 128 + ((((ks & 1) * 4 + g) ^ swz128(rr)) << 4)));
  }
  const int row = m0 + l15;
  const uint16_t* x1 = a.X1 + (size_t)row * a.ld1 + g * 8;
  const uint16_t* x2 = a.X2 ? a.X2 + (size_t)row * a.ld2 + g * 8 - a.K1 : x1;
  const int K1 = a.X2 ? a.K1 : NK * 32;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks)
    xf[ks] = __builtin_bit_cast(typename P::vec8, *reinterpret_cast<const uint4*>((ks * 32 < K1 ? x1 : x2) + ks * 32));
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ks = 0; ks < NK; ++ks)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (TRANS) acc[u] = P::mfma(xf[ks], wf[u][ks], acc[u]);
      else acc[u] = P::mfma(wf[u][ks], xf[ks], acc[u]);
    }
  if (cb >= a.cb_total) return;
  if constexpr (!TRANS) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = acc[0][e];
      v[4 + e] = acc[1][e];
    }
    gemm_store_run<P>(a, row, cb * 64 + tp * 32 + g * 8, v);
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int co = cb * 64 + slab_row_to_feature((2 * tp + u) * 16 + l15);
      const float bv = a.bias[co];
      const int h = co >> 6, d = co & 63;
      const int row0 = m0 + g * 4;
      const int sq = row0 / a.Np, n = row0 - sq * a.Np;
      uint16_t* o = reinterpret_cast<uint16_t*>(a.out) + (((size_t)sq * a.H + h) * 64 + d) * a.Np + n;
      *reinterpret_cast<uint2*>(o) = pack4<P>(acc[u][0] + bv, acc[u][1] + bv, acc[u][2] + bv, acc[u][3] + bv);
    }
  }
}

template <class P, bool TRANS>
static void gemm_small_launch_t(int K, const GemmArgs& a, hipStream_t st) {
  dim3 grid((unsigned)(a.M / 32), (unsigned)a.cb_total);
  if (K == 128) hipLaunchKernelGGL((gemm_small_kernel<P, TRANS, 4>), grid, dim3(256), 0, st, a);
  else if (K == 256) hipLaunchKernelGGL((gemm_small_kernel<P, TRANS, 8>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_small_kernel<P, TRANS, 16>), grid, dim3(256), 0, st, a);
}

void launch_gemm(int prec, int K, bool trans, const GemmArgs& a, hipStream_t st) {
  if (a.M >= a.gr_min && gemmr_applicable(K, trans, a)) {   // HBM-bound K = 256 linears: weights in registers, tokens streamed
    launch_gemmr(prec, trans, a, st);
    return;
  }
  if (a.M % 256 == 0 && a.M >= a.g8_min && a.M > a.small_max) {       // large-M path: 8-wave, 3-stage LDS-DMA ring (kernels_gemm8.hip)
    launch_gemm8(prec, K, trans, a, st);
    return;
  }
  if (a.M <= a.small_max && a.M % 32 == 0 && (K == 128 || K == 256 || K == 512)) {
    if (prec == 1) {
      if (trans) gemm_small_launch_t<PF16, true>(K, a, st); else gemm_small_launch_t<PF16, false>(K, a, st);
    } else {
      if (trans) gemm_small_launch_t<PBF16, true>(K, a, st); else gemm_small_launch_t<PBF16, false>(K, a, st);
    }
    return;
  }
  if (prec == 1) {
    if (trans) gemm_launch_t<PF16, true>(K, a, st); else gemm_launch_t<PF16, false>(K, a, st);
  } else {
    if (trans) gemm_launch_t<PBF16, true>(K, a, st); else gemm_launch_t<PBF16, false>(K, a, st);
  }
}

}  // namespace airfe
