// airfe — persistent 3x3 convolution for the 128-input-channel layers (conv3b, conv4a, conv4b, convPa, convDa),
// 128 output channels per pass.
//
// The 128x128x9 filter bank (288 KiB) cannot live in LDS, so it streams PER TAP: 32 KiB (4 packed slabs) per tap through a
// two-stage LDS ring filled by LDS-DMA, in one endless cyclic stream across tiles (the pipeline never drains at a tile
// boundary).  The halo'ed input tile (16x8 pixels + halo = 10x18x128 ch = 45 KiB) is double-buffered and also arrives by
// LDS-DMA one tile ahead.  Per tap a wave issues 64 MFMAs (2 pixel rows x 8 cout tiles x 4 k-steps) against one barrier;
// the generic slab kernel (kernels_mm.hip) had one barrier per 16 MFMAs and re-streamed weights through VGPRs.
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr128;

__device__ __forceinline__ void c128_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int C128_TILE_BYTES = 10 * 18 * 256;     // 46080
constexpr int C128_PIECES = 10 * 18 * 16;          // 2880 sixteen-byte pieces
constexpr int C128_WSTAGE = 4 * SLAB_BYTES;        // one tap: [2 cout blocks][2 cin halves][8 KiB]

template <class P, bool POOL>
__global__ __launch_bounds__(256, 1) void conv128ws_kernel(ConvArgs a, int tiles_x, int tiles_y, int ntiles, int cb0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W, COUT = a.COUT;
  const size_t in_row = (size_t)(W + 2) * 256;
  const size_t in_img = (size_t)(H + 2) * in_row;
  const int per_img = tiles_x * tiles_y;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr128)smem);
  const unsigned wring = lds_base + 2 * C128_TILE_BYTES;

  // ---- loop-invariant per-thread offsets
  int goff[12];                                        // input DMA: piece q = j*256 + tid -> pixel q>>4, slot q&15
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int q = min(j * 256 + tid, C128_PIECES - 1);
    const int p = q >> 4, c = (q & 15) ^ swz256(p);
    const int pr = p / 18, pc = p - pr * 18;
    goff[j] = pr * (int)in_row + pc * 256 + c * 16;
  }
  const bool last_piece = 11 * 256 + tid < C128_PIECES;
  // weights of tap `tap`: blocks cb0 and cb0+1, both cin halves: two contiguous 16 KiB regions of the packed buffer
  const char* wsrc0 = reinterpret_cast<const char*>(a.Wp) + (size_t)cb0 * 18 * SLAB_BYTES + (size_t)tid * 16;
  int boff[4][3];                                      // pixel fragments: rows wave*2 + {0..3}, dx in {0,1,2}, chunk g
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = (wave * 2 + r) * 18 + l15 + dx;
      boff[r][dx] = p * 256 + ((g ^ swz256(p)) << 4);
    }
  int aoff[4];                                         // weight fragments inside one 8 KiB slab
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rr = t * 16 + l15;
    aoff[t] = rr * 128 + ((g ^ swz128(rr)) << 4);
  }
  float bias[2][2][8];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) bias[hb][tp][e] = a.bias[(cb0 + hb) * 64 + tp * 32 + g * 8 + e];

  auto stage_tile = [&](int tile, int buf) {
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const char* xin = reinterpret_cast<const char*>(a.X) + (size_t)b * in_img + (size_t)ty * 8 * in_row + (size_t)tx * 16 * 256;
    const unsigned dst = lds_base + buf * C128_TILE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < 11; ++j) c128_glds16(xin + goff[j], dst + j * 4096);
    if (last_piece) c128_glds16(xin + goff[11], dst + 11 * 4096);
  };
  auto stage_tap = [&](int tap, int wbuf) {
    const char* s0 = wsrc0 + (size_t)tap * 2 * SLAB_BYTES;          // (cb0, tap, cc = 0..1): 16 KiB contiguous
    const char* s1 = s0 + (size_t)18 * SLAB_BYTES;                  // (cb0 + 1, tap, cc = 0..1)
    const unsigned dst = wring + wbuf * C128_WSTAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) c128_glds16(s0 + i * 4096, dst + i * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) c128_glds16(s1 + i * 4096, dst + 16384 + i * 4096);
  };

  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int opad = a.out_pad;
  const size_t orow = (size_t)(Wo + 2 * opad) * COUT;

  int tile = blockIdx.x;
  if (tile < ntiles) {
    stage_tile(tile, 0);
    stage_tap(0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int wbuf = 0;
  for (int i = 0; tile < ntiles; ++i, tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const int xoff = (i & 1) * C128_TILE_BYTES;
    f32x4 acc[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // request the NEXT tap's weights (tap 0 of the next tile after tap 8) and, once per tile, the next input tile
      if (tap < 8) stage_tap(tap + 1, wbuf ^ 1);
      else if (next < ntiles) stage_tap(0, wbuf ^ 1);
      if (tap == 0 && next < ntiles) stage_tile(next, (i + 1) & 1);

      const int dy = tap / 3, dx = tap - dy * 3;
      const char* wst = smem + 2 * C128_TILE_BYTES + wbuf * C128_WSTAGE;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          typename P::vec8 bf[2];
#pragma unroll
          for (int m = 0; m < 2; ++m) bf[m] = lds_frag<P>(smem, (boff[m + dy][dx] ^ (ks << 6) ^ (cc << 7)) + xoff);
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            typename P::vec8 af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = lds_frag<P>(wst, (hb * 2 + cc) * SLAB_BYTES + (aoff[t] ^ (ks << 6)));
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[m][hb * 4 + t] = P::mfma(af[t], bf[m], acc[m][hb * 4 + t]);
          }
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next tap's weights (and, at tap 0, the next tile) have landed
      __syncthreads();
      wbuf ^= 1;
    }

    // ---- epilogue: bias, ReLU, optional 2x2 max-pool, full-line 16-byte stores
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int cofs = (cb0 + hb) * 64;
      if constexpr (!POOL) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float v0[8], v1[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = acc[m][hb * 4 + 0][e] + bias[hb][0][e];
            v0[4 + e] = acc[m][hb * 4 + 1][e] + bias[hb][0][4 + e];
            v1[e] = acc[m][hb * 4 + 2][e] + bias[hb][1][e];
            v1[4 + e] = acc[m][hb * 4 + 3][e] + bias[hb][1][4 + e];
          }
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
          }
          uint4 r1 = pack8<P>(v0), r2 = pack8<P>(v1);
          line_exchange(r1, r2, l15);
          const int y = ty * 8 + wave * 2 + m, x = tx * 16 + (l15 & 7);
          char* o = reinterpret_cast<char*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT + cofs) +
                    (l15 < 8 ? 0 : 64) + g * 16;
          *reinterpret_cast<uint4*>(o) = r1;
          *reinterpret_cast<uint4*>(o + (size_t)8 * COUT * 2) = r2;
        }
      } else {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = fmaxf(acc[0][hb * 4 + 0][e], acc[1][hb * 4 + 0][e]);
          v0[4 + e] = fmaxf(acc[0][hb * 4 + 1][e], acc[1][hb * 4 + 1][e]);
          v1[e] = fmaxf(acc[0][hb * 4 + 2][e], acc[1][hb * 4 + 2][e]);
          v1[4 + e] = fmaxf(acc[0][hb * 4 + 3][e], acc[1][hb * 4 + 3][e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v0[e] = fmaxf(v0[e], __shfl_xor(v0[e], 1)) + bias[hb][0][e];
          v1[e] = fmaxf(v1[e], __shfl_xor(v1[e], 1)) + bias[hb][1][e];
          if (a.relu) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
        }
        const uint4 r = (l15 & 1) ? pack8<P>(v1) : pack8<P>(v0);
        const int y = (ty * 8 + wave * 2) / 2, x = tx * 8 + (l15 >> 1);
        char* o = reinterpret_cast<char*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT + cofs) +
                  (l15 & 1) * 64 + g * 16;
        *reinterpret_cast<uint4*>(o) = r;
      }
    }
  }
}

template <class P, bool POOL>
static void conv128ws_launch_t(const ConvArgs& a, hipStream_t st) {
  constexpr int LDS = 2 * C128_TILE_BYTES + 2 * C128_WSTAGE;
  static bool attr_done = false;
  auto kfn = conv128ws_kernel<P, POOL>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / 8;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int grid = ntiles < 256 ? ntiles : 256;
  for (int cb0 = 0; cb0 < a.COUT / 64; cb0 += 2)          // 128 output channels per pass
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), LDS, st, a, tiles_x, tiles_y, ntiles, cb0);
}

// requires CIN == 128, COUT % 128 == 0, H % 8 == 0, W % 16 == 0
void launch_conv128ws(int prec, const ConvArgs& a, hipStream_t st) {
  if (prec == 1) {
    if (a.pool) conv128ws_launch_t<PF16, true>(a, st); else conv128ws_launch_t<PF16, false>(a, st);
  } else {
    if (a.pool) conv128ws_launch_t<PBF16, true>(a, st); else conv128ws_launch_t<PBF16, false>(a, st);
  }
}

}  // namespace airfe
