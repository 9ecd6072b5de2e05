// airfe — weight-stationary persistent 3x3 convolution for the 64 -> 64 channel layers (conv1b, conv2a, conv2b:
// 29 of the encoder's 44.5 GFLOP per image).
//
// Why a second conv kernel: rocprofv3 PMCs on the generic slab-streaming kernel (kernels_mm.hip) showed its waves
// parked 35-53 % of the time (one barrier per 8 KiB weight slab, weights re-streamed from L2 for every 256-pixel
// tile, tile staging not overlapped).  For Cin = Cout = 64 the whole filter bank is only 72 KiB of 2-byte data:
//   * each wave keeps ALL 9 taps x 64 x 64 weights as MFMA A-fragments in registers (288 VGPRs; one wave per SIMD has
//     the full 512-entry file) -> no weight traffic and no LDS weight reads inside the loop at all;
//   * workgroups are persistent (one per CU) and walk 16x16-pixel tiles; the NEXT tile's halo'ed input (18x18x64 ch =
//     40.5 KiB) is fetched with global_load_lds (LDS-DMA, no VGPRs, swizzle applied on the source address) into the
//     other half of a double buffer while the current tile feeds 288 MFMAs per wave;
//   * exactly one barrier per tile; LDS read traffic drops to the 4 pixel fragments per 16 MFMAs.
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// LDS-DMA issued from inline asm so that hipcc does not see it: with the builtin the compiler inserts `s_waitcnt vmcnt(0)`
// before the very next ds_read (it cannot prove the DMA target and the tile being read are different buffers), which
// serialises the prefetch with the compute it is meant to overlap.  The wait is placed by hand before the tile barrier.
// (M0 = wave-uniform LDS byte address; data lands at M0 + lane*16.  Recipe: cdna_hip_programming.md §5.7.)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int C64_PATCH = 20 * 20;                 // fused conv1a: fp32 image patch per tile (halo 2)
constexpr int C64_TILE_BYTES = 18 * 18 * 128;     // 41472
constexpr int C64_CHUNKS = 18 * 18 * 8;           // 2592 sixteen-byte pieces

template <class P, bool POOL, int WREG_TAPS, bool FUSE1A>
__global__ __launch_bounds__(256, 1) void conv64ws_kernel(ConvArgs a, int tiles_x, int tiles_y, int ntiles, int cb0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const size_t in_row = (size_t)(W + 2) * 128;
  const size_t in_img = (size_t)(H + 2) * in_row;

  // ---- weights: taps [0, WREG_TAPS) live in registers as MFMA A fragments, the rest stay resident in LDS behind the two
  // input buffers (no per-slab barrier either way: they are loaded exactly once per workgroup)
  constexpr int WT = WREG_TAPS > 0 ? WREG_TAPS : 1;
  typename P::vec8 wreg[WT][2][4];
  char* wlds = smem + 2 * C64_TILE_BYTES;
  {
    const char* wp = reinterpret_cast<const char*>(a.Wp) + (size_t)cb0 * 9 * SLAB_BYTES;   // 64 output channels per pass
#pragma unroll
    for (int tap = 0; tap < WREG_TAPS; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int rr = t * 16 + l15;
          const uint4 u = *reinterpret_cast<const uint4*>(wp + tap * SLAB_BYTES + rr * 128 + (((ks * 4 + g) ^ swz128(rr)) << 4));
          wreg[tap][ks][t] = __builtin_bit_cast(typename P::vec8, u);
        }
    for (int q = tid; q < (9 - WREG_TAPS) * 512; q += 256)
      reinterpret_cast<uint4*>(wlds)[q] = reinterpret_cast<const uint4*>(wp + WREG_TAPS * SLAB_BYTES)[q];
  }
  float bias[2][8];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[tp][e] = a.bias[cb0 * 64 + tp * 32 + g * 8 + e];

  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int opad = a.out_pad;
  const size_t orow = (size_t)(Wo + 2 * opad) * a.COUT;
  const int per_img = tiles_x * tiles_y;

  // LDS-DMA of one halo tile: piece q' = j*256 + tid lands at LDS byte q'*16 (wave-uniform base + lane*16);
  // the piece that belongs there is channel chunk c = c' ^ swz(p) of pixel p = q'/8  (swizzle on the SOURCE side)
  // loop-invariant per-thread offsets, computed ONCE (PMCs showed 2.8 VALU instructions per MFMA when the swizzled LDS
  // addresses and the DMA source addresses were re-derived inside the tile loop)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr)smem);
  int goff[11];
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const int q = min(j * 256 + tid, C64_CHUNKS - 1);
    const int p = q >> 3, c = (q & 7) ^ swz128(p);
    const int pr = p / 18, pc = p - pr * 18;
    goff[j] = pr * (int)in_row + pc * 128 + c * 16;
  }
  const bool last_piece = 10 * 256 + tid < C64_CHUNKS;
  auto stage = [&](int tile, int buf) {
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const char* xin = reinterpret_cast<const char*>(a.X) + (size_t)b * in_img + (size_t)ty * 16 * in_row + (size_t)tx * 16 * 128;
    const unsigned dst = lds_base + buf * C64_TILE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < 10; ++j) glds16(xin + goff[j], dst + j * 4096);
    if (last_piece) glds16(xin + goff[10], dst + 10 * 4096);
  };
  // fragment read offsets inside a tile buffer: pixel rows wave*4 + {0..5}, column shifts {0,1,2}; ks = 1 is XOR 64
  int boff[6][3];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = (wave * 4 + r) * 18 + l15 + dx;
      boff[r][dx] = p * 128 + ((g ^ swz128(p)) << 4);
    }
  int aoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rr = t * 16 + l15;
    aoff[t] = 2 * C64_TILE_BYTES + rr * 128 + ((g ^ swz128(rr)) << 4);
  }

  // ---- fused conv1a (Cin = 1): the 18x18x64 input tile of conv1b is COMPUTED here from a 20x20 image patch instead of
  // being read back from HBM (33.5 MB/image of intermediate traffic and one kernel disappear).  K = 9 taps are padded
  // to one 32-wide fp16 MFMA step: lanes g=0 carry taps 0..7, g=1 tap 8, g=2,3 zeros.
  [[maybe_unused]] f16x8 w1[4];
  [[maybe_unused]] float bias1[2][8];
  [[maybe_unused]] const unsigned pbase = lds_base + 2 * C64_TILE_BYTES + 9 * SLAB_BYTES;
  [[maybe_unused]] const float* pbuf = reinterpret_cast<const float*>(smem + 2 * C64_TILE_BYTES + 9 * SLAB_BYTES);
  if constexpr (FUSE1A) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int co = slab_row_to_feature(t * 16 + l15);
      f16x8 w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = g * 8 + e;
        w[e] = (_Float16)(k < 9 ? a.w1a[co * 9 + k] : 0.f);
      }
      w1[t] = w;
    }
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) bias1[tp][e] = a.b1a[tp * 32 + g * 8 + e];
  }
  // 20x20 fp32 patch of tile `t` -> pbuf[buf] by 4-byte LDS-DMA (image buffer has a 1-pixel zero border; rows/cols that
  // fall outside even that are clamped: they only feed halo pixels that lie outside the image and are forced to 0)
  auto stage_patch = [&](int t, int buf) {
    const int b = t / per_img, rem = t - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const float* img = a.img + (size_t)b * (H + 2) * (W + 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = j * 256 + tid;
      if (q < C64_PATCH) {
        const int r = q / 20, cc = q - r * 20;
        const int gy = min(max(ty * 16 - 1 + r, 0), H + 1), gx = min(max(tx * 16 - 1 + cc, 0), W + 1);
        glds4(img + (size_t)gy * (W + 2) + gx, pbase + buf * (C64_PATCH * 4) + (j * 256 + wave * 64) * 4);
      }
    }
  };
  auto produce = [&](int t, int buf, int xbuf) {
    const int b = t / per_img, rem = t - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    (void)b;
    const float* pb = pbuf + buf * C64_PATCH;
    for (int k = wave; k < 21; k += 4) {
      const int p = k * 16 + l15, pc_ = min(p, 323);
      const int py = pc_ / 18, px = pc_ - py * 18;
      f16x8 bfr;
#pragma unroll
      for (int e = 0; e < 8; ++e) bfr[e] = (_Float16)0.f;
      if (g == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bfr[e] = (_Float16)pb[(py + e / 3) * 20 + px + e % 3];
      } else if (g == 1) {
        bfr[0] = (_Float16)pb[(py + 2) * 20 + px + 2];
      }
      f32x4 c1[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) c1[tt] = PF16::mfma(w1[tt], bfr, f32x4{0.f, 0.f, 0.f, 0.f});
      const int gy = ty * 16 - 1 + py, gx = tx * 16 - 1 + px;
      const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
      if (p < 324) {
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = inside ? fmaxf(c1[2 * tp][e] + bias1[tp][e], 0.f) : 0.f;
            v[4 + e] = inside ? fmaxf(c1[2 * tp + 1][e] + bias1[tp][4 + e], 0.f) : 0.f;
          }
          *reinterpret_cast<uint4*>(smem + xbuf * C64_TILE_BYTES + p * 128 + (((tp * 4 + g) ^ swz128(p)) << 4)) = pack8<P>(v);
        }
      }
    }
  };

  int tile = blockIdx.x;
  if constexpr (FUSE1A) {
    if (tile < ntiles) stage_patch(tile, 0);
  } else {
    if (tile < ntiles) stage(tile, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (FUSE1A) {
    // pipeline prologue: patch(1) in flight, conv1a tile(0) produced, both visible before the loop
    if (tile < ntiles) {
      if (tile + (int)gridDim.x < ntiles) stage_patch(tile + gridDim.x, 1);
      produce(tile, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  for (int i = 0; tile < ntiles; ++i, tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if constexpr (FUSE1A) {
      // patch(i+2) -> pbuf[i&1] (consumed by produce(i) one iteration ago); it has the whole MFMA phase to land
      if (next + (int)gridDim.x < ntiles) stage_patch(next + gridDim.x, i & 1);
    } else {
      if (next < ntiles) stage(next, (i + 1) & 1);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 18 k-steps (9 taps x 2 half-channel groups), software-pipelined by hand: the fragments of step s+1 are requested
    // BEFORE the 16 MFMAs of step s, so with one wave per SIMD the LDS latency hides under ~256 MFMA cycles
    typename P::vec8 af[2][4], bf[2][4];
    const int xoff = (i & 1) * C64_TILE_BYTES;
    auto load_step = [&](int step, int set) {
      const int tap = step >> 1, ks = step & 1;
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int m = 0; m < 4; ++m) bf[set][m] = lds_frag<P>(smem, (boff[m + dy][dx] ^ (ks << 6)) + xoff);
#pragma unroll
      for (int t = 0; t < 4; ++t) af[set][t] = lds_frag<P>(smem, (aoff[t] ^ (ks << 6)) + tap * SLAB_BYTES);
    };
    load_step(0, 0);
#pragma unroll
    for (int step = 0; step < 18; ++step) {
      if (step + 1 < 18) load_step(step + 1, (step + 1) & 1);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = P::mfma(af[step & 1][t], bf[step & 1][m], acc[m][t]);
    }

    // The wait for the next tile's LDS-DMA sits HERE, before this tile's output stores are issued: vmcnt also counts
    // stores, and waiting behind them exposed a full HBM write latency per tile (ablation: the kernel took 2.4 ms/step
    // with neither MFMAs nor DMA).  Now the stores get the whole next tile to retire.
    if constexpr (FUSE1A) {
      // conv1a of the NEXT tile goes into the other buffer right behind this tile's MFMAs (its patch landed an iteration
      // ago); then one wait + barrier covers "next input tile complete", "patch(i+2) landed" and "this buffer is free"
      if (next < ntiles) produce(next, (i + 1) & 1, (i + 1) & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile's LDS-DMA has landed
    __syncthreads();                                   // ... everyone's has, and the buffer just read is free again
    // ---- epilogue: bias, ReLU, optional 2x2 max-pool, 16-byte stores (same mapping as conv3x3_kernel)
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow;
    if constexpr (!POOL) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = fmaxf(acc[m][0][e] + bias[0][e], 0.f);
          v0[4 + e] = fmaxf(acc[m][1][e] + bias[0][4 + e], 0.f);
          v1[e] = fmaxf(acc[m][2][e] + bias[1][e], 0.f);
          v1[4 + e] = fmaxf(acc[m][3][e] + bias[1][4 + e], 0.f);
        }
        uint4 r1 = pack8<P>(v0), r2 = pack8<P>(v1);
        line_exchange(r1, r2, l15);                      // full 128-byte lines per store instruction (common.h)
        const int y = ty * 16 + wave * 4 + m, x = tx * 16 + (l15 & 7);
        char* o = reinterpret_cast<char*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * a.COUT + cb0 * 64) + (l15 < 8 ? 0 : 64) + g * 16;
        *reinterpret_cast<uint4*>(o) = r1;
        *reinterpret_cast<uint4*>(o + 8 * a.COUT * 2) = r2;
      }
    } else {
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = fmaxf(acc[2 * mp][0][e], acc[2 * mp + 1][0][e]);
          v0[4 + e] = fmaxf(acc[2 * mp][1][e], acc[2 * mp + 1][1][e]);
          v1[e] = fmaxf(acc[2 * mp][2][e], acc[2 * mp + 1][2][e]);
          v1[4 + e] = fmaxf(acc[2 * mp][3][e], acc[2 * mp + 1][3][e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v0[e] = fmaxf(fmaxf(v0[e], __shfl_xor(v0[e], 1)) + bias[0][e], 0.f);
          v1[e] = fmaxf(fmaxf(v1[e], __shfl_xor(v1[e], 1)) + bias[1][e], 0.f);
        }
        // both lanes of a pair hold the pooled pixel: the even lane stores its first 64 bytes, the odd lane the rest,
        // so ONE store instruction writes 8 complete 128-byte lines
        const uint4 r = (l15 & 1) ? pack8<P>(v1) : pack8<P>(v0);
        const int y = (ty * 16 + wave * 4) / 2 + mp, x = tx * 8 + (l15 >> 1);
        char* o = reinterpret_cast<char*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * a.COUT + cb0 * 64) + (l15 & 1) * 64 + g * 16;
        *reinterpret_cast<uint4*>(o) = r;
      }
    }

  }
}

constexpr int C64_WREG_TAPS = 0;     // taps kept in registers (0 = all nine filter taps resident in LDS)

template <class P, bool POOL, bool FUSE1A>
static void conv64ws_launch_t(const ConvArgs& a, hipStream_t st) {
  constexpr int LDS = 2 * C64_TILE_BYTES + (9 - C64_WREG_TAPS) * SLAB_BYTES + (FUSE1A ? 2 * C64_PATCH * 4 : 0);
  static bool attr_done = false;
  auto kfn = conv64ws_kernel<P, POOL, C64_WREG_TAPS, FUSE1A>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / 16;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int grid = ntiles < 256 ? ntiles : 256;
  for (int cb0 = 0; cb0 < a.COUT / 64; ++cb0) hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), LDS, st, a, tiles_x, tiles_y, ntiles, cb0);
}

// requires CIN == COUT == 64, relu, H % 16 == 0, W % 16 == 0.  a.img != nullptr selects the fused conv1a+conv1b form
// (input = fp32 image [B][H+2][W+2] with zero border, a.w1a [64][9], a.b1a [64]; always followed by the 2x2 max-pool).
void launch_conv64ws(int prec, const ConvArgs& a, hipStream_t st) {
  if (a.img) {
    if (prec == 1) conv64ws_launch_t<PF16, true, true>(a, st); else conv64ws_launch_t<PBF16, true, true>(a, st);
    return;
  }
  if (prec == 1) {
    if (a.pool) conv64ws_launch_t<PF16, true, false>(a, st); else conv64ws_launch_t<PF16, false, false>(a, st);
  } else {
    if (a.pool) conv64ws_launch_t<PBF16, true, false>(a, st); else conv64ws_launch_t<PBF16, false, false>(a, st);
  }
}

}  // namespace airfe
