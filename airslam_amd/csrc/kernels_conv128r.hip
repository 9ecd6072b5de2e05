// airfe — persistent 3x3 convolution for the 128-input-channel layers (conv3b, conv4a, conv4b, convPa, convDa) with the
// filter bank RESIDENT IN REGISTERS, 128 output channels per pass.
//
// Eight waves, two per SIMD (a lone wave cannot issue MFMAs faster than every 16 cycles = half the matrix pipe's rate,
// tools/microbench/mfma_rate.hip).  Wave w owns 16 output channels for the WHOLE 16x8-pixel tile:
//   * its filters are 9 taps x 128 cin x 16 cout = 36 A-fragments = 144 VGPRs, loaded once per workgroup: no weight stream,
//     no per-tap barrier (the tap-streamed predecessor waited on a 32 KiB LDS-DMA every 64 MFMAs and sat at 37 % of peak);
//   * a pixel fragment (16 pixels x 32 channels) is fetched once per (column shift, k-step) and feeds the three filter rows:
//     10 ds_read_b128 per 24 MFMAs;
//   * LDS holds the double-buffered halo tile (2 x 45 KiB, LDS-DMA one tile ahead) and a 32 KiB output staging tile: a lane
//     owns only 4 couts of a pixel, so results are transposed through LDS into whole 256-byte pixel rows before they leave.
#include "common.h"
#include "kernels.h"

namespace airfe {

typedef __attribute__((address_space(3))) void* las_ptr128r;

__device__ __forceinline__ void c128r_glds16(const void* gsrc, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_off)
               : "memory");
}

constexpr int C128R_TILE_BYTES = 10 * 18 * 256;    // 46080: 16x8 pixels + 1-pixel halo, 128 channels of 2 bytes
constexpr int C128R_PIECES = 10 * 18 * 16;         // 2880 sixteen-byte pieces
constexpr int C128R_OUT_OFF = 2 * C128R_TILE_BYTES;
constexpr int C128R_LDS = C128R_OUT_OFF + 128 * 256;

template <class P, bool POOL>
__global__ __launch_bounds__(512, 1) void conv128r_kernel(ConvArgs a, int tiles_x, int tiles_y, int ntiles, int cb0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = cb0 + (wave >> 2), t = wave & 3;      // 64-cout block and 16-row tile inside its packed slabs
  const int H = a.H, W = a.W, COUT = a.COUT;
  const size_t in_row = (size_t)(W + 2) * 256;
  const size_t in_img = (size_t)(H + 2) * in_row;
  const int per_img = tiles_x * tiles_y;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(las_ptr128r)smem);

  // ---- filters: slab (cb, tap, cin half cc), rows t*16 + l15, k-step ks -> MFMA A fragments
  typename P::vec8 wreg[9][2][2];
  {
    const char* wp = reinterpret_cast<const char*>(a.Wp) + (size_t)cb * 18 * SLAB_BYTES;
    const int rr = t * 16 + l15;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint4 u = *reinterpret_cast<const uint4*>(wp + (tap * 2 + cc) * SLAB_BYTES + rr * 128 + (((ks * 4 + g) ^ swz128(rr)) << 4));
          wreg[tap][cc][ks] = __builtin_bit_cast(typename P::vec8, u);
        }
  }
  // the lane's 4 couts: accumulator rows g*4 + r of tile t  <->  features (t>>1)*32 + g*8 + (t&1)*4 + r of block cb
  const int f0 = (t >> 1) * 32 + g * 8 + (t & 1) * 4;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + f0);

  int goff[6];                                         // input DMA: piece q = j*512 + tid -> pixel q>>4, LDS slot q&15
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = min(j * 512 + tid, C128R_PIECES - 1);
    const int p = q >> 4;
    const int pr = p / 18, pc = p - pr * 18;
    const int c = (q & 15) ^ (pc & 15);                // chunk swizzle by COLUMN: a fragment's 16 consecutive pixels of a row still
                                                       // land on 16 different bank groups, and the read address splits into
                                                       // (per-lane column term) + (compile-time row offset) -> 3 registers, not 30
    goff[j] = pr * (int)in_row + pc * 256 + c * 16;
  }
  const bool last_piece = 5 * 512 + tid < C128R_PIECES;
  auto stage_tile = [&](int tile, int buf) {
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const char* xin = reinterpret_cast<const char*>(a.X) + (size_t)b * in_img + (size_t)ty * 8 * in_row + (size_t)tx * 16 * 256;
    const unsigned dst = lds_base + buf * C128R_TILE_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < 5; ++j) c128r_glds16(xin + goff[j], dst + j * 8192);
    if (last_piece) c128r_glds16(xin + goff[5], dst + 5 * 8192);
  };
  int cbase[3];                                        // pixel fragments: column l15 + dx, chunk g (row r adds r * 18 * 256)
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) cbase[dx] = (l15 + dx) * 256 + ((g ^ ((l15 + dx) & 15)) << 4);

  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int opad = a.out_pad;
  const size_t orow = (size_t)(Wo + 2 * opad) * COUT;
  char* ost = smem + C128R_OUT_OFF;

  int tile = blockIdx.x;
  if (tile < ntiles) stage_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int i = 0; tile < ntiles; ++i, tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < ntiles) stage_tile(next, (i + 1) & 1);
    const int xoff = (i & 1) * C128R_TILE_BYTES;
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = bias;
    // 12 combos (column shift dx, cin half cc, k-step ks): 10 pixel-row fragments, then 3 filter rows x 8 pixel rows
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const int dx = c >> 2, cc = (c >> 1) & 1, ks = c & 1;
      typename P::vec8 bf[10];
      const char* fb = smem + ((cbase[dx] ^ (ks << 6) ^ (cc << 7)) + xoff);
#pragma unroll
      for (int r = 0; r < 10; ++r) bf[r] = lds_frag<P>(fb, r * (18 * 256));
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = P::mfma(wreg[dy * 3 + dx][cc][ks], bf[m + dy], acc[m]);
    }
    // next tile landed (wait BEFORE this tile's stores: vmcnt counts them too); everyone is done with the buffer just read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: (bias is in the accumulators) round, ReLU / 2x2 max-pool on packed pairs, transpose through LDS
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    uint16_t* ybase = a.Y + (size_t)b * (Ho + 2 * opad) * orow + cb0 * 64;
    const int fch = (wave >> 2) * 8 + (f0 >> 3);        // 16-byte chunk of the 128-cout pass that holds this lane's 4 couts
    uint2 pk[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) pk[m] = uint2{P::pack2(acc[m][0], acc[m][1]), P::pack2(acc[m][2], acc[m][3])};
    if constexpr (!POOL) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int px = m * 16 + l15;
        uint2 v = pk[m];
        if (a.relu) { v.x = relu_packed(v.x); v.y = relu_packed(v.y); }
        *reinterpret_cast<uint2*>(ost + px * 256 + ((fch ^ (px & 15)) << 4) + (t & 1) * 8) = v;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = j * 512 + tid, px = q >> 4, c = q & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(ost + px * 256 + ((c ^ (px & 15)) << 4));
        const int y = ty * 8 + (px >> 4), x = tx * 16 + (px & 15);
        *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT + c * 8) = v;
      }
    } else {
#pragma unroll
      for (int mp = 0; mp < 4; ++mp) {
        uint2 v;
        v.x = max_packed_pre_relu(pk[2 * mp].x, pk[2 * mp + 1].x);
        v.y = max_packed_pre_relu(pk[2 * mp].y, pk[2 * mp + 1].y);
        v.x = max_packed_pre_relu(v.x, dpp_xor1(v.x));
        v.y = max_packed_pre_relu(v.y, dpp_xor1(v.y));
        if (a.relu) { v.x = relu_packed(v.x); v.y = relu_packed(v.y); }
        const int px = mp * 8 + (l15 >> 1);                                   // pooled pixel of the 8x4 output tile
        if (!(l15 & 1)) *reinterpret_cast<uint2*>(ost + px * 256 + ((fch ^ (px & 15)) << 4) + (t & 1) * 8) = v;
      }
      __syncthreads();
      {
        const int q = tid, px = q >> 4, c = q & 15;                           // 32 pixels x 16 chunks = 512 pieces
        const uint4 v = *reinterpret_cast<const uint4*>(ost + px * 256 + ((c ^ (px & 15)) << 4));
        const int y = ty * 4 + (px >> 3), x = tx * 8 + (px & 7);
        *reinterpret_cast<uint4*>(ybase + (size_t)(y + opad) * orow + (size_t)(x + opad) * COUT + c * 8) = v;
      }
    }
  }
}

template <class P, bool POOL>
static void conv128r_launch_t(const ConvArgs& a, hipStream_t st) {
  static PerDeviceOnce attr_once;
  auto kfn = conv128r_kernel<P, POOL>;
  if (auto once_token = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, C128R_LDS);
  }
  const int tiles_x = a.W / 16, tiles_y = a.H / 8;
  const int ntiles = tiles_x * tiles_y * a.B;
  const int grid = ntiles < 256 ? ntiles : 256;
  for (int cb0 = 0; cb0 < a.COUT / 64; cb0 += 2)          // 128 output channels per pass
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), C128R_LDS, st, a, tiles_x, tiles_y, ntiles, cb0);
}

// requires CIN == 128, COUT % 128 == 0, H % 8 == 0, W % 16 == 0
void launch_conv128r(int prec, const ConvArgs& a, hipStream_t st) {
  if (prec == 1) {
    if (a.pool) conv128r_launch_t<PF16, true>(a, st); else conv128r_launch_t<PF16, false>(a, st);
  } else {
    if (a.pool) conv128r_launch_t<PBF16, true>(a, st); else conv128r_launch_t<PBF16, false>(a, st);
  }
}

}  // namespace airfe
